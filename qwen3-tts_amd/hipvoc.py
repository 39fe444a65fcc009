"""placeholder, replaced below"""
class HipVocError(RuntimeError): pass
class HipVocoder: pass
def lib_path(): return None
def load_library(): raise HipVocError("not built")
