"""Import helper: the package directory is ``qwen3-tts_amd`` (hyphen), the module ``qwen3_tts_amd``."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))


def load_pkg():
    name = "qwen3_tts_amd"
    if name in sys.modules:
        return sys.modules[name]
    d = os.path.join(ROOT, "qwen3-tts_amd")
    spec = importlib.util.spec_from_file_location(name, os.path.join(d, "__init__.py"),
                                                  submodule_search_locations=[d])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_oracle():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import voc_oracle
    return voc_oracle


def load_enc_oracle():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import enc_oracle
    return enc_oracle
