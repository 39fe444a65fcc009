"""hip-vocoder: MI355X-native vocoder (speech-tokenizer decoder) backend for qwen3-tts.

The directory name carries a hyphen (it mirrors the reference's name), so import it through
``tests/conftest.py::load_pkg`` / ``_pkgload.py`` under the module name ``qwen3_tts_amd``.
"""
from . import sharding, synth  # noqa: F401
from . import hipvoc  # noqa: F401
from . import wire  # noqa: F401
from . import voice  # noqa: F401
from . import encoder  # noqa: F401
from .encoder import HipSpeechTokenizer  # noqa: F401
from .hipvoc import HipVocoder, HipVocError, lib_path, load_library  # noqa: F401
