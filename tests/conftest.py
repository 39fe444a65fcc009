import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import _pkgload  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def pkg():
    return _pkgload.load_pkg()


@pytest.fixture(scope="session")
def synth(pkg):
    return pkg.synth


@pytest.fixture(scope="session")
def vo():
    return _pkgload.load_oracle()


@pytest.fixture(scope="session")
def eo():
    return _pkgload.load_enc_oracle()


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


_W_CACHE = {}


def weights_for(synth, cfg_name, seed):
    key = (cfg_name, int(seed))
    if key not in _W_CACHE:
        _W_CACHE[key] = synth.make_weights(getattr(synth, cfg_name), int(seed))
    return _W_CACHE[key]


_O_CACHE = {}


def oracle_for(vo, synth, cfg_name, seed):
    key = (cfg_name, int(seed))
    if key not in _O_CACHE:
        cfg = getattr(synth, cfg_name)
        _O_CACHE[key] = vo.Oracle(weights_for(synth, cfg_name, seed), {k: float(v) for k, v in cfg.metadata().items()})
    return _O_CACHE[key]


def snr_db(ref, x):
    ref = np.asarray(ref, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    num = np.sum(ref * ref)
    den = np.sum((ref - x) ** 2)
    if den == 0:
        return float("inf")
    return 10.0 * np.log10(num / den)


_EW_CACHE = {}
_EO_CACHE = {}


def enc_weights_for(synth, cfg_name, seed):
    key = (cfg_name, int(seed))
    if key not in _EW_CACHE:
        _EW_CACHE[key] = synth.make_encoder_weights(getattr(synth, cfg_name), int(seed))
    return _EW_CACHE[key]


def enc_oracle_for(eo, synth, cfg_name, seed):
    key = (cfg_name, int(seed))
    if key not in _EO_CACHE:
        cfg = getattr(synth, cfg_name)
        _EO_CACHE[key] = eo.EncOracle(enc_weights_for(synth, cfg_name, seed),
                                      {k: float(v) for k, v in cfg.metadata().items()})
    return _EO_CACHE[key]
