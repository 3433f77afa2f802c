import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402
from tools import gen_model
from oracle import oracle as orc  # noqa: E402  (tests may use the checker)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def w2xc():
    """The product package (waifu2x-converter-cpp_amd) -- import fails loudly if the .so is missing."""
    lib = os.path.join(graft.PKG_DIR, "lib", "libw2xc_hip.so")
    if not os.path.exists(lib):
        graft.build()
    return graft.load_package()


@pytest.fixture(scope="session")
def oracle_built():
    if not os.path.exists(orc.ORACLE_SO):
        orc.build()
    return True


@pytest.fixture(scope="session")
def models_dir(tmp_path_factory):
    """Synthetic models in the reference JSON schema (the shipped ones are stripped from the reference)."""
    d = str(tmp_path_factory.mktemp("models"))
    for name in ("noise1", "noise2", "scale2.0x"):
        gen_model.ensure_model(name, d)
    return d


@pytest.fixture(scope="session")
def noise1_layers():
    return gen_model.synth_layers(seed=gen_model.SEEDS["noise1"])


@pytest.fixture(scope="session")
def scale_layers():
    return gen_model.synth_layers(seed=gen_model.SEEDS["scale2.0x"])


def small_layers(planes, seed):
    return gen_model.synth_layers(planes, seed)


def rand_plane(h, w, seed):
    return np.random.default_rng(seed).random((h, w), dtype=np.float32)


def ramp_plane(h, w):
    """asymmetric (x and y distinguishable, non-linear) test image"""
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    return (0.013 * x + 0.029 * y + 0.0007 * x * y + 0.1 * np.sin(0.37 * x) * np.cos(0.21 * y)).astype(np.float32)


# fp32 tolerance from BASELINE.json north_star: 1e-4 relative (+ a small atol for near-zero outputs,
# SURVEY 8c) and the global max-norm form
RTOL, ATOL = 1e-4, 1e-5


def assert_close(got, want, what=""):
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = max(float(np.abs(want).max()), 1e-30)
    err = float(np.abs(got - want).max())
    assert np.allclose(got, want, rtol=RTOL, atol=ATOL), "%s: max abs err %g (max |want| %g)" % (what, err, scale)
    assert err / scale <= RTOL, "%s: max-norm rel err %g" % (what, err / scale)
