import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as ge  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return ge.load_package()


@pytest.fixture(scope="session")
def ck(pkg):
    return pkg.checkpoint


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (C restatement of src/main.zig) -- the checker."""
    o = ge.load_oracle()
    o.lib()
    yield o
    o.set_mode(8, False, False)


@pytest.fixture(scope="session")
def B(pkg):
    """ctypes binding over libllama2_hip.so; builds it if missing."""
    if not os.path.exists(pkg.binding.LIB_PATH):
        ge.build()
    pkg.binding.lib()
    return pkg.binding


@pytest.fixture(scope="session")
def gpu(B):
    """The HIP path.  Fails (does not skip) when no device is visible: GPU tests
    must never pass on a fallback."""
    n = B.device_count()
    assert n >= 1, "gpu-marked test started without a HIP device"
    return B


@pytest.fixture
def options(gpu):
    """Set tuning knobs of csrc/tunables.h for one test (they are read from the environment once,
    at first use, so tests go through l2z_option_set) and put the defaults back afterwards."""
    defaults = {"L2Z_ATTN_SPLIT": -1, "L2Z_ATTN_SPLIT_POS": -1, "L2Z_FUSE_SMALL": 1, "L2Z_PREFILL": 1, "L2Z_NO_GRAPH": 0,
                "L2Z_PF_CHUNK": 0, "L2Z_PF_PANEL": 1, "L2Z_PF_PANEL_MAX": -1, "L2Z_ARGMAX_XCHG": 1, "L2Z_GRID_CAP": 0,
                "L2Z_P2P_CONSUME": -1, "L2Z_SCHEME_B": 0, "L2Z_PF_X3": 1, "L2Z_PF_X3_STREAM_MIN": 33, "L2Z_PF_FUSE_PLANES": 1}
    touched = []

    def set_options(**kw):
        for k, v in kw.items():
            assert k in defaults, f"add the default of {k} to the options fixture"
            gpu.option_set(k, v)
            touched.append(k)
    yield set_options
    for k in touched:
        gpu.option_set(k, defaults[k])
