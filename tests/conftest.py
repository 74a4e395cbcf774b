import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# The 12-wave tap-fused weight-gradient kernel (wgrad_t3.hip) declines launches of fewer than 128 blocks in production (the per-tap
# kernel fills the chip better there).  The parity suite runs small volumes: lift the floor so that every eligible shape of the
# suite goes through that kernel -- the environment supplies the initial value of the library's switch table (config.hip); the
# weight-gradient op tests run the >= 64-channel cases a second time at the production floor (ops.config(M1_T3_MIN_BLOCKS=128)).
os.environ.setdefault("M1_T3_MIN_BLOCKS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "subprocess_last: harness tests that spawn bench.py / the trainer; collected last")


# Oracle-parity tests first, whole-model tests next, full-size property tests after them, subprocess harness tests
# last: with `-x` one flaky harness test must never hide the parity suite (round-1 lesson).
_ORDER = ["test_oracle_kat", "test_host_logic", "test_abi", "test_hip_ops", "test_hip_model", "test_hip_fused",
          "test_readme_model", "test_cascade", "test_trainer", "test_full_size"]


def pytest_collection_modifyitems(session, config, items):
    def key(it):
        mod = os.path.splitext(os.path.basename(str(it.fspath)))[0]
        last = 1 if it.get_closest_marker("subprocess_last") is not None else 0
        rank = _ORDER.index(mod) if mod in _ORDER else len(_ORDER)
        return (last, rank)
    items.sort(key=key)          # stable: the order inside a module is kept


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("prostatemr_3d-cad-cspca_amd")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
