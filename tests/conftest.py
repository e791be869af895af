"""Shared fixtures.  CPU tests (-m "not gpu") exercise the oracle, the golden vectors, host logic
and symbol export; GPU tests (-m gpu) are the HIP-vs-oracle parity tests through the C-ABI."""
import importlib.util
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by gpurun / the round-end driver)")


def _load_pkg():
    name = "beatrice_vst_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(REPO, "beatrice-vst_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def bv():
    return _load_pkg()


@pytest.fixture(scope="session")
def built():
    """Build the oracle and the product if their shared objects are missing (no-op on the GPU box,
    where the prebuilt .so files travel with the snapshot)."""
    if not os.path.exists(os.path.join(REPO, "oracle", "libbeatrice_oracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle"), "libbeatrice_oracle.so"], env={k: v for k, v in os.environ.items() if k != "LD_PRELOAD"})
    if not os.path.exists(os.path.join(REPO, "beatrice-vst_amd", "csrc", "libbeatrice_hip.so")):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "beatrice-vst_amd"), "-j8"], env={k: v for k, v in os.environ.items() if k != "LD_PRELOAD"})
    return True


@pytest.fixture(scope="session")
def model_dir(tmp_path_factory):
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import make_model
    d = str(tmp_path_factory.mktemp("model"))
    make_model.make_model(d, n_speakers=3)
    return d


@pytest.fixture(scope="session")
def oracle(bv, built):
    return bv.Abi(os.path.join(REPO, "oracle", "libbeatrice_oracle.so"))


@pytest.fixture(scope="session")
def product(bv, built):
    # torch ships its own copy of the HIP runtime.  A process that uses both torch's GPU side and the product must let
    # torch bring its runtime up FIRST (as bench.py does): the product's libamdhip64 dependency then resolves to the
    # copy already loaded, and both see the GPU.  The other order leaves torch without a device.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    return bv.load_product()
