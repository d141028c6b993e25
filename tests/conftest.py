import importlib
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hrt():
    """The product package (directory name has hyphens, hence importlib)."""
    return importlib.import_module("nvidia-optix-ray-tracer_amd")


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def gpu_available():
    import torch
    return torch.cuda.is_available()


@pytest.fixture(params=["production", "counting"])
def renderer(hrt, gpu_available, request):
    """Every scenario test runs twice: on the production kernels (context flags 0: the default fused path mode, v_rcp_f32
    in the slab test, postponed leaf passes -- what bench.py times) and on the counting build (HRT_CTX_COUNT: wavefront
    kernels, exact division, canonical walk order)."""
    if not gpu_available:
        pytest.skip("no GPU in this container")
    r = hrt.Renderer(0, 0 if request.param == "production" else hrt.CTX_COUNT)
    yield r
    r.close()
