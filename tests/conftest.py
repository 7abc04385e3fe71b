import json
import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

_REHEARSAL = {}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """GPU runs only: the multi-rank rehearsal (two ranks on cuda:0 over gloo, and the library's RCCL
    communicator with a world of one) needs fresh processes, and a process that has initialised the
    GPU must not be the one that starts them -- so they run here, before any test touches the GPU
    (torch.cuda.device_count() does not initialise it), and tests/test_gpu_kernels.py reads the result."""
    expr = session.config.getoption("-m") or ""
    if "gpu" not in expr or "not gpu" in expr:
        return
    try:
        import torch
        if torch.cuda.device_count() < 1:
            return
    except Exception:
        return
    out = tempfile.mkdtemp(prefix="gmrf_rehearsal_")
    _REHEARSAL["dir"] = out
    try:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rehearse_driver.py"), out], cwd=ROOT,
                           capture_output=True, text=True, timeout=900)
        _REHEARSAL["rc"] = p.returncode
        _REHEARSAL["tail"] = (p.stdout + p.stderr)[-2000:]
    except Exception as e:      # noqa: BLE001
        _REHEARSAL["rc"] = -1
        _REHEARSAL["tail"] = repr(e)
    pth = os.path.join(out, "result.json")
    if os.path.exists(pth):
        _REHEARSAL["result"] = json.load(open(pth))


@pytest.fixture(scope="session")
def rehearsal():
    return _REHEARSAL


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


@pytest.fixture(scope="session")
def lib(pkg):
    """The C-ABI library; GPU tests fail loudly when it has not been built."""
    return pkg._cabi.load()
