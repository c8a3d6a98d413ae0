import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    # torch first: torch bundles its own HIP runtime, the C-ABI library links the system one; whichever is loaded first
    # serves the whole process, and a torch that comes second finds no GPU.  Tests that hand torch device tensors to the
    # *_device entry points need torch's, so it is loaded before the library (same order as bench.py).
    import torch  # noqa: F401
    import tetra_amd
    return tetra_amd.pkg


@pytest.fixture(scope="session")
def synth(pkg):
    return pkg.synth


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.build()
    return binding


@pytest.fixture(scope="session")
def emul():
    from tests.emul import emul_bind
    emul_bind.build()
    return emul_bind


@pytest.fixture(scope="session")
def ref():
    """The reference's own phy/tetra_burst.c built into oracle/_ref (oracle/build_ref.sh)."""
    from oracle import ref_binding
    if not ref_binding.available():
        pytest.skip("oracle/_ref not built and /root/reference not present")
    return ref_binding
