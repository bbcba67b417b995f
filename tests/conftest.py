import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full 256^3 problem on one GPU (about two minutes; still part of -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(autouse=True)
def _skip_cases_of_kernels_that_are_not_built(monkeypatch):
    """kernel variants 2 and 3 for Q3..Q5 are the superseded kernels ns_ho.hip / ns_hop.hip: in the library only when it was built
    with ADAFLO_BUILD_VARIANTS=1 (adaflo_amd/build.py).  Their parity cases are skipped, not failed, on the product build."""
    import adaflo_amd
    from adaflo_amd import _lib
    orig = adaflo_amd.NavierStokesMatrix.set_kernel_variant

    def guarded(self, variant):
        try:
            return orig(self, variant)
        except _lib.AdafloError as e:
            if "not in this build" in str(e):
                pytest.skip("kernel variant %d is not in this build of the library (ADAFLO_BUILD_VARIANTS=1)" % variant)
            raise
    monkeypatch.setattr(adaflo_amd.NavierStokesMatrix, "set_kernel_variant", guarded)
