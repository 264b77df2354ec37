import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the native pieces are build products (git-ignored): build them once if this checkout has none
    lib = os.path.join(ROOT, "climate_toolbox_amd", "lib", "libwagg.so")
    orc = os.path.join(ROOT, "oracle", "_build", "libwagg_oracle.so")
    if not (os.path.exists(lib) and os.path.exists(orc)):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def ref_fixture(golden_dir):
    """The reference's own pytest inputs (replayed) + restatement-derived expectations."""
    import numpy as np
    from tests.fixture_replay import reference_fixture, sha256_of
    fx = reference_fixture()
    gold = np.load(os.path.join(golden_dir, "reference_fixture.npz"), allow_pickle=False)
    assert sha256_of(fx["temp"]) == str(gold["temp_sha256"]), "legacy RNG replay drifted"
    for k in ("seg_lat", "seg_lon", "areawt", "popwt", "hierid", "ISO"):
        np.testing.assert_array_equal(fx[k], gold[k])
    return fx, gold

