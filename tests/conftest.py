import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "tuning: exercises kernel variants that only a TUNING=1 build of the library carries "
                            "(make -C tim_amd/csrc TUNING=1; TIM_AMD_LIB=tim_amd/libtimhip_tuning.so); skipped on the product library")


def pytest_collection_modifyitems(config, items):
    """`tuning` tests run only against a TUNING=1 library (decided without touching the GPU: a flag the library exports)"""
    need = [it for it in items if it.get_closest_marker("tuning")]
    if not need:
        return
    try:
        from tim_amd import _lib
        have = _lib.tuning_build()
    except Exception:  # noqa: BLE001  (no library here: the gpu tests are deselected anyway)
        have = False
    if not have:
        skip = pytest.mark.skip(reason="product build of libtimhip.so: the variant lives in the TUNING=1 build only")
        for it in need:
            it.add_marker(skip)


@pytest.fixture
def knobs():
    """set TIMHIP_* launcher knobs for one test: `knobs(TIMHIP_GEMM_LD="0", ...)`.  The library caches its knobs
    (timhip_reload_env), so the fixture re-reads them after setting and again after restoring the environment."""
    from tim_amd import _lib
    saved = {}

    def set_(**kw):
        for k, v in kw.items():
            if k not in saved:
                saved[k] = os.environ.get(k)
            os.environ[k] = str(v)
        _lib.reload_env()
    yield set_
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    _lib.reload_env()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
