import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # The oracle is ~21 000 tiny ATen calls per train step: on a CPU-only box with other work on its cores the
    # intra-op thread pool makes every one of them a contended fork/join (measured here: 2.3 s single-threaded,
    # 44 s with 2 threads, 130 s with 4 for one whole-model case).  RAL_TEST_THREADS overrides.
    import torch
    if os.environ.get("RAL_TEST_THREADS"):
        torch.set_num_threads(int(os.environ["RAL_TEST_THREADS"]))
    elif not torch.cuda.is_available():
        torch.set_num_threads(1)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionstart(session):
    """`RAL_TEST_OPTIONS="attn_f16=0,mlp_fwd_w=2"`: process-wide library switches for this test process, set through the C ABI
    (ral_global_option) before the library is first used - the library itself reads no environment variable for them.
    The tests that run every kernel choice start pytest subprocesses with this variable."""
    import os
    opts = os.environ.get("RAL_TEST_OPTIONS", "")
    if not opts:
        return
    from ecg_denoise_amd import _lib
    _lib.apply_options(opts)
