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
