import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The GPU tests run every kernel behind a launch that fills the LDS of all CUs with NaN bit patterns, and on a workspace
# that starts as NaN (mmnn_sts_amd/csrc/common.hpp: MMNN_LAUNCH).  A kernel that reads an LDS or workspace word it never
# wrote (typically a zero-weight padding lane of an MFMA operand) then fails deterministically, instead of passing or
# failing with whatever its predecessor left behind.  Read once, when the library launches its first kernel.
os.environ.setdefault("MMNN_POISON_LDS", "1")
os.environ.setdefault("MMNN_POISON_WS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
