import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for d in (HERE, ROOT):
    if d not in sys.path:
        sys.path.insert(0, d)


# Every device buffer the test wrappers allocate starts with its own byte pattern (srslte-emane_amd/__init__.py:DevBuf): a test that compares
# or a kernel that reads bytes nobody wrote then fails every time instead of once in a while. tests/tools/poison_hipmalloc.cpp does the same
# for the library's internal allocations (LD_PRELOAD).
os.environ.setdefault("SRSLTE_HIP_TEST_POISON", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
