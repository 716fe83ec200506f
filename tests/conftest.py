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

# torch brings its own HIP runtime (libamdhip64.so of its ROCm build); the product library links the system's. Whichever is loaded first serves
# both, and torch only finds the GPU through its own: a selection of tests that loads the library before anything imports torch then fails in
# the one test that uses torch.cuda ("No HIP GPUs are available", tests/test_gpu_fullsize.py::test_cfg4_eight_ues_on_one_device run on its own).
# The whole suite never saw it because test_dist_gloo.py imports torch at collection time. Make that order a property of every selection.
try:
    import torch  # noqa: F401,E402
except ImportError:  # the CPU-side tests that need no torch still run
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
