"""The pure host side of the library (fec_tables.cpp: code-block segmentation, QPP interleaver tables for every block length and window
count, rate de-matching tables for every redundancy version; compat_refsignal.cpp: CRS / MBSFN-RS tables with put / get on exactly sized
grids, filter taps, the 25.212 interleaver for every block size) under AddressSanitizer + UndefinedBehaviorSanitizer. The GPU pool offers no
sanitizer for device code; this is the CPU build the environment allows (race / memory checking)."""
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_host_tables_under_asan_ubsan():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "drv")
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-D__HIP_PLATFORM_AMD__",
                               "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "srslte-emane_amd", "csrc"),
                               os.path.join(HERE, "host_asan_driver.cpp"), os.path.join(ROOT, "srslte-emane_amd", "csrc", "fec_tables.cpp"),
                               os.path.join(ROOT, "srslte-emane_amd", "csrc", "compat_refsignal.cpp"),
                               "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
        out = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
        assert out.returncode == 0, out.stdout + out.stderr
        assert "host sanitizer run ok" in out.stdout and "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr
