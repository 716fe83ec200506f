#!/usr/bin/env python3
"""The reference's WHOLE default CTest matrix for the test programs of the link-time drop-in (oracle/ref_hip.mk), not the sample of
tests/test_gpu_dropin.py: every `add_test(... pdsch_test ...)` line of lib/src/phy/phch/test/CMakeLists.txt:97-200, the pusch_test loops of
:248-316 (default, non-"Paranoid" extension) and the phy_dl_test loops of lib/test/phy/CMakeLists.txt:27-58 (6 bandwidths x 256QAM off / on x
TM1-4 x MCS 0 / 7 / 14 / 21 / 28), each run as a child process against libsrslte_phy_hip.so on the GPU box; exit code 0 = the reference's own
pass criterion. TM3 rows of phy_dl_test run the binary whose one difference is -fsigned-zeros on the reference's mimo/precoding.c
(phy_dl_test_sz, oracle/ref_hip.mk: with this image's gcc the reference's -Ofast folds the CDD pre-decoder's sign masks).

  python scripts/dropin_full_matrix.py --regen     (where /root/reference exists) rewrites tests/golden/ctest_pdsch_test_args.json: the
                                                   argument lists of the pdsch_test lines (data of the reference's test configuration)
  python scripts/dropin_full_matrix.py [out.txt]   (GPU box) runs everything, prints a summary, writes one line per invocation"""
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
BIN = os.path.join(ROOT, "oracle", "_ref", "hip")
FIX = os.path.join(ROOT, "tests", "golden", "ctest_pdsch_test_args.json")


def regen():
    src = open("/root/reference/lib/src/phy/phch/test/CMakeLists.txt").read()
    rows = [m.group(1).split() for m in re.finditer(r"^add_test\(\S+\s+pdsch_test\s*([^)]*)\)", src, re.M)]
    with open(FIX, "w") as f:
        json.dump(rows, f)
    print("%d pdsch_test invocations -> %s" % (len(rows), FIX))


def matrix():
    rows = [("pdsch_test", a) for a in json.load(open(FIX))]
    for n_prb in (50,):  # cell_n_prb_valid of the default extension, for every cell bandwidth that holds it
        for cell in (6, 15, 25, 50, 75, 100):
            if n_prb > cell:
                continue
            for mcs in range(0, 29, 10):
                for ack in (-1, 0):
                    for cqi in ("none", "wideband"):
                        m, a = mcs, ["-n", str(cell), "-L", str(n_prb)]
                        if ack != -1:
                            a += ["-p", "uci_ack", str(ack)]
                            m = 27 if m == 28 else m
                        if cqi != "none":
                            a += ["-p", "cqi", cqi]
                        rows.append(("pusch_test", a + ["-m", str(m)]))
    rows.append(("phy_dl_test", []))
    for cell in (6, 15, 25, 50, 75, 100):
        for q256 in (0, 1):
            for tm in (1, 2, 3, 4):
                for mcs in range(0, 29, 7):
                    a = ["-p", str(cell), "-t", str(tm)]
                    if q256:
                        mcs = (26 if cell == 15 else 27) if mcs == 28 else mcs
                        a.append("-q")
                    rows.append(("phy_dl_test_sz" if tm == 3 else "phy_dl_test", a + ["-m", str(mcs)]))
    return rows


def main():
    if "--regen" in sys.argv:
        return regen()
    out = sys.argv[1] if len(sys.argv) > 1 else None
    rows, bad, lines, t0 = matrix(), [], [], time.time()
    for i, (prog, args) in enumerate(rows):
        t = time.time()
        try:
            r = subprocess.run([os.path.join(BIN, prog)] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
            rc, tail = r.returncode, r.stdout.decode(errors="replace")[-400:]
        except subprocess.TimeoutExpired:
            rc, tail = -999, "timeout"
        lines.append("%-4s %6.2f s  %s %s" % ("ok" if rc == 0 else "FAIL", time.time() - t, prog, " ".join(args)))
        if rc != 0:
            bad.append((prog, args, rc, tail))
        if i % 25 == 24:
            print("%d / %d, %d failed, %.0f s" % (i + 1, len(rows), len(bad), time.time() - t0), flush=True)
    summary = "%d invocations, %d exit 0, %d failed, %.0f s" % (len(rows), len(rows) - len(bad), len(bad), time.time() - t0)
    print(summary)
    for prog, args, rc, tail in bad:
        print("FAIL rc=%d: %s %s\n%s\n" % (rc, prog, " ".join(args), tail))
    if out:
        with open(out, "w") as f:
            f.write("# scripts/dropin_full_matrix.py: the reference's default CTest matrix of pdsch_test, pusch_test and phy_dl_test through the drop-in\n")
            f.write("# " + summary + "\n" + "\n".join(lines) + "\n")
            for prog, args, rc, tail in bad:
                f.write("\nFAIL rc=%d: %s %s\n%s\n" % (rc, prog, " ".join(args), tail))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
