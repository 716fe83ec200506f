#!/usr/bin/env python3
"""The reference's WHOLE default CTest matrix for the test programs of the link-time drop-in (oracle/ref_hip.mk), not the sample of
tests/test_gpu_dropin.py: every `add_test(... pdsch_test ...)` line of lib/src/phy/phch/test/CMakeLists.txt:97-200, the pusch_test loops of
:248-316 (default, non-"Paranoid" extension) and the phy_dl_test loops of lib/test/phy/CMakeLists.txt:27-58 (6 bandwidths x 256QAM off / on x
TM1-4 x MCS 0 / 7 / 14 / 21 / 28), each run as a child process against libsrslte_phy_hip.so on the GPU box; exit code 0 = the reference's own
pass criterion. TM3 rows of phy_dl_test run the binary whose one difference is -fsigned-zeros on the reference's mimo/precoding.c
(phy_dl_test_sz, oracle/ref_hip.mk: with this image's gcc the reference's -Ofast folds the CDD pre-decoder's sign masks).

  python scripts/dropin_full_matrix.py --regen     (where /root/reference exists) rewrites tests/golden/ctest_pdsch_test_args.json: the
                                                   argument lists of the pdsch_test lines (data of the reference's test configuration)
  python scripts/dropin_full_matrix.py [out.txt]   (GPU box) runs everything, prints a summary, writes one line per invocation
  python scripts/dropin_full_matrix.py --paranoid-sample [out.txt]   a slice of upstream's "Paranoid" pusch_test extension: every allocation size"""
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


def paranoid_sample():
    """A slice of the "Paranoid" extension of the pusch_test loops (CMakeLists.txt:250-262): EVERY valid PUSCH allocation size of a 100-PRB cell
    (34 sizes 2^a 3^b 5^c: every SC-FDMA transform length) x MCS 0 / 14 / 28 (27 with HARQ-ACK, as upstream's loop) x without / with a
    HARQ-ACK bit x without / with a CQI report; the 64QAM rows with `-p enable_64qam`."""
    sizes = (1, 2, 3, 4, 5, 6, 8, 9, 10, 12, 15, 16, 18, 20, 24, 25, 27, 30, 32, 36, 40, 45, 48, 50, 54, 60, 64, 72, 75, 80, 81, 90, 96, 100)
    rows = []
    for n_prb in sizes:
        for mcs in (0, 14, 28):
            for ack in (-1, 1):
                for cqi in ("none", "wideband"):
                    m, a = mcs, ["-n", "100", "-L", str(n_prb)]
                    if ack != -1:
                        a += ["-p", "uci_ack", str(ack)]
                        m = 27 if m == 28 else m
                    if cqi != "none":
                        a += ["-p", "cqi", cqi]
                    if m >= 21:  # pusch_test leaves 64QAM off by default, pusch.c:326-331 then sends these MCS as 16QAM - at a code rate above 1
                        a += ["-p", "enable_64qam", "1"]  # for the top MCS, which nothing can decode; with 64QAM on they are what the table says
                    rows.append(("pusch_test", a + ["-m", str(m)]))
    return rows


def main():
    if "--regen" in sys.argv:
        return regen()
    sample = "--paranoid-sample" in sys.argv
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out = args[0] if args else None
    rows, bad, lines, t0 = (paranoid_sample() if sample else matrix()), [], [], time.time()
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
            f.write("# scripts/dropin_full_matrix.py%s: %s through the drop-in\n" % (" --paranoid-sample" if sample else "", "every valid PUSCH allocation size of a 100-PRB cell (pusch_test, a slice of upstream's Paranoid extension)" if sample else "the reference's default CTest matrix of pdsch_test, pusch_test and phy_dl_test"))
            f.write("# " + summary + "\n" + "\n".join(lines) + "\n")
            for prog, args, rc, tail in bad:
                f.write("\nFAIL rc=%d: %s %s\n%s\n" % (rc, prog, " ".join(args), tail))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
