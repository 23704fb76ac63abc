"""Make tests/golden/oracle_memo.npz -- the memo of slow oracle evaluations of the GPU suite (see tests/oracle_memo.py).

The keys are hashes of the calls' inputs, which the tests build themselves (several on the device), so the entries are recorded by a
run of the suite, not enumerated here:

    DQ_ORACLE_MEMO_WRITE=gpurun_out/oracle_memo_new.npz python -m pytest tests -m gpu -q     # on the GPU box: records every call >= 1 s
    python tests/golden/make_oracle_memo.py gpurun_out/oracle_memo_new.npz                   # here: filter and write the fixture
    DQ_ORACLE_MEMO_CHECK=1 python -m pytest tests -m gpu -q                                  # --check: every answered key is recomputed
                                                                                             #   by the oracle and compared (1e-12)
    python tests/golden/make_oracle_memo.py --list                                           # what the fixture holds

Entries that would cost more than 300 KB per second of oracle time they save are left out (the oracle computes them on the spot).
Test infrastructure only: everything in the fixture is an output of oracle/trotter_oracle.py."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "oracle_memo.npz")


def entries(z):
    for k in z.files:
        if k.endswith(".layout"):
            h = k[:-7]
            n = len(str(z[k])) - 1
            yield h, n, sum(z["{}.{}".format(h, i)].nbytes for i in range(n)), float(z[h + ".seconds"]), str(z[h + ".fn"])


def main():
    if "--list" in sys.argv:
        with np.load(FIXTURE) as z:
            rows = sorted(entries(z), key=lambda r: -r[3])
            for h, n, b, sec, fn in rows:
                print("{}  {:24s} {:7.1f} s  {:9d} B".format(h, fn, sec, b))
            print("{} entries, {:.1f} MB, {:.0f} s of oracle time".format(len(rows), sum(r[2] for r in rows) / 1e6, sum(r[3] for r in rows)))
        return
    src = sys.argv[1]
    keep, saved, size, dropped = {}, 0.0, 0, 0
    with np.load(src) as z:
        for h, n, b, sec, fn in entries(z):
            if b > 600e3 and b / sec > 300e3:
                dropped += 1
                continue
            for suf in [".layout", ".fn", ".seconds"] + [".%d" % i for i in range(n)]:
                keep[h + suf] = z[h + suf]
            saved += sec
            size += b
    np.savez_compressed(FIXTURE, **keep)
    print("wrote {}: {} entries ({} left out), {:.1f} MB raw, {:.0f} s of oracle time".format(
        FIXTURE, sum(1 for k in keep if k.endswith(".layout")), dropped, size / 1e6, saved))


if __name__ == "__main__":
    main()
