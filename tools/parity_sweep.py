#!/usr/bin/env python3
"""One-off wide parity sweep on a GPU box: product (HIP path through the C-ABI) against the C
oracle (test infrastructure) on many seeded scenarios, FASTA byte-for-byte and per-level
abundance traces to 1e-9.  Usage: python3 tools/parity_sweep.py FIRST_SEED N [--big]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sc_testlib as T  # noqa: E402


big_case = T.big_case


def main():
    first, n = int(sys.argv[1]), int(sys.argv[2])
    big = "--big" in sys.argv
    bad = crashed = 0
    t0 = time.time()
    for seed in range(first, first + n):
        d = tempfile.mkdtemp(prefix="sweep%d_" % seed)
        if big:
            args, kw = big_case(seed, d)
        else:
            args, kw = T.make_case(seed, d), T.scenario(seed)[0]
        try:
            exp_fa, exp_tr = T.run_oracle(args, d, trace=True, check=False)
            if exp_fa is None:
                crashed += 1
                print("seed %d: the oracle (like the reference) crashes on this input -- no defined output, skipped" % seed, flush=True)
                continue
            trf = os.path.join(d, "trace.txt")
            got_fa = T.run_product(args, trace_file=trf)
            assert got_fa == exp_fa, "FASTA differs"
            T.compare_traces(open(trf).read(), exp_tr)
            print("seed %d ok (%d contigs) %.0fs" % (seed, got_fa.count(">"), time.time() - t0), flush=True)
        except Exception as e:   # noqa: BLE001
            bad += 1
            print("seed %d FAILED: %s | %r" % (seed, str(e)[:300], kw), flush=True)
    print("sweep done: %d scenarios, %d failures, %d without a defined reference output, %.0f s" % (n, bad, crashed, time.time() - t0), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
