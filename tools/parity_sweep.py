#!/usr/bin/env python3
"""Wide parity sweep on a GPU box: product (HIP path through the C-ABI) against the C oracle (test
infrastructure) on many seeded scenarios, FASTA byte-for-byte and per-level abundance traces to 1e-9.
The oracle runs of different scenarios proceed in parallel on the host cores; the product runs one
region at a time.  Usage: python3 tools/parity_sweep.py FIRST_SEED N [--big | --params] [--jobs J]"""
import concurrent.futures as cf
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
    params = "--params" in sys.argv
    jobs = int(sys.argv[sys.argv.index("--jobs") + 1]) if "--jobs" in sys.argv else 1
    bad = crashed = 0
    t0 = time.time()

    def prepare(seed):
        d = tempfile.mkdtemp(prefix="sweep%d_" % seed)
        if big:
            args, kw = big_case(seed, d)
        elif params:
            args, kw = T.param_case(seed, d)
            kw = dict(kw, argv=args[:-2])
        else:
            args, kw = T.make_case(seed, d), T.scenario(seed)[0]
        return seed, d, args, kw

    def oracle(job):
        seed, d, args, kw = job
        try:
            return job, T.run_oracle(args, d, trace=True, check=False), None
        except Exception as e:   # noqa: BLE001
            return job, None, e

    with cf.ThreadPoolExecutor(max_workers=max(jobs, 1)) as pool:
        futs = [pool.submit(oracle, prepare(seed)) for seed in range(first, first + n)]
        for fut in futs:
            (seed, d, args, kw), res, err = fut.result()
            try:
                if err is not None:
                    raise err
                exp_fa, exp_tr = res
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
