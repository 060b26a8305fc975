#!/usr/bin/env python3
"""Parity sweep of the MANY-REGIONS path on a GPU box: the seeded scenarios of tools/parity_sweep.py, but `R` of them in
flight at a time in one context (stage5.run_regions), so that their levels meet in the level server's batches -- mixed
kinds in one launch (k_level_any), shared launch streams.  FASTA byte-for-byte against the C oracle (test
infrastructure), whose runs proceed in parallel on the host cores first.
Usage: python3 tools/parity_sweep_inflight.py FIRST_SEED N [--params] [--inflight R] [--jobs J] [--chunk C]"""
import concurrent.futures as cf
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sc_testlib as T  # noqa: E402


def opt(name, default):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


def main():
    first, n = int(sys.argv[1]), int(sys.argv[2])
    params = "--params" in sys.argv
    inflight, jobs, chunk = opt("--inflight", 64), opt("--jobs", 14), opt("--chunk", 512)
    from rambl_amd import capi, cli, stage5
    bad = crashed = done = 0
    t0 = time.time()

    def prepare(seed):
        d = tempfile.mkdtemp(prefix="sweepf%d_" % seed)
        if params:
            args, kw = T.param_case(seed, d)
        else:
            args, kw = T.make_case(seed, d), T.scenario(seed)[0]
        try:
            exp = T.run_oracle(args, d, check=False)[0]
        except Exception as e:   # noqa: BLE001
            return seed, args, kw, None, e
        return seed, args, kw, exp, None

    ctx = capi.Context(0, inflight)
    for c0 in range(first, first + n, chunk):
        seeds = range(c0, min(c0 + chunk, first + n))
        with cf.ThreadPoolExecutor(max_workers=max(jobs, 1)) as pool:
            cases = list(pool.map(prepare, seeds))
        live = []
        for seed, args, kw, exp, err in cases:
            if err is not None:
                bad += 1
                print("seed %d FAILED (oracle): %s" % (seed, str(err)[:200]), flush=True)
            elif exp is None:
                crashed += 1
            else:
                live.append((seed, args, kw, exp))
        prepared = []
        for seed, args, kw, exp in live:
            try:
                pa = cli.parse_cmd_line(args)
                prepared.append((pa, cli.load_regions(pa)))
            except Exception as e:   # noqa: BLE001 - an ingest failure yields no contig, like the reference's empty <roi>.fa
                prepared.append(stage5.RegionFailure("seed%d" % seed, str(e)))
        errors = []
        texts, _ = stage5.run_regions(ctx, prepared, inflight, None, errors)
        for (seed, args, kw, exp), got in zip(live, texts):
            done += 1
            if got != exp:
                bad += 1
                print("seed %d FAILED: FASTA differs (%d vs %d contigs) | %r" % (seed, got.count(">"), exp.count(">"), kw), flush=True)
        print("seeds %d..%d: %d compared so far, %d failures, %.0f s" % (seeds[0], seeds[-1], done, bad, time.time() - t0), flush=True)
    ctx.close()
    print("sweep done: %d scenarios (%d in flight), %d compared, %d failures, %d without a defined reference output, %.0f s" % (
        n, inflight, done, bad, crashed, time.time() - t0), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
