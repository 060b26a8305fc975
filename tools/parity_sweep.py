#!/usr/bin/env python3
"""One-off wide parity sweep on a GPU box: product (HIP path through the C-ABI) against the C
oracle (test infrastructure) on many seeded scenarios, FASTA byte-for-byte and per-level
abundance traces to 1e-9.  Usage: python3 tools/parity_sweep.py FIRST_SEED N [--big]"""
import os
import random
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sc_testlib as T  # noqa: E402
from rambl_amd import synth  # noqa: E402


def big_case(seed, outdir):
    """Larger regions than the unit tests use: more strains, noisier reads -> many candidate strains."""
    rng = random.Random(seed * 7919 + 3)
    kw = dict(glen=rng.randint(500, 900), n_strains=rng.randint(3, 7), n_reads=rng.randint(1200, 3500),
              rlen=rng.choice([100, 150]), err=rng.choice([0.005, 0.01, 0.02, 0.03]),
              n_sub=rng.randint(6, 20), n_ins=rng.randint(0, 3), n_del=rng.randint(0, 3),
              paired=rng.random() < 0.3, shared_ins_site=rng.random() < 0.3)
    gene = synth.make_gene(seed, name="b%d" % seed, **kw)
    fa, sam = synth.write_dataset(outdir, [gene])
    D = rng.choice([200, 400, 800])
    t = rng.choice([0.02, 0.01, 0.005])
    return ["-r", "%s:1-%d" % (gene["name"], len(gene["ref"])), "-q", "0", "-D", str(D), "-I", "13", "-l", "70",
            "-t", str(t), "-d", "0.02", "-w", "5000", fa, sam], kw


def main():
    first, n = int(sys.argv[1]), int(sys.argv[2])
    big = "--big" in sys.argv
    bad = 0
    t0 = time.time()
    for seed in range(first, first + n):
        d = tempfile.mkdtemp(prefix="sweep%d_" % seed)
        if big:
            args, kw = big_case(seed, d)
        else:
            args, kw = T.make_case(seed, d), T.scenario(seed)[0]
        try:
            exp_fa, exp_tr = T.run_oracle(args, d, trace=True)
            trf = os.path.join(d, "trace.txt")
            got_fa = T.run_product(args, trace_file=trf)
            assert got_fa == exp_fa, "FASTA differs"
            T.compare_traces(open(trf).read(), exp_tr)
            print("seed %d ok (%d contigs) %.0fs" % (seed, got_fa.count(">"), time.time() - t0), flush=True)
        except Exception as e:   # noqa: BLE001
            bad += 1
            print("seed %d FAILED: %s | %r" % (seed, str(e)[:300], kw), flush=True)
    print("sweep done: %d scenarios, %d failures, %.0f s" % (n, bad, time.time() - t0), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
