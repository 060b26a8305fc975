#!/usr/bin/env python3
"""Generates the full-size fixtures tests/golden/config2_full and tests/golden/config4_deep from the
REFERENCE itself (oracle/_ref/StrainCall_ref, build container only; minutes to tens of minutes each).

  config2_full  BASELINE.json configs[1]: 10 000 x 150 bp reads, 1 500 bp gene, 3 strains (seed 21)
  config4_deep  BASELINE.json configs[3] scaled to 100 000 reads: 50 strains, depth 10 000 thinned by
                -D 800 -- many candidate strains per level (the wide sampler variants)

Only the inputs' digests, the argv and the reference's stdout are stored.
  tie_case525   sc_testlib.big_case(525): equal-abundance candidates told apart below fp64 resolution

usage: python tests/golden/make_golden_large.py config2_full|config4_deep|tie_case525
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sc_testlib as T  # noqa: E402
from rambl_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "StrainCall_ref")
OPTS = ["-q", "0", "-D", "800", "-I", "13", "-l", "70", "-t", "0.02", "-d", "0.02", "-w", "5000"]


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def dataset(name, outdir):
    if name == "config2_full":
        fa, sam, _ = synth.config2(outdir)
        return fa, sam, "gene21:1-1500", "rambl_amd.synth.config2(outdir, seed=21, n_reads=10000, glen=1500, n_strains=3)"
    if name == "config4_deep":
        gene = synth.make_gene(4, glen=1500, n_strains=50, n_reads=100000, name="deep4")
        fa, sam = synth.write_dataset(outdir, [gene])
        return fa, sam, "deep4:1-1500", "rambl_amd.synth.make_gene(4, glen=1500, n_strains=50, n_reads=100000, name='deep4') + write_dataset"
    if name == "tie_case525":
        args, _ = T.big_case(525, outdir)
        return args[-2], args[-1], None, "tests/sc_testlib.big_case(525, outdir)", args[:-2]
    raise SystemExit("unknown case " + name)


def main():
    name = sys.argv[1]
    with tempfile.TemporaryDirectory() as d:
        ds = dataset(name, d)
        fa, sam, roi, gen = ds[:4]
        argv = ds[4] if len(ds) > 4 else ["-r", roi] + OPTS
        env = dict(os.environ)
        env["PATH"] = T.TOOLS + os.pathsep + env.get("PATH", "")
        env["TMPDIR"] = d
        t0 = time.time()
        p = subprocess.run([REF] + argv + [fa, sam], cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        assert p.returncode == 0
        out = os.path.join(HERE, name)
        os.makedirs(out, exist_ok=True)
        open(os.path.join(out, "expected.fa"), "wb").write(p.stdout)
        json.dump(dict(argv=argv, fasta_sha256=sha(fa), sam_sha256=sha(sam), generator=gen,
                       note="stdout of the reference itself", reference_build="oracle/_ref/StrainCall_ref (-O2, s=0)",
                       reference_seconds_build_container=int(time.time() - t0)),
                  open(os.path.join(out, "meta.json"), "w"), indent=1, sort_keys=True)
        print(name, p.stdout.count(b">"), "contigs", int(time.time() - t0), "s")


if __name__ == "__main__":
    main()
