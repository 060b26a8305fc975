#!/usr/bin/env python3
"""Generates tests/golden/ from the REFERENCE itself (build container only).

For every case: the argv (tests/sc_testlib.make_case(seed) -> seeded synthetic
FASTA + SAM), then stdout FASTA, `-G` graph dump and the 17-digit per-level trace
of oracle/_ref/StrainCall_ref (the reference sources compiled by oracle/Makefile),
cross-checked byte for byte against the shipped binary
/root/reference/StrainCall/StrainCall for FASTA and graph.  The MSA vectors come
from oracle/_ref/msa_ref (the reference's MultipleSequenceAlignmentSP::align).
Only inputs' digests and expected outputs are stored -- no reference source.

usage: python tests/golden/make_golden.py [--seeds 0-11]
"""
import argparse
import gzip
import hashlib
import json
import os
import random
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sc_testlib as T  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "StrainCall_ref")
MSA_REF = os.path.join(ROOT, "oracle", "_ref", "msa_ref")
SHIPPED = "/root/reference/StrainCall/StrainCall"


def run(exe, args, cwd, trace=False, graph=False):
    env = dict(os.environ)
    env["PATH"] = T.TOOLS + os.pathsep + env.get("PATH", "")
    env["TMPDIR"] = cwd
    if trace:
        env["SC_TRACE"] = "1"
        env["SC_TRACE_PREC"] = "17"
    p = subprocess.run([exe] + (["-G"] if graph else []) + args, cwd=cwd, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE)
    assert p.returncode == 0
    return p.stdout, p.stderr


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="0-11")
    a = ap.parse_args()
    lo, hi = a.seeds.split("-")
    index = {}
    for seed in range(int(lo), int(hi) + 1):
        with tempfile.TemporaryDirectory() as d:
            args = T.make_case(seed, d)
            fa, sam = args[-2], args[-1]
            rel = args[:-2] + [os.path.basename(fa), os.path.basename(sam)]
            out_fa, out_tr = run(REF, rel, d, trace=True)
            out_g, _ = run(REF, rel, d, graph=True)
            ship_fa, _ = run(SHIPPED, rel, d)
            ship_g, _ = run(SHIPPED, rel, d, graph=True)
            assert ship_fa == out_fa and ship_g == out_g, "rebuilt reference differs from the shipped binary"
            cdir = os.path.join(HERE, "case%02d" % seed)
            os.makedirs(cdir, exist_ok=True)
            open(os.path.join(cdir, "expected.fa"), "wb").write(out_fa)
            with gzip.GzipFile(os.path.join(cdir, "expected.graph.gz"), "wb", mtime=0) as f:
                f.write(out_g)
            with gzip.GzipFile(os.path.join(cdir, "expected.trace.gz"), "wb", mtime=0) as f:
                f.write(out_tr)
            index["case%02d" % seed] = dict(seed=seed, argv=rel[:-2], fasta_sha256=sha(fa), sam_sha256=sha(sam),
                                            contigs=out_fa.count(b">"), graph_lines=out_g.count(b"\n"),
                                            trace_lines=out_tr.count(b"\n"), checked_against_shipped_binary=True)
            print("case%02d" % seed, index["case%02d" % seed]["contigs"], "contigs", flush=True)
    json.dump(index, open(os.path.join(HERE, "index.json"), "w"), indent=1, sort_keys=True)

    # MSA vectors (row a7): sequences in the order the caller passes them (length descending)
    rng = random.Random(11)
    cases = []
    for it in range(120):
        n = rng.randint(2, 24)
        alpha = "ACGT" if it % 4 else "ACGTacgtN-"
        seqs = ["".join(rng.choice(alpha) for _ in range(rng.randint(1, 12))) for _ in range(n)]
        seqs.sort(key=len, reverse=True)
        p = subprocess.run([MSA_REF], input=("\n".join(seqs) + "\n").encode(), stdout=subprocess.PIPE)
        lines = p.stdout.decode().splitlines()
        cases.append(dict(seqs=seqs, rows=lines[:n], ncol=int(lines[n].split()[1])))
    with gzip.GzipFile(os.path.join(HERE, "msa_vectors.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(cases).encode())
    print("msa vectors", len(cases))


if __name__ == "__main__":
    main()
