#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Differential test: the C oracle
(oracle/straincall_oracle) against the reference itself (oracle/_ref/StrainCall_ref,
built from /root/reference by oracle/Makefile) on seeded synthetic data sets.
Compares stdout FASTA, the -G graph dump and the 17-digit per-level trace.
Only runs where oracle/_ref exists (the build container).

usage: difftest.py [--seeds A-B] [--jobs N] [--keep DIR]
"""
import argparse
import os
import random
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ProcessPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from rambl_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "StrainCall_ref")
ORA = os.path.join(ROOT, "oracle", "straincall_oracle")


def scenario(seed):
    """Seed -> (generator kwargs, argv options).  Covers SNP-only, indels, MSA
    sites, paired reads, thinning (-D), partial ROI / windows, long inserts."""
    rng = random.Random(seed * 104729 + 7)
    kind = seed % 8
    kw = dict(glen=rng.randint(260, 520), n_strains=rng.randint(1, 4), n_reads=rng.randint(40, 260),
              rlen=rng.choice([100, 120, 150]), err=rng.choice([0.0, 0.003, 0.01]),
              n_sub=rng.randint(2, 10), n_ins=rng.randint(0, 2), n_del=rng.randint(0, 2))
    opts = dict(q=0, D=800, I=13, l=70, t=0.02, d=0.02, w=5000, roi="full")
    if kind == 1:
        kw.update(shared_ins_site=True, n_strains=rng.randint(2, 4))
    elif kind == 2:
        kw.update(paired=True, n_reads=rng.randint(80, 300))
    elif kind == 3:
        opts.update(D=rng.choice([5, 10, 20]))
    elif kind == 4:
        opts.update(roi="part")
    elif kind == 5:
        kw.update(ins_len=(1, 6), n_ins=2, shared_ins_site=True, err=0.01)
        opts.update(I=rng.choice([5, 13]))
    elif kind == 6:
        opts.update(w=rng.choice([200, 250]), o=rng.choice([50, 100]), l=rng.choice([40, 70]))
    elif kind == 7:
        kw.update(n_reads=rng.randint(300, 600), err=0.02, n_strains=rng.randint(2, 5))
    return kw, opts


def argv_for(gene, opts, fa, sam, rng):
    glen = len(gene["ref"])
    args = []
    if opts["roi"] == "full":
        args += ["-r", "%s:1-%d" % (gene["name"], glen)]
    elif opts["roi"] == "part":
        a = rng.randint(1, glen // 3)
        b = rng.randint(2 * glen // 3, glen)
        args += ["-r", "%s:%d-%d" % (gene["name"], a, b)]
    args += ["-q", str(opts["q"]), "-D", str(opts["D"]), "-I", str(opts["I"]), "-l", str(opts["l"]),
             "-t", str(opts["t"]), "-d", str(opts["d"]), "-w", str(opts["w"])]
    if "o" in opts:
        args += ["-o", str(opts["o"])]
    return args + [fa, sam]


def run(exe, args, cwd, env, trace=False, graph=False):
    e = dict(env)
    if trace:
        e["SC_TRACE"] = "1"
        e["SC_TRACE_PREC"] = "17"
    a = [exe] + (["-G"] if graph else []) + args
    p = subprocess.run(a, cwd=cwd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=3000)
    return p.returncode, p.stdout, p.stderr


def one(seed, keep=None):
    kw, opts = scenario(seed)
    d = tempfile.mkdtemp(prefix="scdiff_%d_" % seed)
    try:
        gene = synth.make_gene(seed, name="g%d" % seed, **kw)
        fa, sam = synth.write_dataset(d, [gene])
        rng = random.Random(seed)
        args = argv_for(gene, opts, os.path.basename(fa), os.path.basename(sam), rng)
        env = dict(os.environ)
        env["PATH"] = HERE + os.pathsep + env.get("PATH", "")
        env["TMPDIR"] = d
        res = {}
        for name, exe in (("ref", REF), ("ora", ORA)):
            wd = os.path.join(d, name)
            os.makedirs(wd)
            for f in os.listdir(d):
                if os.path.isfile(os.path.join(d, f)):
                    os.symlink(os.path.join(d, f), os.path.join(wd, f))
            rc1, fa_out, tr = run(exe, args, wd, env, trace=True)
            rc2, g_out, _ = run(exe, args, wd, env, graph=True)
            res[name] = (rc1, fa_out, tr, rc2, g_out)
        ok_fa = res["ref"][1] == res["ora"][1]
        ok_tr = res["ref"][2] == res["ora"][2]
        ok_g = res["ref"][4] == res["ora"][4]
        ok = ok_fa and ok_tr and ok_g and res["ref"][0] == 0
        msg = "seed %d kind %d: fasta=%s trace=%s graph=%s rc=%s/%s contigs=%d graphlines=%d tracelines=%d %s" % (
            seed, seed % 8, ok_fa, ok_tr, ok_g, res["ref"][0], res["ora"][0], res["ref"][1].count(b">"),
            res["ref"][4].count(b"\n"), res["ref"][2].count(b"\n"), " ".join(args[:-2]))
        if (not ok) and keep:
            dst = os.path.join(keep, "seed%d" % seed)
            shutil.rmtree(dst, ignore_errors=True)
            shutil.copytree(d, dst, symlinks=True)
            for name in res:
                open(os.path.join(dst, name + ".fa"), "wb").write(res[name][1])
                open(os.path.join(dst, name + ".trace"), "wb").write(res[name][2])
                open(os.path.join(dst, name + ".graph"), "wb").write(res[name][4])
            open(os.path.join(dst, "argv.txt"), "w").write(" ".join(args))
        return ok, msg
    finally:
        shutil.rmtree(d, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="0-15")
    ap.add_argument("--jobs", type=int, default=6)
    ap.add_argument("--keep", default=None)
    a = ap.parse_args()
    lo, hi = a.seeds.split("-")
    seeds = list(range(int(lo), int(hi) + 1))
    bad = 0
    with ProcessPoolExecutor(a.jobs) as ex:
        for ok, msg in ex.map(one, seeds, [a.keep] * len(seeds)):
            print(("OK   " if ok else "FAIL ") + msg, flush=True)
            bad += (not ok)
    print("failures: %d / %d" % (bad, len(seeds)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
