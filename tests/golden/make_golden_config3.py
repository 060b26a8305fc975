#!/usr/bin/env python3
"""Generates tests/golden/config3_regions from the REFERENCE itself (oracle/_ref/StrainCall_ref, build
container only): BASELINE.json configs[2] as SURVEY.md section 8(d) defines it -- 100 independent seed
genes (rambl_amd.synth.config3: seeds 100-199, 2 000-10 000 reads each) in ONE FASTA + ONE SAM, every
region run the way scripts/rambl.py:169-201 runs it (`name:1-len` from the .fai, rambl.py's options).

Stored: the digests of the generated inputs, the argv options, and the reference's stdout per region
(expected.json.gz: roi -> FASTA text, in .fai order).  Each region is one reference process (7-10 minutes,
-O2 build), run in its own working directory (the reference's temp files collide otherwise).  Finished
regions are cached under tests/golden/_config3_work/ so the run can be resumed.

usage: python tests/golden/make_golden_config3.py JOBS [FIRST_N_REGIONS]
"""
import gzip
import hashlib
import json
import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sc_testlib as T  # noqa: E402
from rambl_amd import stage5, synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "StrainCall_ref")
WORK = os.path.join(HERE, "_config3_work")
OUT = os.path.join(HERE, "config3_regions")


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def one(job):
    roi, fa, sam = job
    dst = os.path.join(WORK, "out", roi.replace(":", "_") + ".json")
    if os.path.exists(dst):
        return roi, json.load(open(dst))
    cwd = os.path.join(WORK, "cwd", roi.replace(":", "_"))
    os.makedirs(cwd, exist_ok=True)
    env = dict(os.environ)
    env["PATH"] = T.TOOLS + os.pathsep + env.get("PATH", "")
    env["TMPDIR"] = cwd
    t0 = time.time()
    p = subprocess.run([REF] + stage5.straincall_argv(roi, fa, sam), cwd=cwd, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.DEVNULL)
    rec = dict(rc=p.returncode, fasta=p.stdout.decode("ascii"), seconds=int(time.time() - t0))
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    json.dump(rec, open(dst + ".tmp", "w"))
    os.replace(dst + ".tmp", dst)
    return roi, rec


def main():
    jobs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    first_n = int(sys.argv[2]) if len(sys.argv) > 2 else None
    data = os.path.join(WORK, "data")
    fa, sam = os.path.join(data, "seed_otus.fasta"), os.path.join(data, "reads.sam")
    if not (os.path.exists(fa) and os.path.exists(sam)):
        synth.config3(data)
    rois = stage5.roi_list(fa + ".fai")
    if first_n:
        rois = rois[:first_n]
    t0 = time.time()
    results = {}
    with ThreadPoolExecutor(jobs) as ex:
        for roi, rec in ex.map(one, [(r, fa, sam) for r in rois]):
            results[roi] = rec
            print("%s rc=%d contigs=%d %ds (elapsed %ds)" % (roi, rec["rc"], rec["fasta"].count(">"), rec["seconds"],
                                                           time.time() - t0), flush=True)
    if first_n:
        return
    os.makedirs(OUT, exist_ok=True)
    with gzip.GzipFile(os.path.join(OUT, "expected.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps([[r, results[r]["fasta"]] for r in rois]).encode("ascii"))
    n_aln = sum(1 for ln in open(sam) if not ln.startswith("@"))
    json.dump(dict(generator="rambl_amd.synth.config3(outdir)  (seeds 100-199, 2 000-10 000 reads each, one FASTA + one SAM)",
                   options=stage5.straincall_argv("ROI", "FASTA", "SAM")[2:-2], regions=len(rois), alignments=n_aln,
                   fasta_sha256=sha(fa), sam_sha256=sha(sam), contigs=sum(results[r]["fasta"].count(">") for r in rois),
                   nonzero_exit=[r for r in rois if results[r]["rc"] != 0],
                   note="stdout of the reference itself per region, scripts/rambl.py:169-201 order",
                   reference_build="oracle/_ref/StrainCall_ref (-O2, s=0)",
                   reference_cpu_seconds_build_container=sum(results[r]["seconds"] for r in rois)),
              open(os.path.join(OUT, "meta.json"), "w"), indent=1, sort_keys=True)
    print("config3_regions: %d regions, %d contigs" % (len(rois), sum(results[r]["fasta"].count(">") for r in rois)))


if __name__ == "__main__":
    main()
