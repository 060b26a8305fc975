#!/usr/bin/env python3
"""CPU side of a wide parity check of the LARGE scenarios (sc_testlib.big_case), whose oracle runs take
minutes each: run the oracle here (any machine, several at a time), keep FASTA + a per-level signature
of the trace under tools/_big_expect/ (git-ignored, travels to the GPU box with gpurun), then run
tools/big_expect_check.py on a GPU box.  Usage: python3 tools/big_expect_make.py FIRST_SEED N JOBS"""
import sys, os, json, gzip, tempfile, shutil, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import sc_testlib as T
from concurrent.futures import ThreadPoolExecutor

OUT = os.path.join(ROOT, 'tools', '_big_expect')
os.makedirs(OUT, exist_ok=True)

def signature(trace_text):
    sig = []
    for when, level, rows in T.parse_trace(trace_text):
        sig.append([when[0], level, len(rows), sum(a for _, a in rows if a == a)])
    return sig

def one(seed):
    dst = os.path.join(OUT, "%d.json.gz" % seed)
    if os.path.exists(dst):
        return seed, "cached"
    d = tempfile.mkdtemp(prefix="bx%d_" % seed)
    try:
        args, kw = T.big_case(seed, d)
        t = time.time()
        fa, tr = T.run_oracle(args, d, trace=True, check=False, timeout=3000)
        if fa is None:
            rec = dict(seed=seed, crashed=True)
        else:
            rec = dict(seed=seed, fasta=fa, sig=signature(tr), seconds=time.time() - t)
        with gzip.open(dst, "wt") as f:
            json.dump(rec, f)
        return seed, "ok %.0fs" % (time.time() - t)
    except Exception as e:
        return seed, "ERR %s" % str(e)[:100]
    finally:
        shutil.rmtree(d, ignore_errors=True)

first, n, jobs = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
with ThreadPoolExecutor(jobs) as ex:
    for seed, msg in ex.map(one, range(first, first + n)):
        print(seed, msg, flush=True)
