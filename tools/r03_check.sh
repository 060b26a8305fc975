#!/bin/bash
# Seed 71150 of the in-flight sweep: alone, and among its neighbours with and without the level server.
out=gpurun_out/r03v
mkdir -p $out
python3 - <<'PY' 2>&1 | tail -n 30
import os, sys, tempfile
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import sc_testlib as T
from rambl_amd import capi, cli, stage5
seed = 71150
d = tempfile.mkdtemp(prefix="one_")
args = T.make_case(seed, d)
exp = T.run_oracle(args, d, check=False)[0]
got = T.run_product(args)
print("alone (a launch per level): equal =", got == exp, len(got), len(exp))
if got != exp:
    g, e = got.splitlines(), exp.splitlines()
    for i, (a, b) in enumerate(zip(g, e)):
        if a != b:
            print("first difference at line", i, a[:80], "|", b[:80]); break
def inflight(poll, n=224, reps=3):
    os.environ["SC_POLL_EXEC"] = poll
    seeds = list(range(seed - 60, seed + n - 60))
    prepared, exps = [], []
    for s in seeds:
        dd = tempfile.mkdtemp(prefix="nb%d_" % s)
        a = T.make_case(s, dd)
        pa = cli.parse_cmd_line(a)
        try:
            prepared.append((pa, cli.load_regions(pa)))
        except Exception as ex:
            prepared.append(stage5.RegionFailure("seed%d" % s, str(ex)))
        exps.append(T.run_oracle(a, dd, check=False)[0] if s == seed else None)
    for rep in range(reps):
        with capi.Context(0, n) as ctx:
            texts, _ = stage5.run_regions(ctx, prepared, n, None, [])
        i = seeds.index(seed)
        print("poll", poll, "rep", rep, "in flight: equal to oracle =", texts[i] == exps[i], "equal to alone =", texts[i] == got)
inflight("1"); inflight("0")
PY
