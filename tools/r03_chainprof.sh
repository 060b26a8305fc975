#!/bin/bash
# Experiment build (-DSC_CHAIN_PROF): cycles of a chain pass by section, one region.
out=gpurun_out/r03p
mkdir -p $out
python3 - <<'PY' > $out/prof.txt 2> $out/prof.err || { echo failed; tail $out/prof.err; exit 1; }
import ctypes, os, sys
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import inflight_probe as ip
from rambl_amd import capi, stage5
base = ip.prepare(1, 10000)
ctx = capi.Context(0, 1)
params = capi.default_params(0.01, 0.02, 0.02)
stage5.run_regions(ctx, base * 2, 1, params)
out = (ctypes.c_ulonglong * 12)()
rc = capi.lib().sc_debug_chain_prof(out)
v = list(out)
n = max(v[5], 1)
print("rc", rc, "passes", v[5], "cycles per pass by section:", [round(x / n, 1) for x in v[:5]], "sum", round(sum(v[:5]) / n, 1))
ctx.close()
PY
cat $out/prof.txt
