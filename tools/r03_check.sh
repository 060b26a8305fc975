#!/bin/bash
# What the weight rows in HBM / L2 instead of LDS cost the chain (one region, a launch per level).
out=gpurun_out/r03v
mkdir -p $out
for v in 1 0; do
  echo "== SC_ROWS_LDS=$v"
  SC_ROWS_LDS=$v SC_PROBE_ROUNDS=3 timeout -k 10 200 python3 tools/inflight_probe.py 1 > $out/rows_$v.txt 2> $out/rows_$v.err || { echo failed; tail $out/rows_$v.err; exit 1; }
  python3 - <<PY
import json
r = json.loads(open("gpurun_out/r03v/rows_$v.txt").read().strip().splitlines()[-1])
print({k: r[k] for k in ("reads_per_s", "cluster_ms", "level_kernel_ms", "chain_ms", "chain_mcycles")})
PY
done
