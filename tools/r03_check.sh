#!/bin/bash
# Final build: more executors than the default sixteen (steady state at 224 in flight).
out=gpurun_out/r03v
mkdir -p $out
for cfg in "16 8" "17 8" "18 9" "17 9"; do
  set -- $cfg
  echo "== SC_EXEC_THREADS=$1 SC_EXEC_LONG=$2"
  SC_EXEC_THREADS=$1 SC_EXEC_LONG=$2 SC_PROBE_ROUNDS=8 timeout -k 10 300 python3 tools/inflight_probe.py 224 > $out/probe_ab.txt 2> $out/probe_ab.err || { echo "probe failed rc=$?"; tail -n 20 $out/probe_ab.err; exit 1; }
  python3 - <<'PY'
import json
r = json.loads(open("gpurun_out/r03v/probe_ab.txt").read().strip().splitlines()[-1])
print({k: r[k] for k in ("seconds", "reads_per_s", "cu_busy_frac", "cluster_ms", "graph_ms", "place_ms", "host_us_per_level", "wake_us_per_level", "cpu_cores_used", "nr_throttled", "throttled_ms")})
PY
done
