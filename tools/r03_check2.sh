#!/bin/bash
# A/B on one box: threads and host memory next to the GPU (SC_NUMA_BIND: the library's own; SC_PROBE_BIND: the whole process).
out=gpurun_out/r03w
mkdir -p $out
for cfg in "0 0" "1 0" "1 1" "0 0" "1 0" "1 1" "0 0" "1 1"; do
  set -- $cfg
  echo "== SC_NUMA_BIND=$1 SC_PROBE_BIND=$2"
  SC_NUMA_BIND=$1 SC_PROBE_BIND=$2 SC_PROBE_ROUNDS=3 timeout -k 10 200 python3 tools/inflight_probe.py 224 > $out/probe_ab.txt 2> $out/probe_ab.err || { echo "probe failed rc=$?"; tail -n 20 $out/probe_ab.err; exit 1; }
  python3 - <<'PY'
import json
r = json.loads(open("gpurun_out/r03w/probe_ab.txt").read().strip().splitlines()[-1])
print({k: r[k] for k in ("reads_per_s", "cu_busy_frac", "cluster_ms", "graph_ms", "place_ms", "level_kernel_ms", "host_us_per_level", "wake_us_per_level", "cpu_cores_used", "nr_throttled")})
PY
done
