#!/bin/bash
# Quick parity pass + the steady state at 224 in flight (after host-side changes).
out=gpurun_out/r03v
mkdir -p $out
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_region_parity or golden or tie or config2 or wide_sampler or mixed or mailboxes" > $out/pytest.txt 2>&1 || { echo "tests failed rc=$?"; tail -n 30 $out/pytest.txt | cut -c1-300; exit 1; }
tail -n 2 $out/pytest.txt
SC_PROBE_ROUNDS=8 timeout -k 10 300 python3 tools/inflight_probe.py 224 > $out/probe_ab.txt 2> $out/probe_ab.err || { echo "probe failed rc=$?"; tail -n 20 $out/probe_ab.err; exit 1; }
python3 - <<'PY'
import json
r = json.loads(open("gpurun_out/r03v/probe_ab.txt").read().strip().splitlines()[-1])
print({k: r[k] for k in ("seconds", "reads_per_s", "cu_busy_frac", "cluster_ms", "graph_ms", "place_ms", "host_us_per_level", "wake_us_per_level", "cpu_cores_used", "nr_throttled")})
PY
