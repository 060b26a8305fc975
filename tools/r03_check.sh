#!/bin/bash
# Quick parity pass + the 224-in-flight probe (after host-side changes).
out=gpurun_out/r03v
mkdir -p $out
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_region_parity or golden or tie or config2 or wide_sampler or mixed or iupac or ambiguity or resident or in_flight" > $out/pytest.txt 2>&1 || { echo "tests failed rc=$?"; tail -n 30 $out/pytest.txt | cut -c1-300; exit 1; }
tail -n 2 $out/pytest.txt
for sw in 0 56 112; do
  echo "== SC_SETUP_WORKERS=$sw"
  SC_SETUP_WORKERS=$sw SC_PROBE_ROUNDS=3 timeout -k 10 200 python3 tools/inflight_probe.py 224 > $out/probe_$sw.txt 2> $out/probe_$sw.err || { echo "probe failed rc=$?"; tail -n 20 $out/probe_$sw.err; exit 1; }
  cut -c1-1200 $out/probe_$sw.txt
done
