#!/bin/bash
# First runs of the resident level workers on the GPU box: smallest first, each under its own timeout, stop at the first failure.
set -o pipefail
out=gpurun_out/r03c
mkdir -p $out
export SC_RESIDENT=1
timeout -k 10 180 python3 __graft_entry__.py smoke > $out/smoke.txt 2>&1 || { echo "smoke failed rc=$?"; tail -20 $out/smoke.txt; exit 1; }
tail -1 $out/smoke.txt
timeout -k 10 420 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_region_parity or golden or tie or in_flight or stage5_many or edge_support or iupac" > $out/pytest_small.txt 2>&1 || { echo "small tests failed rc=$?"; tail -30 $out/pytest_small.txt; exit 1; }
tail -2 $out/pytest_small.txt
timeout -k 10 420 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mixed or config2 or config4_deep or config3" > $out/pytest_big.txt 2>&1 || { echo "big tests failed rc=$?"; tail -30 $out/pytest_big.txt; exit 1; }
tail -2 $out/pytest_big.txt
SC_PROBE_ROUNDS=1 timeout -k 10 200 python3 tools/inflight_probe.py 1 16 > $out/probe_res_small.txt 2> $out/probe_res_small.err || { echo "probe small failed"; tail -5 $out/probe_res_small.err; exit 1; }
cat $out/probe_res_small.txt
SC_PROBE_ROUNDS=3 timeout -k 10 300 python3 tools/inflight_probe.py 64 128 232 > $out/probe_res.txt 2> $out/probe_res.err || { echo "probe failed"; tail -5 $out/probe_res.err; exit 1; }
cat $out/probe_res.txt
SC_RESIDENT=0 SC_PROBE_ROUNDS=3 timeout -k 10 300 python3 tools/inflight_probe.py 1 128 256 > $out/probe_launch.txt 2> $out/probe_launch.err || { echo "probe launch failed"; tail -5 $out/probe_launch.err; exit 1; }
cat $out/probe_launch.txt
