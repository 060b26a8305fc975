#!/bin/bash
# Quick parity pass + the 224-in-flight probe (after host-side changes).
out=gpurun_out/r03v
mkdir -p $out
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_region_parity or golden or tie or config2 or wide_sampler or mixed or iupac or ambiguity" > $out/pytest.txt 2>&1 || { echo "tests failed rc=$?"; tail -n 30 $out/pytest.txt | cut -c1-300; exit 1; }
tail -n 2 $out/pytest.txt
SC_PROBE_ROUNDS=3 timeout -k 10 300 python3 tools/inflight_probe.py 1 224 > $out/probe.txt 2> $out/probe.err; cat $out/probe.txt | cut -c1-1000
