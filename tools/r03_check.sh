#!/bin/bash
# Grid levels: parity of the deep fixtures, then the unthinned region's time.
out=gpurun_out/r03v
mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config4 or grid or deep or unthinned or wide_sampler or mixed" > $out/pytest.txt 2>&1 || { echo "tests failed rc=$?"; tail -n 40 $out/pytest.txt | cut -c1-300; exit 1; }
tail -n 2 $out/pytest.txt
timeout -k 10 600 python3 tools/unthinned_probe.py 100000 /tmp/unthinned > $out/deep.txt 2> $out/deep.err || { echo "probe failed"; tail $out/deep.err; exit 1; }
grep -E "^run |runs_agree" $out/deep.txt | cut -c1-200; grep -o "graph_ms[^,]*\|cluster_ms[^,]*" $out/deep.txt | head -4
