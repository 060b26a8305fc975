#!/bin/bash
# configs[3] without thinning: host graph phases (experiment build with -DSC_GRAPH_TIMING) + the wide-class test.
out=gpurun_out/r03d
mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wide_classes or thread_kernels or config4 or deep" > $out/pytest.txt 2>&1 || { echo "tests failed rc=$?"; grep -v "phase\|thread_" $out/pytest.txt | tail -n 40 | cut -c1-300; exit 1; }
grep -v "phase\|thread_" $out/pytest.txt | tail -n 3
timeout -k 10 900 python3 tools/unthinned_probe.py 100000 /tmp/unthinned > $out/probe.txt 2> $out/probe.err || { echo "probe failed rc=$?"; tail -n 20 $out/probe.err; exit 1; }
grep -E "phase|thread_|run |dataset" $out/probe.err $out/probe.txt | cut -c1-300 | tail -50
grep -o "graph_ms[^,]*\|cluster_ms[^,]*\|setup_ms[^,]*" $out/probe.txt | head
tail -n 4 $out/probe.txt | cut -c1-600
