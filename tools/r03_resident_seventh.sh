#!/bin/bash
out=gpurun_out/r03j
mkdir -p $out
timeout -k 10 200 python3 tools/wide_trace_dump.py $out/product_trace_300_350.txt.gz 300 350 > $out/wide.txt 2>&1; tail -n 2 $out/wide.txt
SC_PROBE_ROUNDS=3 SC_PROBE_SWEEP="SC_X=0;SC_EXEC_THREADS=24,SC_EXEC_LONG=16;SC_EXEC_THREADS=20,SC_EXEC_LONG=12,SC_EXEC_SPINNERS=8;SC_EXEC_LONG=0,SC_SETUP_LIMIT=24" timeout -k 10 400 python3 tools/inflight_probe.py 224 > $out/sweep224.txt 2> $out/sweep224.err || { echo "sweep failed"; tail -n 5 $out/sweep224.err; cat $out/sweep224.txt; exit 1; }
cat $out/sweep224.txt
SC_PROBE_ROUNDS=3 timeout -k 10 300 python3 tools/inflight_probe.py 1 64 128 > $out/curve.txt 2> $out/curve.err; cat $out/curve.txt
