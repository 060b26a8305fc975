#!/bin/bash
# Phases of a configs[1] region's set-up (experiment build with -DSC_GRAPH_TIMING), one region at a time.
out=gpurun_out/r03v
mkdir -p $out
SC_PROBE_ROUNDS=6 timeout -k 10 300 python3 tools/inflight_probe.py 1 > $out/probe_g.txt 2> $out/probe_g.err || { echo "probe failed rc=$?"; tail -n 20 $out/probe_g.err; exit 1; }
cut -c1-600 $out/probe_g.txt
grep -E "phase|thread_" $out/probe_g.err | tail -n 22
