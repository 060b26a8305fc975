#!/bin/bash
# Parity sweeps with many regions in flight on resident workers (the final build: executors watch the stamps).
out=gpurun_out/r03t
mkdir -p $out
timeout -k 10 500 python3 tools/parity_sweep_inflight.py 70000 2500 --inflight 224 --jobs 14 --chunk 1250 > $out/sweep_inflight224.txt 2>&1; tail -n 1 $out/sweep_inflight224.txt
timeout -k 10 600 python3 tools/parity_sweep_inflight.py 90000 3000 --inflight 224 --jobs 14 --chunk 1500 > $out/sweep_inflight224b.txt 2>&1; tail -n 1 $out/sweep_inflight224b.txt
