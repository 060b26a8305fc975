#!/bin/bash
# Parity sweeps with many regions in flight on resident workers (the final build: executors watch the stamps).
out=gpurun_out/r03t
mkdir -p $out
timeout -k 10 500 python3 tools/parity_sweep_inflight.py 70000 2500 --inflight 224 --jobs 14 --chunk 1250 > $out/sweep_inflight224.txt 2>&1; tail -n 1 $out/sweep_inflight224.txt
timeout -k 10 400 python3 tools/parity_sweep_inflight.py 80000 1500 --params --inflight 96 --jobs 14 --chunk 750 > $out/sweep_params96.txt 2>&1; tail -n 1 $out/sweep_params96.txt
timeout -k 10 300 python3 tools/big_expect_check.py --inflight 100 > $out/big_inflight.txt 2>&1; tail -n 1 $out/big_inflight.txt
