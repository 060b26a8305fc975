#!/bin/bash
# Final build: single-region parity sweep with traces, the precomputed large scenarios (one at a time and in flight).
out=gpurun_out/r03v
mkdir -p $out
timeout -k 10 420 python3 tools/parity_sweep.py 60000 600 --jobs 14 > $out/sweep_single.txt 2>&1; tail -n 2 $out/sweep_single.txt
timeout -k 10 300 python3 tools/big_expect_check.py > $out/big_single.txt 2>&1; tail -n 2 $out/big_single.txt
timeout -k 10 300 python3 tools/big_expect_check.py --inflight 64 > $out/big_inflight.txt 2>&1; tail -n 2 $out/big_inflight.txt
