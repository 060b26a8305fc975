#!/bin/bash
# Parity sweep with 224 regions in flight on the final code.
out=gpurun_out/r03t
mkdir -p $out
timeout -k 10 500 python3 tools/parity_sweep_inflight.py 120000 2500 --inflight 224 --jobs 14 --chunk 1250 > $out/sweep_final224.txt 2>&1; tail -n 1 $out/sweep_final224.txt
