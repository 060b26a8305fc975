#!/bin/bash
# Parity sweeps with many regions in flight (final code): varied parameters on resident workers, and the launch-per-level server.
out=gpurun_out/r03t
mkdir -p $out
timeout -k 10 500 python3 tools/parity_sweep_inflight.py 100000 2500 --params --inflight 224 --jobs 14 --chunk 1250 > $out/sweep_params224.txt 2>&1; tail -n 1 $out/sweep_params224.txt
SC_RESIDENT=0 timeout -k 10 400 python3 tools/parity_sweep_inflight.py 110000 1500 --inflight 100 --jobs 14 --chunk 750 > $out/sweep_launch100.txt 2>&1; tail -n 1 $out/sweep_launch100.txt
