#!/bin/bash
# Parity sweeps with many regions in flight on resident workers, the two-rank rehearsal of bench.py on one GPU.
out=gpurun_out/r03t
mkdir -p $out
timeout -k 10 500 python3 tools/parity_sweep_inflight.py 40000 2500 --inflight 128 --jobs 14 --chunk 1250 > $out/sweep_inflight128.txt 2>&1; tail -n 2 $out/sweep_inflight128.txt
timeout -k 10 400 python3 tools/parity_sweep_inflight.py 50000 1500 --params --inflight 224 --jobs 14 --chunk 750 > $out/sweep_params224.txt 2>&1; tail -n 2 $out/sweep_params224.txt
SC_BENCH_BACKEND=gloo timeout -k 10 500 python3 bench.py --gpus 2 --steps 2 --warmup 1 --streams 100 --sat-rounds 2 --sat-distinct 16 > $out/bench_gloo_2ranks_one_gpu.json 2> $out/bench_gloo.err; echo "gloo rc=$?"; cut -c1-600 $out/bench_gloo_2ranks_one_gpu.json; tail -n 3 $out/bench_gloo.err | cut -c1-300
