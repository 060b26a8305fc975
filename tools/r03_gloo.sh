#!/bin/bash
out=gpurun_out/r03u
mkdir -p $out
SC_BENCH_BACKEND=gloo timeout -k 10 500 python3 bench.py --gpus 2 --steps 2 --warmup 1 --streams 100 --sat-rounds 2 --sat-distinct 16 > $out/bench_gloo_2ranks_one_gpu.json 2> $out/bench_gloo.err; echo "gloo rc=$?"; cut -c1-900 $out/bench_gloo_2ranks_one_gpu.json; grep -v "hostname of the client" $out/bench_gloo.err | tail -n 3 | cut -c1-300
SC_PROBE_ROUNDS=3 timeout -k 10 300 python3 tools/inflight_probe.py 224 > $out/probe224.txt 2> $out/probe224.err; cat $out/probe224.txt | cut -c1-900
