#!/bin/bash
# The whole GPU suite (default policy), the smoke entry, the bench's default run, the two-rank rehearsal on one GPU.
out=gpurun_out/r03s
mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; rc=$?; tail -n 3 $out/pytest.txt; [ $rc -ne 0 ] && { tail -n 40 $out/pytest.txt | cut -c1-300; exit 1; }
timeout -k 10 120 python3 __graft_entry__.py smoke 2>&1 | tail -n 1
timeout -k 10 500 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"; cut -c1-300 $out/bench_default.json; tail -n 2 $out/bench_default.err
SC_BENCH_BACKEND=gloo timeout -k 10 500 python3 bench.py --gpus 2 --steps 2 --warmup 1 --streams 100 --sat-rounds 2 --sat-distinct 16 > $out/bench_gloo_2ranks_one_gpu.json 2> $out/bench_gloo.err; echo "gloo rc=$?"; cut -c1-300 $out/bench_gloo_2ranks_one_gpu.json
