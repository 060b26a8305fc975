#!/bin/bash
# Parity of the compacted host models (small + golden tests, resident and launch mode), then the host time split at 224 in flight.
out=gpurun_out/r03h
mkdir -p $out
SC_RESIDENT=1 timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_region_parity or golden or tie or iupac or ambiguity or config2" > $out/pytest_res.txt 2>&1 || { echo "resident tests failed rc=$?"; tail -n 30 $out/pytest_res.txt; exit 1; }
tail -n 2 $out/pytest_res.txt
SC_RESIDENT=0 timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_region_parity or golden or wide_sampler or msa_kernel_on" > $out/pytest_launch.txt 2>&1 || { echo "launch tests failed rc=$?"; tail -n 30 $out/pytest_launch.txt; }
tail -n 2 $out/pytest_launch.txt
export SC_RESIDENT=1
SC_PROBE_ROUNDS=3 SC_PROBE_SWEEP="SC_SETUP_LIMIT=24,SC_EXEC_THREADS=15,SC_EXEC_SPINNERS=4;SC_SETUP_LIMIT=24,SC_EXEC_THREADS=15,SC_EXEC_SPINNERS=15;SC_SETUP_LIMIT=16,SC_EXEC_THREADS=12,SC_EXEC_SPINNERS=12" timeout -k 10 400 python3 tools/inflight_probe.py 224 > $out/sweep224.txt 2> $out/sweep224.err || { echo "sweep failed"; tail -n 5 $out/sweep224.err; cat $out/sweep224.txt; exit 1; }
cat $out/sweep224.txt
cat /sys/fs/cgroup/cpu.max /sys/fs/cgroup/cpu.stat 2>&1 | head -12; nproc; lscpu | grep -E "Model name|Thread|Core|Socket" | head -6
