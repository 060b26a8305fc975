#!/bin/bash
# Default policy (resident for several regions in flight, a launch per level for one): tests, then 224 in flight with the split pool.
out=gpurun_out/r03i
mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wide_sampler or msa_kernel_on or test_region_parity or golden or mixed or launch_per_level or config2 or tie" > $out/pytest.txt 2>&1 || { echo "tests failed rc=$?"; tail -n 30 $out/pytest.txt; exit 1; }
tail -n 2 $out/pytest.txt
SC_PROBE_ROUNDS=3 SC_PROBE_SWEEP="SC_X=0;SC_EXEC_THREADS=24,SC_EXEC_LONG=16;SC_EXEC_THREADS=20,SC_EXEC_LONG=12,SC_EXEC_SPINNERS=8;SC_EXEC_LONG=0,SC_SETUP_LIMIT=24" timeout -k 10 400 python3 tools/inflight_probe.py 224 > $out/sweep224.txt 2> $out/sweep224.err || { echo "sweep failed"; tail -n 5 $out/sweep224.err; cat $out/sweep224.txt; exit 1; }
cat $out/sweep224.txt
SC_PROBE_ROUNDS=3 timeout -k 10 300 python3 tools/inflight_probe.py 1 64 128 > $out/curve.txt 2> $out/curve.err; cat $out/curve.txt
