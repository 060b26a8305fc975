#!/bin/bash
# Phases of a level of a single region (a launch per level): where the time outside the chain goes.
out=gpurun_out/r03l
mkdir -p $out; rm -f $out/lv.*
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_region_parity or golden or tie or config2 or wide_sampler or mixed or iupac or ambiguity or resident or in_flight or grid or deep" > $out/pytest.txt 2>&1 || { echo "tests failed rc=$?"; tail -n 30 $out/pytest.txt | cut -c1-300; exit 1; }
tail -n 2 $out/pytest.txt
SC_LEVEL_LOG=$out/lv SC_PROBE_ROUNDS=4 timeout -k 10 200 python3 tools/inflight_probe.py 1 > $out/probe.txt 2> $out/probe.err || { echo "probe failed rc=$?"; tail -n 20 $out/probe.err; exit 1; }
cut -c1-900 $out/probe.txt
python3 tools/level_log_summary.py $out/lv
gzip -f $out/lv.*
