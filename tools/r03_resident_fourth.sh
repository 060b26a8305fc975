#!/bin/bash
# Resident workers, 224 in flight, after limiting the spinning executors: set-up places and executor counts.
out=gpurun_out/r03g
mkdir -p $out
export SC_RESIDENT=1
SC_PROBE_ROUNDS=3 SC_PROBE_SWEEP="SC_SETUP_LIMIT=16,SC_EXEC_THREADS=15;SC_SETUP_LIMIT=24,SC_EXEC_THREADS=15;SC_SETUP_LIMIT=32,SC_EXEC_THREADS=15;SC_SETUP_LIMIT=24,SC_EXEC_THREADS=24;SC_SETUP_LIMIT=16,SC_EXEC_THREADS=15,SC_EXEC_SPINNERS=0;SC_SETUP_LIMIT=16,SC_EXEC_THREADS=15,SC_EXEC_SPINNERS=4" timeout -k 10 500 python3 tools/inflight_probe.py 224 > $out/sweep224.txt 2> $out/sweep224.err || { echo "sweep failed"; tail -n 5 $out/sweep224.err; cat $out/sweep224.txt; exit 1; }
cat $out/sweep224.txt
SC_LEVEL_LOG=$out/lvl SC_SETUP_LIMIT=16 SC_PROBE_ROUNDS=2 timeout -k 10 200 python3 tools/inflight_probe.py 224 > $out/probe224_log.txt 2> $out/probe224_log.err || { echo "probe with level log failed"; tail -n 5 $out/probe224_log.err; exit 1; }
python3 tools/level_log_summary.py $out/lvl > $out/level_summary.txt 2>&1; cat $out/level_summary.txt; rm -f $out/lvl.*
