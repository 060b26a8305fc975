#!/bin/bash
# rocprofv3 kernel statistics of 224 regions in flight on the final build (executors watch the stamps: no level server).
set -o pipefail
repo=$(pwd)
out=$repo/gpurun_out/r03p
mkdir -p $out/summary
cd /tmp && export TMPDIR=/tmp
export SC_PROBE_ROUNDS=1 SC_PROBE_DISTINCT=25
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/resident_stats -- python3 $repo/tools/inflight_probe.py 224 > $out/resident_stats.log 2>&1 || { echo "failed"; tail -n 5 $out/resident_stats.log; exit 1; }
grep -h '^{' $out/resident_stats.log | tail -n 1 > $out/summary/resident_stats.json
f=$(find $out/resident_stats -name '*_kernel_stats.csv' | tail -n 1); cp $f $out/summary/resident_kernel_stats.csv
rm -rf $out/resident_stats
head -n 4 $out/summary/resident_kernel_stats.csv | cut -c1-160; cut -c1-400 $out/summary/resident_stats.json
