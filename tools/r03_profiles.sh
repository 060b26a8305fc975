#!/bin/bash
# Round-3 profiles (run on the GPU box from the repo root): rocprofv3 kernel statistics and the two PMC passes of
#   bench    python3 bench.py --steps 2 --warmup 1 --no-cpu --no-set --no-saturation --no-depth   (configs[1], one region, a launch per level)
#   set      SC_RESIDENT=0 python3 tools/inflight_probe.py 100                                      (100 regions in flight, a launch per level: per-level kernel spread)
#   resident python3 tools/inflight_probe.py 224                                                    (224 regions in flight on resident workers: one dispatch per generation)
#   depth    python3 tools/depth_bench.py 1e8 20 3                                                  (stage 1, k_depth_fused)
# The program itself follows `--` (no env / bash -c hop); counters in passes of their own.
set -o pipefail
repo=$(pwd)
out=$repo/gpurun_out/r03p
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
pass() {   # name, kind (stats | FETCH_SIZE | WRITE_SIZE), command...
    name=$1; kind=$2; shift 2
    if [ $kind = stats ]; then opts="--kernel-trace --stats"; else opts="--pmc $kind --kernel-trace"; fi
    timeout -k 10 400 rocprofv3 $opts --output-format csv -d $out/${name}_$kind -- "$@" > $out/${name}_$kind.log 2>&1 || { echo "$name $kind failed"; tail -n 5 $out/${name}_$kind.log; return 1; }
    grep -h '^{' $out/${name}_$kind.log | tail -n 1 > $out/${name}_$kind.json
    echo "$name $kind ok"
}
B="python3 $repo/bench.py --no-cpu --no-set --no-saturation --no-depth"
pass bench stats $B --steps 2 --warmup 1 || exit 1
pass bench FETCH_SIZE $B --steps 1 --warmup 0 || exit 1
pass bench WRITE_SIZE $B --steps 1 --warmup 0 || exit 1
D="python3 $repo/tools/depth_bench.py 1e8 20 3"
pass depth stats $D || exit 1
pass depth FETCH_SIZE $D || exit 1
pass depth WRITE_SIZE $D || exit 1
export SC_PROBE_ROUNDS=1 SC_PROBE_DISTINCT=25
SC_RESIDENT=0 pass set stats python3 $repo/tools/inflight_probe.py 100 || exit 1
SC_RESIDENT=1 pass resident stats python3 $repo/tools/inflight_probe.py 224 || exit 1
# configs[3] without thinning (a million reads, 59 000 per class): the kernels of the set-up, k_thread_sort_big among them
pass deep stats python3 $repo/tools/unthinned_probe.py 100000 /tmp/unthinned_prof || echo "deep pass skipped"
cd $repo
python3 tools/summarize_profiles.py $out/bench_stats $out/bench_FETCH_SIZE $out/bench_WRITE_SIZE $out/summary bench_config2 > $out/summary_bench.txt
python3 tools/summarize_profiles.py $out/depth_stats $out/depth_FETCH_SIZE $out/depth_WRITE_SIZE $out/summary depth_1e8 > $out/summary_depth.txt
for n in set resident deep; do f=$(find $out/${n}_stats -name '*_kernel_stats.csv' 2>/dev/null | tail -n 1); [ -n "$f" ] && cp $f $out/summary/${n}_kernel_stats.csv; done
grep -h "^run \|^{" $out/deep_stats.log > $out/summary/deep_probe.txt 2>/dev/null
cp $out/*.json $out/summary/ 2>/dev/null
# raw traces are large: keep the summaries only
rm -rf $out/bench_stats $out/bench_FETCH_SIZE $out/bench_WRITE_SIZE $out/depth_stats $out/depth_FETCH_SIZE $out/depth_WRITE_SIZE $out/set_stats $out/resident_stats $out/deep_stats
ls -la $out/summary
