mkdir -p gpurun_out/r2M
for cfg in "16 11" "24 16" "32 22" "32 16" "48 30"; do
  set -- $cfg
  SC_PROBE_QUEUES=$1 SC_LAUNCH_STREAMS=$2 timeout -k 10 200 python3 tools/inflight_probe.py 100 > gpurun_out/r2M/probe_q$1_s$2.log 2>&1
  echo "queues $1 streams $2: $(tail -1 gpurun_out/r2M/probe_q$1_s$2.log | cut -c1-330)"
done
