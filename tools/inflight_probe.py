#!/usr/bin/env python3
"""Where does the time go with R regions in flight on one GPU?  For each R: wall time of R regions of the
configs[1] shape (seeds 21..), and per region the host-side level-walk time against the time spent INSIDE
the level kernels (sc_stats.level_kernel_ticks, 100 MHz ticks from kernel start to completion stamp) and
inside the urn chains.  usage: python3 tools/inflight_probe.py [R ...]   (env SC_PROBE_READS, SC_PROBE_PROCS)"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def prepare(n, reads):
    from rambl_amd import cli, stage5, synth
    d = tempfile.mkdtemp(prefix="probe_")
    out = []
    for k in range(n):
        g = synth.make_gene(21 + k, glen=1500, n_strains=3, n_reads=reads, name="gene%d" % (21 + k))
        fa, sam = synth.write_dataset(os.path.join(d, "r%d" % k), [g])
        pa = cli.parse_cmd_line(stage5.straincall_argv("%s:1-1500" % g["name"], fa, sam))
        out.append((pa, cli.load_regions(pa)))
    return out


def multi(procs, r):
    """`procs` processes with r regions in flight each, started together (children wait for a common time)."""
    import subprocess
    t_start = time.time() + float(os.environ.get("SC_PROBE_LEAD", "45"))
    env = dict(os.environ, SC_PROBE_START=repr(t_start), SC_PROBE_PROCS="1")
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(r)], env=env, stdout=subprocess.PIPE) for _ in range(procs)]
    outs = [json.loads(p.communicate()[0].decode().strip().splitlines()[-1]) for p in ps]
    t_end = max(o["t_end"] for o in outs)
    reads = int(os.environ.get("SC_PROBE_READS", "10000"))
    print(json.dumps(dict(procs=procs, regions_per_proc=r, seconds=round(t_end - t_start, 3),
                          reads_per_s=round(procs * r * reads / (t_end - t_start)),
                          level_kernel_ms=[o["level_kernel_ms"] for o in outs], cluster_ms=[o["cluster_ms"] for o in outs])), flush=True)


def main():
    procs = int(os.environ.get("SC_PROBE_PROCS", "1"))
    if procs > 1:
        for r in [int(x) for x in sys.argv[1:]]:
            multi(procs, r)
        return
    rs = [int(x) for x in sys.argv[1:]] or [1, 4, 16, 32]
    reads = int(os.environ.get("SC_PROBE_READS", "10000"))
    os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("SC_PROBE_QUEUES", "24"))
    from rambl_amd import capi, stage5
    if os.environ.get("SC_PROBE_BIND") == "1":
        print("bound to %d CPUs next to the GPU" % capi.host_bind(0), file=sys.stderr)
    # distinct data sets (seeds 21..), reused cyclically; SC_PROBE_ROUNDS x R regions pass through the context with R in
    # flight, so that the ramp at the start and the drain at the end weigh 1 / rounds (the steady state is what a node sees)
    distinct = int(os.environ.get("SC_PROBE_DISTINCT", "48"))
    rounds = int(os.environ.get("SC_PROBE_ROUNDS", "1"))
    base = prepare(min(max(rs), distinct), reads)
    # SC_PROBE_SWEEP="SC_SETUP_LIMIT=4,SC_EXEC_THREADS=8;SC_SETUP_LIMIT=16": every R once per setting (the library reads
    # these when a context is created)
    sweep = [dict(kv.split("=") for kv in part.split(",") if kv) for part in os.environ.get("SC_PROBE_SWEEP", "").split(";")] or [{}]
    for r, knobs in [(r, k) for r in rs for k in sweep]:
        os.environ.update(knobs)
        prep = [base[i % len(base)] for i in range(r * rounds)]
        ctx = capi.Context(0, r)
        params = capi.default_params(0.01, 0.02, 0.02)
        # warm every slot (device buffers are allocated on a slot's first region: SC_PROBE_WARM_ALL=0 leaves that in the clock)
        stage5.run_regions(ctx, prep[:min(r, 4)] if os.environ.get("SC_PROBE_WARM_ALL") == "0" else prep[:r], r, params)
        if os.environ.get("SC_PROBE_START"):
            while time.time() < float(os.environ["SC_PROBE_START"]):
                time.sleep(0.0005)
        import resource
        ru0 = resource.getrusage(resource.RUSAGE_SELF)
        try:
            cs0 = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        except OSError:
            cs0 = {}
        t0 = time.time()
        _, stats = stage5.run_regions(ctx, prep, r, params)
        t_end = time.time()
        dt = t_end - t0
        ctx.close()
        n = len(stats)
        rec = dict(regions=len(prep), in_flight=r, seconds=round(dt, 3), reads_per_s=round(len(prep) * reads / dt),
                   level_execution=("launch per level" if os.environ.get("SC_RESIDENT", "1" if r > 1 else "0") == "0" else "resident workers"),
                   cluster_ms=round(sum(s["cluster_ms"] for s in stats) / n, 1),
                   graph_ms=round(sum(s["graph_ms"] for s in stats) / n, 1),
                   level_kernel_ms=round(sum(s["level_kernel_ticks"] for s in stats) / n / 1e5, 1),
                   chain_ms=round(sum(s["chain_wall_ticks"] for s in stats) / n / 1e5, 1),
                   chain_mcycles=round(sum(s["chain_cycles"] for s in stats) / n / 1e6, 1),
                   xcd=[sum(s["xcd_levels"][k] for s in stats) for k in range(8)],
                   levels=round(sum(s["level_launches"] for s in stats) / n))
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
        rec["cpu_cores_used"] = round(((ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)) / dt, 2)
        rec["t_end"] = t_end
        rec["setup_ms"] = round(sum(s["setup_ms"] for s in stats) / n, 1)
        rec["queue_ms"] = round(sum(s["queue_ms"] for s in stats) / n, 1)
        rec["place_ms"] = round(sum(s["place_ms"] for s in stats) / n, 1)
        rec["mailbox_ms"] = round(sum(s["mailbox_ms"] for s in stats) / n, 1)
        rec["host_us_per_level"] = [round(sum(s["host_us"][k] for s in stats) / max(sum(s["level_launches"] for s in stats), 1), 1) for k in range(3)]
        rec["wake_us_per_level"] = [round(sum(s["wake_us"][k] for s in stats) / max(sum(s["level_launches"] for s in stats), 1), 1) for k in range(2)]
        try:
            cs1 = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
            rec["throttled_ms"] = round((int(cs1["throttled_usec"]) - int(cs0["throttled_usec"])) / 1e3, 1)
            rec["nr_throttled"] = int(cs1["nr_throttled"]) - int(cs0["nr_throttled"])
        except (OSError, KeyError, ValueError, NameError):
            pass
        rec["knobs"] = {k: os.environ[k] for k in ("SC_EXEC_THREADS", "SC_EXEC_LONG", "SC_EXEC_SPINNERS", "SC_SETUP_LIMIT", "SC_SETUP_WORKERS", "SC_RESIDENT_SLOTS", "SC_NUMA_BIND", "SC_PROBE_BIND") if k in os.environ}
        rec["gap_us_per_level"] = round(1e3 * (rec["cluster_ms"] - rec["setup_ms"] - rec["mailbox_ms"] - rec["level_kernel_ms"]) / max(rec["levels"], 1), 1)
        # share of the GPU's 256 CUs that held a level workgroup, averaged over the run
        rec["cu_busy_frac"] = round(sum(s["level_kernel_ticks"] for s in stats) / 1e5 / 1e3 / (256 * dt), 3)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
