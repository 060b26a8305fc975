#!/usr/bin/env python3
"""Benchmark of the StrainCall hot path on MI355X.

N = 1 (the headline, BASELINE.json configs[1]): a step = one pass of the path (graph build + level walk with its HIP
kernels + read_assign) over one region, 10 000 synthetic 150 bp reads against one 1 500 bp gene (seed 21).
N > 1 (BASELINE.json configs[2], what north_star shards over the GPUs of a node): a step = the 100-seed-gene set
(seeds 100-199, 575 103 alignments) partitioned longest-processing-time-first over the ranks, every rank with many
regions in flight on its GPU, no exchange while regions run, one gather of the FASTA bytes to rank 0 at the end
(RCCL).  Total work is fixed: strong scaling.

`python bench.py --gpus N` starts the N ranks itself (children are spawned before anything touches a GPU, the parent
only relays rank 0's line); under `torch.distributed.run` the ranks come from the environment.  Prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_READ = 320      # L + 16 + 4*ops + L for 150 bp, one-op reads
HBM_PEAK = 8.0e12             # MI355X_MICROARCH.md: HBM3E peak 8 TB/s
RAMBL_OPTS = "-q 0 -D 800 -I 13 -l 70 -t 0.02 -d 0.02 -w 5000"


def cpu_budget():
    """CPUs this process may use (the GPU box hands out a cgroup share of its host)."""
    n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    return max(1, n // max(int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1), 1))       # ranks sharing the host


def cpu_baseline(data_dir, fasta, sam, roi):
    """The reference itself (oracle/_ref, kind "reference") or the C oracle (kind "port") on a bounded interior sample
    of the bench data set: one process on one core, then one process per available core at once -- rambl.py's
    Pool(cores) over regions (scripts/rambl.py:190-194)."""
    tools = os.path.join(ROOT, "oracle", "tools")
    ref = os.path.join(ROOT, "oracle", "_ref", "StrainCall_ref")
    port = os.path.join(ROOT, "oracle", "straincall_oracle")
    if os.path.exists(ref):
        exe, kind = ref, "reference"
    else:
        if not os.path.exists(port):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"], stdout=subprocess.DEVNULL)
        exe, kind = port, "port"
    env = dict(os.environ)
    env["PATH"] = tools + os.pathsep + env.get("PATH", "")
    args = ["-r", roi] + RAMBL_OPTS.split() + [fasta, sam]
    n_reads = len(subprocess.run([os.path.join(tools, "samtools"), "view", sam, "-q", "0", "-F", "1804", roi],
                                 stdout=subprocess.PIPE, env=env).stdout.splitlines())

    def run(k):
        cwd = os.path.join(data_dir, "cpu%d" % k)      # the reference's temp files collide in a shared directory
        os.makedirs(cwd, exist_ok=True)
        e = dict(env, TMPDIR=cwd)
        return subprocess.Popen([exe] + args, cwd=cwd, env=e, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)

    t0 = time.time()
    p = run(0)
    fasta_text = p.communicate()[0].decode()
    dt = time.time() - t0
    cores = cpu_budget()
    t0 = time.time()
    ps = [run(1 + k) for k in range(cores)]
    outs = [q.communicate()[0].decode() for q in ps]
    dt_all = time.time() - t0
    return dict(value=n_reads / dt, unit="reads/s", cores=1, kind=kind, seconds=dt, sample_reads=n_reads,
                sample="%s of the bench data set (reads cropped to the window), -O2 build, rambl.py options" % roi,
                all_cores={"cores": cores, "processes": cores, "value": cores * n_reads / dt_all, "unit": "reads/s", "seconds": dt_all,
                           "same_output": all(o == fasta_text for o in outs),
                           "note": "one reference process per available core on the same sample, as rambl.py's Pool(cores) runs regions"},
                fasta=fasta_text), args


def depth_scan_leg(device, bases=100_000_000, depth=20.0, reps=3):
    """rambl.py stage 1 (coverage_all_samples.py) on a synthetic 10^8-base input, beside the headline (whose sampler is
    latency-bound by construction).  One kernel, k_depth_fused: a wavefront builds the per-base depth of its reference in LDS
    from the reference's aligned runs and reduces it to intervals, so what crosses HBM is 8 bytes per run in and a few
    intervals per reference out.  `algorithmic_bytes` keeps round 2's definition of the stage's work -- 4 bytes per reference
    base (the depth the reference's pipeline prints per base) + 8 bytes per run -- so the figures compare across rounds;
    `hbm_bytes` is what this formulation has to move.  Durations are HIP events inside sc_depth_scan_runs."""
    import ctypes as C
    import numpy as np
    from rambl_amd import capi, stage1
    lib = capi.lib()
    rng = np.random.default_rng(3)
    n_refs = bases // 1500
    ref_len = rng.integers(1400, 1601, n_refs).astype(np.int32)
    n_runs = int(ref_len.sum() * depth / 150)
    run_ref = np.sort(rng.integers(0, n_refs, n_runs)).astype(np.int32)
    start = np.maximum((rng.random(n_runs) * (ref_len[run_ref] - 150)).astype(np.int64) + 1, 1).astype(np.int32)
    end = np.minimum(start + 149, ref_len[run_ref]).astype(np.int32)
    ip = C.POINTER(C.c_int)
    cap = 4 * n_refs
    iv = [np.zeros(cap, dtype=np.int32) for _ in range(3)]
    sm, cn = np.zeros(cap, dtype=np.int64), np.zeros(cap, dtype=np.int32)
    lib.sc_depth_scan_runs.argtypes = [C.c_int, ip, C.c_int, ip, ip, ip, C.c_long, C.c_int, ip, ip, ip, C.POINTER(C.c_long), ip, C.c_int, ip,
                                       C.POINTER(stage1.DepthStats)]
    best = None
    for _ in range(reps):
        n, st = C.c_int(), stage1.DepthStats()
        rc = lib.sc_depth_scan_runs(device, ref_len.ctypes.data_as(ip), n_refs, run_ref.ctypes.data_as(ip), start.ctypes.data_as(ip),
                                    end.ctypes.data_as(ip), n_runs, 10, iv[0].ctypes.data_as(ip), iv[1].ctypes.data_as(ip),
                                    iv[2].ctypes.data_as(ip), sm.ctypes.data_as(C.POINTER(C.c_long)), cn.ctypes.data_as(ip), cap, C.byref(n),
                                    C.byref(st))
        if rc != 0:
            return {"error": rc}
        if best is None or st.kernel_ms < best[0]:
            best = (st.kernel_ms, st.upload_ms, st.cells, st.runs, n.value, st.prepare_ms)
    ok = int(sm[:best[4]].sum()) == int((end - start + 1).sum())
    alg = 4 * best[2] + 8 * best[3]
    hbm = 8 * best[3] + 4 * (n_refs + 1) + 4 * n_refs + 24 * best[4]
    gbs = alg / (best[0] * 1e-3) / 1e9
    return {"kernel": "k_depth_fused<4> (rambl.py stage 1: aligned runs -> per-base depth of a reference in LDS -> merged intervals with mean depth; "
                      "one kernel for the whole stage, k_depth_mark + k_depth_segments of round 2 took 0.76 + 0.09 ms)",
            "workload": "%d reference bases in %d genes, %d aligned runs of 150 bases" % (int(ref_len.sum()), n_refs, n_runs),
            "bound": "hbm", "algorithmic_bytes": alg, "algorithmic_bytes_definition": "4 B per reference base + 8 B per run (round 2's two-kernel formulation)",
            "launch_ms": best[0], "achieved": gbs, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": gbs * 1e9 / HBM_PEAK,
            "hbm_bytes": hbm, "hbm_GBps": hbm / (best[0] * 1e-3) / 1e9, "upload_ms": best[1], "host_prepare_ms": best[5],
            "intervals": best[4], "depth_sums_add_up": ok,
            "note": "the per-base array never reaches HBM: the rate against round 2's byte count can exceed what streaming that array allowed"}


def spawn_ranks(a):
    """`python bench.py --gpus N` outside a launcher: start the N ranks (before this process touches any GPU) and
    relay rank 0's line."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out = procs[0].communicate()[0].decode()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out)
    sys.stdout.flush()
    return max(rcs)


def roofline(all_stats, steps, world, pmc_round="r03"):
    k_ms = sum(s["sampler_kernel_ms"] for s in all_stats)
    k_n = sum(s["sampler_launches"] for s in all_stats)
    k_copies = sum(s["sampler_read_copies"] for s in all_stats)
    draws = sum(s["draws"] for s in all_stats)
    timing = "HIP events around every sampler launch, on the stream it is launched on"
    if not k_ms:
        # resident level workers / many regions in flight: no launch per level, so the duration of a sampler level is the
        # kernel's own 100 MHz wall clock from the start of the level to its completion stamp
        k_ms = sum(s["sampler_level_ticks"] for s in all_stats) / 1e5
        timing = "the level kernels' own wall clock (s_memrealtime), start of the level to its completion stamp"
    avg_ms = k_ms / max(k_n, 1)
    achieved = (ALG_BYTES_PER_READ * k_copies / max(k_n, 1)) / (avg_ms * 1e-3) if k_n and k_ms else 0.0
    traffic, traffic_src = None, None
    for rnd in (pmc_round, "r02"):
        pmc = os.path.join(ROOT, "profiles", rnd, "pmc_summary.json")
        if os.path.exists(pmc) and world == 1:
            # HBM bytes per sampler launch from the committed rocprofv3 --pmc passes of this command
            traffic = json.load(open(pmc)).get("sample_traffic_bytes_per_launch")
            traffic_src = "profiles/%s/pmc_summary.json (separate FETCH_SIZE / WRITE_SIZE passes, corrected as MI355X_MICROARCH.md prescribes)" % rnd
            break
    return {"bound": "hbm", "kernel": "sc::k_level_sample<NB,L> (one sampler level: read log-likelihood update, weight rows, urn chain)",
            "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": traffic,
            "traffic_source": traffic_src, "algorithmic_bytes_per_launch": ALG_BYTES_PER_READ * k_copies / max(k_n, 1),
            "avg_launch_ms": avg_ms, "launch_timing": timing, "launches_per_step": k_n / max(steps, 1),
            "draws_per_s_per_region": draws / (k_ms * 1e-3) if k_ms else 0.0,
            "note": "a latency-bound serial chain working out of LDS: the HBM fraction is tiny by construction (SURVEY.md section 8(d)); "
                    "chain_cycles_per_draw is the rate that matters"}


def breakdown(all_stats, steps):
    draws = sum(s["draws"] for s in all_stats)
    k_n = sum(s["sampler_launches"] for s in all_stats)
    return {"graph_host": sum(s["graph_ms"] for s in all_stats) / steps,
            "level_walk": sum(s["cluster_ms"] for s in all_stats) / steps,
            "level_kernels": sum(s["level_kernel_ticks"] for s in all_stats) / 1e5 / steps,
            "urn_chains": sum(s["chain_wall_ticks"] for s in all_stats) / 1e5 / steps,
            "levels": sum(s["level_launches"] for s in all_stats) / steps,
            "draws": draws / steps,
            "slow_tier_draws": sum(s["slow_draws"] for s in all_stats) / steps,
            "exact_draws": sum(s["exact_draws"] for s in all_stats) / steps,
            "chain_passes": sum(s["chain_passes"] for s in all_stats) / steps,
            "chain_cycles_per_draw": sum(s["chain_cycles"] for s in all_stats) / max(draws, 1),
            "chain_ns_per_draw": 10.0 * sum(s["chain_wall_ticks"] for s in all_stats) / max(draws, 1),
            "avg_candidates_per_sampler_launch": sum(s["sampler_strains"] for s in all_stats) / max(k_n, 1)}


N_CU = 256                    # MI355X_MICROARCH.md: CUs (a level workgroup holds one while it runs)


def make_context(local, streams, resident):
    """A context of `streams` region slots with resident level workers (one workgroup per slot polls a mailbox: no launch
    per level) or with a launch per level (the level server batches the levels of the regions in flight)."""
    from rambl_amd import capi
    old = os.environ.get("SC_RESIDENT")
    os.environ["SC_RESIDENT"] = "1" if resident else "0"
    try:
        return capi.Context(local, streams)
    finally:
        if old is None:
            os.environ.pop("SC_RESIDENT", None)
        else:
            os.environ["SC_RESIDENT"] = old


def shape_regions(dirname, n, seed0, reads=10000, glen=1500, strains=3):
    """n regions of the configs[1] shape (the generator of the headline, seeds seed0, seed0 + 1, ...), ingested."""
    from rambl_amd import cli, stage5, synth
    out = []
    for k in range(n):
        g = synth.make_gene(seed0 + k, glen=glen, n_strains=strains, n_reads=reads, name="gene%d" % (seed0 + k))
        fa, sam = synth.write_dataset(os.path.join(dirname, "s%d" % (seed0 + k)), [g])
        pa = cli.parse_cmd_line(stage5.straincall_argv("%s:1-%d" % (g["name"], glen), fa, sam))
        out.append((pa, cli.load_regions(pa)))
    return out


def run_in_flight(local, base, in_flight, rounds, resident, reads):
    """`rounds` x `in_flight` regions (the data sets of `base`, cyclically) through a context with `in_flight` slots: what a
    GPU sustains with that many regions in flight -- the ramp at the start and the drain at the end weigh 1 / rounds."""
    import resource
    from rambl_amd import stage5
    ctx = make_context(local, in_flight, resident)
    try:
        slots = in_flight
        prep = [base[i % len(base)] for i in range(in_flight * rounds)]
        stage5.run_regions(ctx, prep[:in_flight], in_flight)                 # every slot has had a region (its device buffers exist)
        ru0 = resource.getrusage(resource.RUSAGE_SELF)
        t0 = time.time()
        _, stats = stage5.run_regions(ctx, prep, in_flight)
        dt = time.time() - t0
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
    finally:
        ctx.close()
    busy = sum(s["level_kernel_ticks"] for s in stats) / 1e5 / 1e3 / dt        # level workgroups on the GPU, averaged over the run
    n = max(len(stats), 1)
    return {"in_flight": slots, "regions": len(prep), "seconds": dt, "reads_per_s": len(prep) * reads / dt, "busy_cus": busy,
            "cu_busy_frac": busy / N_CU, "level_kernel_s_per_region": sum(s["level_kernel_ticks"] for s in stats) / 1e5 / 1e3 / n,
            "region_latency_s": sum(s["cluster_ms"] + s["graph_ms"] for s in stats) / 1e3 / n,
            "host_cores_used": ((ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)) / dt}


def scaling_model(n_gpus, work_kernel_s, slowest_alone_s, busy_cus_sat, units, fixed_s=0.0):
    """makespan >= max(slowest region alone, sum of work / (N x saturated rate)): the regions of a set share nothing, a GPU
    runs `busy_cus_sat` level workgroups side by side when saturated, and no region finishes before its own chain of
    dependent levels has been walked."""
    out = {"inputs": {"work_level_kernel_seconds": work_kernel_s, "slowest_region_alone_s": slowest_alone_s,
                      "saturated_busy_cus_per_gpu": busy_cus_sat, "units": units},
           "formula": "makespan >= max(slowest region alone, work / (N x saturated busy CUs))", "by_gpus": {}}
    for n in (1, 2, 4, 8):
        t = max(slowest_alone_s, work_kernel_s / max(n * busy_cus_sat, 1e-9)) + fixed_s
        out["by_gpus"][str(n)] = {"makespan_s": t, "value": units / t, "bound": "latency of the slowest region" if slowest_alone_s >= work_kernel_s / max(n * busy_cus_sat, 1e-9) else "throughput"}
    if n_gpus:
        out["this_run"] = out["by_gpus"].get(str(n_gpus))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10000)
    ap.add_argument("--glen", type=int, default=1500)
    ap.add_argument("--strains", type=int, default=3)
    ap.add_argument("--sample-roi", default="700-860")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-set", action="store_true", help="N = 1: skip the 100-region leg (regions_in_flight)")
    ap.add_argument("--no-depth", action="store_true", help="N = 1: skip the stage-1 depth-scan leg (hbm_bound_kernel)")
    ap.add_argument("--no-saturation", action="store_true", help="skip the saturation leg (regions of the configs[1] shape in flight)")
    ap.add_argument("--launch-mode", action="store_true", help="one launch per level (the level server) instead of resident level workers")
    ap.add_argument("--streams", type=int, default=224, help="regions in flight per GPU (slots of the context; resident workers: at most 224)")
    ap.add_argument("--sat-points", default="64,128,224", help="N = 1: regions in flight of the saturation curve")
    ap.add_argument("--sat-rounds", type=int, default=8, help="regions per slot of a saturation point: the ramp at the start and the drain at the "
                    "end weigh 1 / rounds (3 rounds: 1.76-1.81 M reads/s at 224 in flight; 8: 2.0 M, 12: 2.0 M)")
    ap.add_argument("--sat-distinct", type=int, default=32, help="distinct data sets of the saturation leg (reused cyclically)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a))

    # the library's streams want a hardware queue each; HIP reads this once, when it initialises
    # (rambl_amd/__init__.py sets it too, but torch touches the device first here)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("LOCAL_WORLD_SIZE", str(world))       # ranks of this run share the host: the library sizes its threads by it
    backend = os.environ.get("SC_BENCH_BACKEND", "nccl")      # "gloo" only to rehearse N > 1 on a one-GPU box
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local) if backend == "nccl" else torch.device("cpu")
    # (ranks that share ONE GPU -- the gloo rehearsal of N > 1 on a one-GPU box -- launch their levels: two processes with
    # a resident grid each on one GPU get context-switched against each other, and the switch of workgroups that hold a whole
    # CU's 160 KB of LDS ended in memory faults on ROCm 7.2; one rank per GPU, the real layout, has the GPU to itself)
    resident = not a.launch_mode and not (world > 1 and backend != "nccl")
    mode = "resident level workers (one workgroup per region slot takes its levels from a host-mapped mailbox)" if resident else \
        "one launch per level (level server, batches on shared streams)"

    from rambl_amd import capi, cli, samio, stage5, synth
    # this rank's threads (the ingest threads and the library's own) onto the CPUs next to its GPU, as a launcher would with
    # numactl: a GPU box is a two-socket host (SC_NUMA_BIND=0 leaves placement to the scheduler)
    near_cpus = capi.host_bind(local)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(dt):
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return dt

    def sum_over_ranks(v):
        if world > 1:
            t = torch.tensor([v], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return float(t.item())
        return v

    def region_set(dirname):
        """The configs[2] set: 100 seed genes in one FASTA + one SAM, ingested (rows a1-a4, native) for this rank's shard."""
        fa, sam = os.path.join(dirname, "seed_otus.fasta"), os.path.join(dirname, "reads.sam")
        if rank == 0 and not os.path.exists(sam + ".done"):
            synth.config3(dirname)
            open(sam + ".done", "w").write("ok")
        if world > 1:
            dist.barrier()
        fai = samio.read_fai(fa + ".fai")
        rois = stage5.roi_list(fa + ".fai")
        t0 = time.time()
        mine, aln = stage5.shard_alignments(fai, sam, world, rank)       # a rank keeps the records of its own shard only
        shared = (samio.Fasta(fa), fai, aln)
        prepared = list(stage5.prepared_stream([rois[i] for i in mine], fa, sam, None, max(min(cpu_budget(), 8), 1), shared))
        ingest_s = time.time() - t0
        n_in = sum(r.n_input for _, regs in prepared for _, r in regs)
        n_graph = sum(sum(r.copies) for _, regs in prepared for _, r in regs)
        return rois, mine, prepared, ingest_s, n_in, n_graph, aln.native.records()       # (records of the whole file, kept or not)

    if world > 1:
        # ---- configs[2]: the 100-region set sharded over the ranks (strong scaling) ...
        d = os.path.join(tempfile.gettempdir(), "scbench_set_%s" % os.environ.get("MASTER_PORT", "0"))
        os.makedirs(d, exist_ok=True)
        rois, mine, prepared, ingest_s, n_in, n_graph, n_total = region_set(d)
        streams = min(a.streams, max(len(mine), 1))
        ctx = make_context(local, streams, resident)

        def step():
            texts, stats = stage5.run_regions(ctx, prepared, streams)
            return stage5.gather_fasta(texts, mine, len(rois), dist, dev), stats

        # the slowest region of the set, alone on a GPU: the floor of the set's makespan however many GPUs share it
        # (rank 0 holds it: the longest-processing-time-first partition deals the costliest region first)
        slowest_alone = 0.0
        if prepared:
            big = max(range(len(prepared)), key=lambda i: sum(sum(r.copies) for _, r in prepared[i][1]) if not isinstance(prepared[i], stage5.RegionFailure) else 0)
            stage5.run_regions(ctx, [prepared[big]], 1)
            t1 = time.time()
            stage5.run_regions(ctx, [prepared[big]], 1)
            slowest_alone = time.time() - t1
        slowest_alone = max_over_ranks(slowest_alone)
        for _ in range(a.warmup):
            step()
        fence()
        t0 = time.time()
        all_stats, fasta_out = [], None
        for _ in range(a.steps):
            fasta_out, st = step()
            all_stats += st
        fence()
        dt = max_over_ranks(time.time() - t0)
        ingest_max = max_over_ranks(ingest_s)
        ctx.close()
        work = sum_over_ranks(sum(s["level_kernel_ticks"] for s in all_stats) / 1e5 / 1e3 / a.steps)
        # ---- ... and the same generator at a size that saturates every GPU (weak scaling): slots x rounds regions of the
        # configs[1] shape per rank, seeds continued from rank to rank
        sat = None
        if not a.no_saturation:
            ds = os.path.join(d, "sat%d" % rank)
            os.makedirs(ds, exist_ok=True)
            base = shape_regions(ds, a.sat_distinct, 1000 + rank * a.sat_distinct, a.reads, a.glen, a.strains)
            fence()
            mine_sat = run_in_flight(local, base, a.streams, a.sat_rounds, resident, a.reads)      # (warm pass, then the timed pass)
            fence()
            dts = max_over_ranks(mine_sat["seconds"])
            busy_min = -max_over_ranks(-mine_sat["busy_cus"])
            sat = {"workload": "%d ranks x %d regions of the configs[1] shape (%d distinct data sets per rank, seeds 1000 + %d * rank ...), %d in flight per GPU" % (
                       world, mine_sat["regions"], a.sat_distinct, a.sat_distinct, a.streams),
                   "scaling": "weak", "value": sum_over_ranks(mine_sat["regions"] * a.reads) / dts, "unit": "reads/s",
                   "seconds_max_over_ranks": dts, "rank0": mine_sat, "min_busy_cus_over_ranks": busy_min}
        if rank == 0:
            line = {"metric": "reads/sec into POA (150bp, ~1.5k-node graph)", "value": n_total * a.steps / dt, "unit": "reads/s",
                    "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
                    "ms_per_step_with_ingest": 1e3 * (dt / a.steps + ingest_max),
                    "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                    "config": {"workload": "configs[2]: 100 seed genes x 1500 bp (seeds 100-199), 2000-10000 x 150bp reads each, %d alignments "
                                           "in one FASTA + one SAM, rambl.py options (%s)" % (n_total, RAMBL_OPTS),
                               "regions": len(rois), "regions_in_flight_per_gpu": streams, "level_execution": mode,
                               "parallelism": "regions sharded LPT over %d ranks, no exchange while they run, FASTA gather over %s" % (
                                   world, "RCCL" if backend == "nccl" else backend)},
                    "rank0": {"regions": len(mine), "alignments": n_in, "read_copies_after_ingest": n_graph,
                              "roofline": roofline(all_stats, a.steps, world), "breakdown_ms_per_step_summed_over_regions": breakdown(all_stats, a.steps)},
                    "contigs": fasta_out.count(">") if fasta_out else 0}
            line["roofline"] = line["rank0"].pop("roofline")
            if sat is not None:
                line["saturating_leg"] = sat
                line["scaling_model"] = scaling_model(world, work, slowest_alone, sat["min_busy_cus_over_ranks"], n_total)
                line["scaling_model"]["measured_makespan_s"] = dt / a.steps
                line["scaling_model"]["note"] = ("the configs[2] set holds 100 regions: sharded over N GPUs every region is in flight at once and the set "
                                                 "cannot finish before its slowest region has walked its ~1500 dependent levels -- strong scaling on this set is "
                                                 "capped by that latency, the saturating leg shows what the GPUs sustain")
            print(json.dumps(line), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return

    # ---- N = 1, configs[1]: one region
    d = tempfile.mkdtemp(prefix="scbench_")
    gene = synth.make_gene(21, glen=a.glen, n_strains=a.strains, n_reads=a.reads, name="gene21")
    fasta, sam = synth.write_dataset(os.path.join(d, "r0"), [gene])
    roi = "gene21:1-%d" % a.glen
    argv = stage5.straincall_argv(roi, fasta, sam)
    pa = cli.parse_cmd_line(argv)
    regions = cli.load_regions(pa)                       # host ingest (rows a1-a4), outside the headline's timed region
    n_graph = sum(sum(r.copies) for _, r in regions)
    params = capi.default_params(float(pa.error_rate), float(pa.tau), float(pa.diff_rate), want_timing=True)
    # a single region: one launch per level (its level kernels are kernels of their own: ~5 % fewer chain cycles than the same
    # code behind a call in the resident workgroup, and HIP events bracket every launch); resident workers are for many regions
    ctx = make_context(local, 1, False)
    single_mode = "one launch per level, straight from the region's host thread (a single region in flight)"

    def step(prepared, c=None):
        texts, stats = stage5.run_regions(c or ctx, prepared, 1, params)
        return "".join(texts), stats

    for _ in range(a.warmup):
        step([(pa, regions)])
    fence()
    t0 = time.time()
    all_stats, fasta_out = [], None
    for _ in range(a.steps):
        fasta_out, st = step([(pa, regions)])
        all_stats += st
    fence()
    dt = time.time() - t0
    # the same steps with rows a1-a4 inside the clock: open + index the SAM, scan windows, view / crop / thin / dedup
    fence()
    t0 = time.time()
    for _ in range(a.steps):
        pa_i = cli.parse_cmd_line(argv)
        with_ingest, _ = step([(pa_i, cli.load_regions(pa_i))])
    fence()
    dt_ingest = time.time() - t0
    ctx.close()
    rl = roofline(all_stats, a.steps, 1)

    line = {"metric": "reads/sec into POA (150bp, ~1.5k-node graph)", "value": a.reads * a.steps / dt, "unit": "reads/s",
            "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            "ms_per_step_with_ingest": 1e3 * dt_ingest / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1]: %d x 150bp reads vs one %dbp gene, %d strains, seed 21, rambl.py options (%s)" % (
                a.reads, a.glen, a.strains, RAMBL_OPTS),
                "regions_per_gpu": 1, "regions_in_flight_per_gpu": 1, "parallelism": "one region, one GPU (a single region does not shard: replicas only)",
                "level_execution": single_mode, "input_reads": a.reads, "read_copies_after_ingest": n_graph},
            "roofline": rl,
            "breakdown_ms_per_step": breakdown(all_stats, a.steps),
            "contigs": fasta_out.count(">") if fasta_out else 0,
            "with_ingest_matches": with_ingest == fasta_out}
    if not a.no_cpu:
        cb, cargs = cpu_baseline(d, fasta, sam, "gene21:%s" % a.sample_roi)
        ref_fa = cb.pop("fasta")
        # the same sample on the GPU (ingest inside the clock, as the CPU figure has it): the like-for-like ratio
        ctx2 = make_context(local, 1, False)
        pa2 = cli.parse_cmd_line(cargs)
        stage5.run_regions(ctx2, [(pa2, cli.load_regions(pa2))], 1)
        t1 = time.time()
        pa2 = cli.parse_cmd_line(cargs)
        got, _ = stage5.run_regions(ctx2, [(pa2, cli.load_regions(pa2))], 1)
        dt2 = time.time() - t1
        ctx2.close()
        cb["gpu_same_sample_reads_per_s"] = cb["sample_reads"] / dt2
        cb["gpu_same_sample_matches_cpu_fasta"] = ("".join(got) == ref_fa)
        cb["gpu_over_cpu_same_sample_one_core"] = cb["gpu_same_sample_reads_per_s"] / cb["value"]
        cb["gpu_over_cpu_same_sample_all_cores"] = cb["gpu_same_sample_reads_per_s"] / cb["all_cores"]["value"]
        cb["comparison"] = "the same sample on both sides, ingest included on both: gpu_same_sample_reads_per_s vs value (1 core) and vs all_cores.value"
        line["cpu_baseline"] = cb
    set_leg = None
    if not a.no_set:
        # the production shape (rambl.py stage 5 hands over one region per seed gene): the configs[2] set on this GPU with every
        # region in flight -- the N = 1 point of the sharded series that `--gpus N` runs; reported beside the headline, not as it
        ds = os.path.join(d, "set")
        os.makedirs(ds, exist_ok=True)
        rois, mine, prepared, ingest_s, n_in, n_graph_set, n_total = region_set(ds)
        streams = min(a.streams, len(mine))
        ctxs = make_context(local, streams, resident)
        stage5.run_regions(ctxs, prepared[:streams], streams)
        big = max(range(len(prepared)), key=lambda i: sum(sum(r.copies) for _, r in prepared[i][1]))
        t1 = time.time()
        stage5.run_regions(ctxs, [prepared[big]], 1)
        slowest_alone = time.time() - t1
        t1 = time.time()
        texts, st_set = stage5.run_regions(ctxs, prepared, streams)
        dts = time.time() - t1
        ctxs.close()
        set_leg = {"workload": "configs[2] set on one GPU: 100 regions, %d alignments" % n_total, "regions": len(rois),
                   "in_flight": streams, "value": n_total / dts, "unit": "reads/s", "seconds": dts,
                   "ingest_seconds": ingest_s, "value_with_ingest": n_total / (dts + ingest_s),
                   "read_copies_after_ingest": n_graph_set, "contigs": "".join(texts).count(">"),
                   "slowest_region_alone_s": slowest_alone, "work_level_kernel_seconds": sum(s["level_kernel_ticks"] for s in st_set) / 1e5 / 1e3,
                   "cu_busy_frac": sum(s["level_kernel_ticks"] for s in st_set) / 1e5 / 1e3 / dts / N_CU,
                   "avg_sampler_level_ms": sum(s["chain_wall_ticks"] for s in st_set) / 1e5 / max(sum(s["sampler_launches"] for s in st_set), 1)}
        line["regions_in_flight"] = set_leg
    if not a.no_saturation:
        # how many regions in flight fill the GPU, and what it then sustains: regions of the headline's shape (configs[1]
        # generator, seeds 1000 ...), slots x rounds of them through a context with that many slots
        dsat = os.path.join(d, "sat")
        os.makedirs(dsat, exist_ok=True)
        base = shape_regions(dsat, a.sat_distinct, 1000, a.reads, a.glen, a.strains)
        pts = [run_in_flight(local, base, int(x), a.sat_rounds, resident, a.reads) for x in a.sat_points.split(",") if int(x) >= 1]
        best = max(pts, key=lambda q: q["reads_per_s"])
        line["saturation"] = {"workload": "regions of the configs[1] shape (%d x 150bp reads, %d distinct data sets reused cyclically), in_flight x %d regions per point" % (
                                  a.reads, a.sat_distinct, a.sat_rounds),
                              "level_execution": mode, "points": pts, "plateau_reads_per_s": best["reads_per_s"],
                              "plateau_in_flight": best["in_flight"], "cu_busy_frac_at_plateau": best["cu_busy_frac"],
                              "definition": "cu_busy_frac = sum over regions of the time inside level kernels / (256 CUs x wall time): a level workgroup holds one CU"}
        if set_leg is not None:
            line["scaling_model"] = scaling_model(1, set_leg["work_level_kernel_seconds"], set_leg["slowest_region_alone_s"], best["busy_cus"], n_total)
            line["scaling_model"]["measured_makespan_s_one_gpu"] = set_leg["seconds"]
            line["scaling_model"]["workload"] = "the configs[2] set (what `--gpus N` shards): value = its alignments / predicted makespan"
    if not a.no_depth:
        line["hbm_bound_kernel"] = depth_scan_leg(local)
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
