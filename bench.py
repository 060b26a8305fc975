#!/usr/bin/env python3
"""Benchmark of the StrainCall hot path on MI355X.

A step = one pass of the path (graph build + level walk with its HIP kernels +
read_assign) over one region: BASELINE.json configs[1], 10 000 synthetic 150 bp
reads against one 1 500 bp gene (seed 21).  With N > 1 ranks every rank runs a
region of the same shape and seed (weak scaling), nothing is exchanged
while regions run, and the step ends with the RCCL gather of the FASTA bytes to
rank 0.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_READ = 320      # L + 16 + 4*ops + L for 150 bp, one-op reads
HBM_PEAK = 8.0e12             # MI355X_MICROARCH.md: HBM3E peak 8 TB/s


def cpu_baseline(data_dir, fasta, sam, roi):
    """Reference (oracle/_ref, kind "reference") or the C oracle (kind "port")
    on a bounded interior sample of the same data set, one core."""
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
    env["TMPDIR"] = data_dir
    args = ["-r", roi, "-q", "0", "-D", "800", "-I", "13", "-l", "70", "-t", "0.02", "-d", "0.02", "-w", "5000", fasta, sam]
    n_reads = len(subprocess.run([os.path.join(tools, "samtools"), "view", sam, "-q", "0", "-F", "1804", roi],
                                 stdout=subprocess.PIPE, env=env).stdout.splitlines())
    t0 = time.time()
    p = subprocess.run([exe] + args, cwd=data_dir, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    dt = time.time() - t0
    return dict(value=n_reads / dt, unit="reads/s", cores=1, kind=kind, seconds=dt, sample_reads=n_reads,
                sample="%s of the bench data set (reads cropped to the window), -O2 build, rambl.py options" % roi,
                fasta=p.stdout.decode()), args


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
    ap.add_argument("--regions", type=int, default=1, help="regions per GPU and step (1 = BASELINE configs[1]; >1 = configs[2]-style batch)")
    ap.add_argument("--streams", type=int, default=0, help="regions in flight per GPU (default: min(regions, 16))")
    a = ap.parse_args()

    streams = a.streams if a.streams > 0 else min(a.regions, 16)
    # one hardware queue per region in flight (the in-flight leg below uses 16); must precede HIP initialisation
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(min(max(streams, 16), 24)))
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("SC_BENCH_BACKEND", "nccl")      # "gloo" only to rehearse N > 1 on a one-GPU box
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local) if backend == "nccl" else torch.device("cpu")

    from rambl_amd import capi, cli, stage5, synth
    d = tempfile.mkdtemp(prefix="scbench_%d_" % rank)
    prepared = []
    for k in range(a.regions):
        seed = 21 + k                                    # the same regions on every rank: per-GPU work is fixed (weak scaling)
        gene = synth.make_gene(seed, glen=a.glen, n_strains=a.strains, n_reads=a.reads, name="gene%d" % seed)
        fasta, sam = synth.write_dataset(os.path.join(d, "r%d" % k), [gene])
        roi = "%s:1-%d" % (gene["name"], a.glen)
        pa = cli.parse_cmd_line(stage5.straincall_argv(roi, fasta, sam))
        prepared.append((pa, cli.load_regions(pa)))      # host ingest, outside the timed region
    pa, regions = prepared[0]
    fasta, sam = pa.gene_file, pa.mapping_file
    gene = {"name": "gene21"}
    params = capi.default_params(float(pa.error_rate), float(pa.tau), float(pa.diff_rate), want_timing=True)
    ctx = capi.Context(local, streams)

    def step():
        texts, stats = stage5.run_regions(ctx, prepared, streams, params)
        full = stage5.gather_fasta(["".join(texts)], [rank], world, dist if world > 1 else None, dev)
        return full, stats

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.time()
    all_stats = []
    fasta_out = None
    for _ in range(a.steps):
        fasta_out, st = step()
        all_stats += st
    fence()
    dt = time.time() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)   # MAX over ranks
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ctx.close()

    if rank == 0:
        total_reads = a.reads * a.regions * a.steps * world
        k_ms = sum(s["sampler_kernel_ms"] for s in all_stats)
        k_n = sum(s["sampler_launches"] for s in all_stats)
        k_copies = sum(s["sampler_read_copies"] for s in all_stats)
        draws = sum(s["draws"] for s in all_stats)
        avg_ms = k_ms / max(k_n, 1)
        achieved = (ALG_BYTES_PER_READ * k_copies / max(k_n, 1)) / (avg_ms * 1e-3) if k_n else 0.0
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "r01", "pmc_summary.json")
        if os.path.exists(pmc) and world == 1:
            # HBM bytes per SAMPLE launch from the committed rocprofv3 --pmc passes of this command
            traffic = json.load(open(pmc)).get("sample_traffic_bytes_per_launch")
            traffic_src = "profiles/r01/pmc_summary.json (separate FETCH_SIZE / WRITE_SIZE passes, FETCH doubled)"
        line = {
            "metric": "reads/sec into POA (150bp, ~1.5k-node graph)", "value": total_reads / dt, "unit": "reads/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1]: %d x 150bp reads vs one %dbp gene, %d strains, seed 21, "
                                   "rambl.py options (-q 0 -D 800 -I 13 -l 70 -t 0.02 -d 0.02 -w 5000)" % (a.reads, a.glen, a.strains),
                       "regions_per_gpu": a.regions, "regions_in_flight_per_gpu": streams,
                       "parallelism": "region-sharded x%d, FASTA gather over RCCL" % world},
            "roofline": {"bound": "hbm", "kernel": "sc::k_chain_w<NB,L> (urn sampler of one level)", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": ALG_BYTES_PER_READ * k_copies / max(k_n, 1),
                         "avg_launch_ms": avg_ms, "launches_per_step": k_n / max(a.steps, 1),
                         "draws_per_s_per_region": draws / (k_ms * 1e-3) if k_ms else 0.0},
            "breakdown_ms_per_step": {"graph_host": sum(s["graph_ms"] for s in all_stats) / a.steps,
                                      "level_walk": sum(s["cluster_ms"] for s in all_stats) / a.steps,
                                      "sampler_kernels": k_ms / a.steps,
                                      "draws": draws / a.steps,
                                      "slow_tier_draws": sum(s["slow_draws"] for s in all_stats) / a.steps,
                                      "exact_draws": sum(s["exact_draws"] for s in all_stats) / a.steps,
                                      "chain_passes": sum(s["chain_passes"] for s in all_stats) / a.steps,
                                      "chain_cycles_per_draw": sum(s["chain_cycles"] for s in all_stats) / max(draws, 1),
                                      "chain_ns_per_draw": 10.0 * sum(s["chain_wall_ticks"] for s in all_stats) / max(draws, 1),
                                      "avg_candidates_per_sampler_launch": sum(s["sampler_strains"] for s in all_stats) / max(k_n, 1)},
            "contigs": fasta_out.count(">") if fasta_out else 0,
        }
        if not a.no_cpu and a.regions == 1:
            cb, cargs = cpu_baseline(d, fasta, sam, "%s:%s" % (gene["name"], a.sample_roi))
            ref_fa = cb.pop("fasta")
            # the same sample on the GPU, for a like-for-like ratio and a parity check of the sample
            pa2 = cli.parse_cmd_line(cargs)
            regs2 = cli.load_regions(pa2)
            ctx2 = capi.Context(local, 1)
            t1 = time.time()
            got = "".join(cli.format_fasta(w, ctx2.run(r, params), pa2.tau) for w, r in regs2)
            dt2 = time.time() - t1
            ctx2.close()
            cb["gpu_same_sample_reads_per_s"] = cb["sample_reads"] / dt2
            cb["gpu_same_sample_matches_cpu_fasta"] = (got == ref_fa)
            line["cpu_baseline"] = cb
            if world == 1:
                # the production shape (rambl.py stage 5 hands over one region per seed gene): 16 regions of the
                # same size in flight on this GPU, separate streams; reported beside the headline, not as it
                prep16 = []
                for k in range(16):
                    g16 = synth.make_gene(21 + k, glen=a.glen, n_strains=a.strains, n_reads=a.reads, name="gene%d" % (21 + k))
                    fa16, sam16 = synth.write_dataset(os.path.join(d, "f%d" % k), [g16])
                    pa16 = cli.parse_cmd_line(stage5.straincall_argv("%s:1-%d" % (g16["name"], a.glen), fa16, sam16))
                    prep16.append((pa16, cli.load_regions(pa16)))
                ctx16 = capi.Context(local, 16)
                params16 = capi.default_params(float(pa.error_rate), float(pa.tau), float(pa.diff_rate))   # no per-launch events
                stage5.run_regions(ctx16, prep16, 16, params16)
                t1 = time.time()
                stage5.run_regions(ctx16, prep16, 16, params16)
                dt16 = time.time() - t1
                ctx16.close()
                line["regions_in_flight"] = {"regions": 16, "reads": 16 * a.reads, "value": 16 * a.reads / dt16, "unit": "reads/s",
                                             "seconds": dt16, "note": "16 regions of the configs[1] shape (seeds 21-36) on 16 streams of one GPU"}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
