"""Stage 5 of rambl.py (strain-level assembly) for one node of MI355X GPUs.

Mirror of /root/reference/scripts/rambl.py:169-201 (`strain_call`) and :236-238
(the `seqtk seq -L 400` length filter): the region list comes from
`seed_otus.fasta.fai` (`name:1-len` per record), every region is one independent
StrainCall run with rambl.py's option defaults (:260-271), the per-region FASTA
files are concatenated in .fai order.

The reference fans regions out over a process pool; here one process drives one
GPU with many regions in flight (the library's level server batches their
kernels), regions are sharded over ranks longest-processing-time-first by the
reads that reach the graph, nothing is exchanged while they run,
and the only collective is the final gather of FASTA bytes to rank 0 in .fai
order (torch.distributed: RCCL over xGMI on GPUs, gloo in CPU tests).
"""
import os

import numpy as np

from . import cli, ingest, samio

RAMBL_DEFAULTS = dict(map_qual=0, max_depth=800, max_ins=13, read_len=70, tau=0.02, diff_rate=0.02)


def roi_list(fai_path):
    """rambl.py:172-175."""
    rois = []
    with open(fai_path) as f:
        for line in f:
            field = line.split()
            if field:
                rois.append("%s:1-%s" % (field[0], field[1]))
    return rois


def straincall_argv(roi, fasta, bam, opts=None):
    """rambl.py:181-187."""
    o = dict(RAMBL_DEFAULTS)
    o.update(opts or {})
    return ["-r", str(roi), "-q", str(o["map_qual"]), "-D", str(o["max_depth"]), "-I", str(o["max_ins"]),
            "-l", str(o["read_len"]), "-t", str(o["tau"]), "-d", str(o["diff_rate"]), "-w", str(5000), fasta, bam]


def lpt_shards(costs, n):
    """Longest-processing-time-first partition of unit indices over n ranks."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * n
    shards = [[] for _ in range(n)]
    for i in order:
        r = min(range(n), key=lambda k: (load[k], k))
        shards[r].append(i)
        load[r] += costs[i]
    return shards


def seqtk_L(fasta_text, min_len=400):
    """`seqtk seq -L 400` on the concatenated result (rambl.py:236-238)."""
    out = []
    name = None
    seq = []
    for line in fasta_text.splitlines():
        if line.startswith(">"):
            if name is not None and len("".join(seq)) >= min_len:
                out.append(name + "\n" + "".join(seq) + "\n")
            name, seq = line, []
        elif name is not None:
            seq.append(line.strip())
    if name is not None and len("".join(seq)) >= min_len:
        out.append(name + "\n" + "".join(seq) + "\n")
    return "".join(out)


def prepare_region(roi, fasta, bam, opts=None, shared=None):
    """argv -> [(window, reads)] (host ingest, rows a1-a4)."""
    pa = cli.parse_cmd_line(straincall_argv(roi, fasta, bam, opts))
    if shared is None:
        return pa, cli.load_regions(pa)
    fa, fai, aln = shared
    windows = ingest.make_scan_window(pa, fai, aln)
    out = []
    for gn, p0, p1 in windows:
        r = "%s:%d-%d" % (gn, p0, p1)
        out.append(((gn, p0, p1), ingest.load_mapping_reads(fa.fetch(r), aln, pa.mapping_qual, pa.read_len, pa.max_ins,
                                                            pa.max_depth, r)))
    return pa, out


def _prepare_guarded(roi, fasta, bam, opts, shared):
    """A region that cannot be ingested (no read covers the ROI: the reference dereferences an empty map there and
    dies, leaving an empty <roi>.fa behind, rambl.py:159-166) must not take the other regions with it."""
    try:
        return prepare_region(roi, fasta, bam, opts, shared)
    except Exception as e:                     # noqa: BLE001 - the message goes to stderr, the region yields no contig
        return RegionFailure(roi, "%s: %s" % (type(e).__name__, e))


class RegionFailure:
    def __init__(self, roi, message):
        self.roi = roi
        self.message = message


def prepared_stream(rois, fasta, bam, opts=None, workers=4, shared=None):
    """Host ingest of the regions, in order, on `workers` threads: the library reads the alignment file once
    (capi.NativeAln) and a window's ingest is one GIL-free call, so threads scale.  Yields (pa, [(window, reads)])
    or a RegionFailure per region."""
    import concurrent.futures
    if shared is None:
        shared = (samio.Fasta(fasta), samio.read_fai(fasta + ".fai"), samio.Alignments(bam))
    if workers <= 1:
        for roi in rois:
            yield _prepare_guarded(roi, fasta, bam, opts, shared)
        return
    with concurrent.futures.ThreadPoolExecutor(workers) as ex:
        window = max(2 * workers, 4)
        futs = []
        it = iter(rois)
        for roi in it:
            futs.append(ex.submit(_prepare_guarded, roi, fasta, bam, opts, shared))
            if len(futs) >= window:
                yield futs.pop(0).result()
        for f in futs:
            yield f.result()


def run_regions(ctx, prepared, streams=1, params=None, errors=None):
    """prepared: iterable of (pa, [(window, reads)]) or RegionFailure -- a generator is consumed lazily, so the host
    ingest of the next region overlaps the regions in flight.  Submits every window, `streams` in flight, returns the
    FASTA text of each entry of `prepared` (in order) and the per-window stats.  A region that fails (ingest, or a
    graph shape outside the device model) yields "" -- the empty <roi>.fa the reference pipeline would be left with --
    and is reported in `errors` (list of (index, message)) and on stderr."""
    import sys
    from . import capi
    texts = []
    pending = []
    stats = []

    def fail(idx, what, msg):
        sys.stderr.write("StrainCall: region %s yields no contig: %s\n" % (what, msg))
        if errors is not None:
            errors.append((idx, msg))

    def drain(k):
        while len(pending) > k:
            idx, wi, window, pa, h = pending.pop(0)
            try:
                res = ctx.wait(h)
            except capi.StrainCallError as e:
                fail(idx, "%s:%d-%d" % window, str(e))
                continue
            texts[idx].append((wi, cli.format_fasta(window, res, pa.tau)))
            stats.append(res.stats)

    for idx, item in enumerate(prepared):
        texts.append([])
        if isinstance(item, RegionFailure):
            fail(idx, item.roi, item.message)
            continue
        pa, regs = item
        p_ = params or capi.default_params(float(pa.error_rate), float(pa.tau), float(pa.diff_rate))
        for wi, (window, reads) in enumerate(regs):
            if len(reads) == 0:
                continue
            try:
                pending.append((idx, wi, window, pa, ctx.submit(reads, p_)))
            except capi.StrainCallError as e:
                fail(idx, "%s:%d-%d" % window, str(e))
                continue
            # a few more than the context has slots: the library queues them, so a slot that falls free finds its next
            # region at once even when the oldest region in flight (the one waited for here) is a slow one
            drain(max(streams, 1) + max(2, streams // 4))
    drain(0)
    return ["".join(t for _, t in sorted(x)) for x in texts], stats


def region_costs(fai, aln, opts=None):
    """Cost of a region for the longest-processing-time-first partition: the alignments that
    reach the graph after thinning to max_depth, plus the gene length (every level costs a launch)."""
    o = dict(RAMBL_DEFAULTS)
    o.update(opts or {})
    costs = []
    for name, ln in fai:
        glen = max(float(ingest.stoi(ln)), 1.0)
        native = getattr(aln, "native", None)
        if native is None:
            costs.append(glen)
            continue
        n, bases = native.ref_stats(name)
        depth = bases / glen
        rho = min(1.0, o["max_depth"] / depth) if depth > 0 else 1.0
        costs.append(n * rho + glen)
    return costs


def shard_alignments(fai, bam, world, rank, opts=None):
    """-> (this rank's region indices, the alignment file holding their records).  One rank: the file is read once.
    Several ranks on one host: every rank prices all regions from a records-free pass over the file (the per-reference
    statistics), takes its longest-processing-time-first shard, and keeps only the records of its own references --
    an eighth of the memory and of the record work per rank at eight ranks (rambl.py:190-194 hands every region's
    process the whole BAM through samtools)."""
    if world <= 1:
        aln = samio.Alignments(bam)
        return lpt_shards(region_costs(fai, aln, opts), 1)[0], aln
    probe = samio.Alignments(bam, only=[])
    costs = region_costs(fai, probe, opts)
    if probe.native is not None:
        probe.native.close()
    mine = lpt_shards(costs, world)[rank]
    return mine, samio.Alignments(bam, only=[fai[i][0] for i in mine])


def gather_fasta(local_texts, local_ids, n_units, dist=None, device=None):
    """Variable-length gather of per-region FASTA bytes to rank 0, returned in
    unit order (replaces `cat` in rambl.py:197-201).  dist=None: single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        out = [""] * n_units
        for i, t in zip(local_ids, local_texts):
            out[i] = t
        return "".join(out)
    import torch
    world = dist.get_world_size()
    dev = device if device is not None else torch.device("cpu")
    blob = bytearray()
    index = []
    for i, t in zip(local_ids, local_texts):
        b = t.encode("ascii")
        index += [i, len(b)]
        blob += b
    meta = torch.tensor([len(index) // 2, len(blob)], dtype=torch.int64, device=dev)
    metas = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(metas, meta)
    max_units = int(max(int(m[0]) for m in metas))
    max_bytes = int(max(int(m[1]) for m in metas))
    idx_t = torch.full((2 * max(max_units, 1),), -1, dtype=torch.int64, device=dev)
    if index:
        idx_t[:len(index)] = torch.tensor(index, dtype=torch.int64, device=dev)
    pay = torch.zeros(max(max_bytes, 1), dtype=torch.uint8, device=dev)
    if blob:
        pay[:len(blob)] = torch.tensor(np.frombuffer(bytes(blob), dtype=np.uint8).copy(), dtype=torch.uint8, device=dev)
    idxs = [torch.empty_like(idx_t) for _ in range(world)]
    pays = [torch.empty_like(pay) for _ in range(world)]
    dist.all_gather(idxs, idx_t)
    dist.all_gather(pays, pay)
    if dist.get_rank() != 0:
        return None
    out = [""] * n_units
    for r in range(world):
        ix = idxs[r].cpu().tolist()
        data = pays[r].cpu().numpy().tobytes()
        off = 0
        for k in range(int(metas[r][0])):
            u, ln = ix[2 * k], ix[2 * k + 1]
            out[u] = data[off:off + ln].decode("ascii")
            off += ln
    return "".join(out)


def strain_call(fasta, bam, out_dir=None, prefix="rambl", opts=None, device=0, streams=4, dist=None, torch_device=None,
                ingest_workers=4, errors=None, ctx=None):
    """rambl.py `strain_call` for this rank's shard; rank 0 returns the concatenated
    FASTA (and writes <out_dir>/3_straincall_results/<roi>.fa + <prefix>.fa)."""
    from . import capi
    rois = roi_list(fasta + ".fai")
    fai = samio.read_fai(fasta + ".fai")
    world = dist.get_world_size() if (dist is not None and dist.is_initialized()) else 1
    rank = dist.get_rank() if world > 1 else 0
    mine, aln = shard_alignments(fai, bam, world, rank, opts)
    shared = (samio.Fasta(fasta), fai, aln)
    prepared = prepared_stream([rois[i] for i in mine], fasta, bam, opts, ingest_workers, shared)   # ingest overlaps the regions in flight
    errs = []
    if ctx is not None:
        texts, _ = run_regions(ctx, prepared, streams, errors=errs)
    else:
        with capi.Context(device, streams) as c:
            texts, _ = run_regions(c, prepared, streams, errors=errs)
    if errors is not None:
        errors.extend((rois[mine[i]], m) for i, m in errs)
    if out_dir is not None:
        sc_dir = os.path.join(out_dir, "3_straincall_results")
        os.makedirs(sc_dir, exist_ok=True)
        for i, t in zip(mine, texts):
            with open(os.path.join(sc_dir, "%s.fa" % rois[i]), "w") as f:
                f.write(t)
    full = gather_fasta(texts, mine, len(rois), dist, torch_device)
    if rank == 0 and out_dir is not None:
        with open(os.path.join(out_dir, "%s.fa" % prefix), "w") as f:
            f.write(full)
    return full


def main(argv=None):
    """`python -m rambl_amd.stage5 seed_otus.fasta to_seed_otus.all.bam -o WORKDIR -p PREFIX`
    (under `torchrun --nproc-per-node N` for N GPUs): stage 5 of rambl.py + the length filter.
    Exit status 1 when a region failed (its <roi>.fa is empty, as under the reference pipeline)."""
    import argparse
    ap = argparse.ArgumentParser(description="rambl.py stage 5 (strain-level assembly) on MI355X")
    ap.add_argument("fasta")
    ap.add_argument("alignments", help="BAM or SAM text of the reads aligned to the seed genes")
    ap.add_argument("-o", "--out-dir", default=".")
    ap.add_argument("-p", "--prefix", default="rambl")
    ap.add_argument("-s", "--streams", type=int, default=224, help="regions in flight per GPU")
    ap.add_argument("-j", "--ingest-workers", type=int, default=4,
                    help="host threads preparing the next regions (at most the rank's share of the host's CPUs: sc_host_plan)")
    for k, v in RAMBL_DEFAULTS.items():
        ap.add_argument("--" + k.replace("_", "-"), default=v, type=type(v))
    a = ap.parse_args(argv)
    opts = {k: getattr(a, k) for k in RAMBL_DEFAULTS}
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = dev = None
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=int(os.environ["RANK"]), world_size=world)
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    errors = []
    from . import capi
    capi.host_bind(local)          # this rank's threads next to its GPU (a two-socket host; SC_NUMA_BIND=0: leave them)
    a.ingest_workers = max(1, min(a.ingest_workers, capi.host_plan(max(a.streams, 1))[2]))      # 8 ranks on 16 CPUs: one each
    full = strain_call(a.fasta, a.alignments, out_dir=a.out_dir, prefix=a.prefix, opts=opts, device=local,
                       streams=a.streams, dist=dist, torch_device=dev, ingest_workers=a.ingest_workers, errors=errors)
    if full is not None:
        with open(os.path.join(a.out_dir, "%s.filtered.fa" % a.prefix), "w") as f:
            f.write(seqtk_L(full, 400))            # rambl.py:236-238
    failed = len(errors)
    if world > 1:
        import torch
        t = torch.tensor([failed], dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        failed = int(t.item())
        dist.barrier()
        dist.destroy_process_group()
    return 1 if failed else 0


if __name__ == "__main__":
    import sys
    sys.exit(main())
