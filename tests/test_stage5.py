"""CPU: rambl.py stage-5 mirror -- region list, argv, LPT sharding, length filter,
and the N>1 gather of FASTA bytes with world_size 2 over gloo."""
import os
import subprocess
import sys
import textwrap

from rambl_amd import stage5

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_roi_list_and_argv(tmp_path):
    fai = tmp_path / "seed_otus.fasta.fai"
    fai.write_text("otuA\t1502\t6\t60\t61\notuB\t1430\t1540\t60\t61\n")
    assert stage5.roi_list(str(fai)) == ["otuA:1-1502", "otuB:1-1430"]
    argv = stage5.straincall_argv("otuA:1-1502", "f.fa", "r.bam")
    assert argv == ["-r", "otuA:1-1502", "-q", "0", "-D", "800", "-I", "13", "-l", "70", "-t", "0.02", "-d", "0.02",
                    "-w", "5000", "f.fa", "r.bam"]


def test_lpt_and_length_filter():
    shards = stage5.lpt_shards([5, 9, 1, 7, 3, 3], 2)
    assert sorted(sum(shards, [])) == list(range(6))
    loads = [sum([5, 9, 1, 7, 3, 3][i] for i in s) for s in shards]
    assert abs(loads[0] - loads[1]) <= 2
    fa = ">a\n" + "A" * 399 + "\n>b\n" + "C" * 400 + "\n"
    assert stage5.seqtk_L(fa) == ">b\n" + "C" * 400 + "\n"


def test_gather_single_process():
    assert stage5.gather_fasta([">x\nAC\n", ">y\nGT\n"], [1, 0], 2) == ">y\nGT\n>x\nAC\n"


def test_gather_world2_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent('''
        import os, sys
        sys.path.insert(0, %r)
        import torch.distributed as dist
        from rambl_amd import stage5
        dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
        r = dist.get_rank()
        units = [">u%%d\\n%%s\\n" %% (i, "ACGT" * (i + 1)) for i in range(5)]
        mine = stage5.lpt_shards([5.0, 4.0, 3.0, 2.0, 1.0], 2)[r]
        full = stage5.gather_fasta([units[i] for i in mine], mine, 5, dist)
        if r == 0:
            assert full == "".join(units), full
            open(%r, "w").write("ok")
        dist.barrier()
        dist.destroy_process_group()
    ''' % (ROOT, str(tmp_path / "ok"))))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)], env=env,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    assert (tmp_path / "ok").read_text() == "ok"


def test_eight_ranks_on_one_host_gloo(tmp_path):
    """Eight ranks under torch.distributed.run on THIS host (gloo, no GPU; LOCAL_WORLD_SIZE = 8 comes from the launcher): every
    rank keeps the records of its own shard only, plans its host threads from an eighth of the CPUs, runs stage 5 of its
    shard through a stand-in context and the gather; rank 0 gets every region in .fai order.  The thread plans of the eight
    ranks together stay within the host (sixteen planned threads on a 16-CPU share: sc_host_plan with the launcher's
    LOCAL_WORLD_SIZE) and no rank's ingest spawns more OS threads than its plan."""
    from rambl_amd import synth
    genes = [synth.make_gene(800 + k, glen=300, n_strains=2, n_reads=120 + 40 * k, rlen=110, name="g%02d" % k) for k in range(12)]
    fa, sam = synth.write_dataset(str(tmp_path / "data"), genes)
    script = tmp_path / "w8.py"
    script.write_text(textwrap.dedent('''
        import json, os, sys
        sys.path.insert(0, %r)
        import torch.distributed as dist
        from rambl_amd import capi, stage5
        dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
        r = dist.get_rank()
        assert os.environ["LOCAL_WORLD_SIZE"] == "8"
        threads_before = len(os.listdir("/proc/self/task"))
        plan16 = capi.host_plan(224, 0, 16.0)             # what this rank starts on a 16-CPU share of the host
        plan_here = capi.host_plan(224)

        class Res:
            def __init__(self, n):
                self.seqs, self.abundance, self.stats = ["ACGT" * n], [1.0], {}

        class Ctx:                                        # stand-in for the device: a region's "strain" says how many reads it had
            def __init__(self):
                self.jobs = {}
            def submit(self, reads, params):
                self.jobs[len(self.jobs) + 1] = len(reads)
                return len(self.jobs)
            def wait(self, h):
                return Res(self.jobs[h])

        peak = [threads_before]
        full = stage5.strain_call(%r, %r, out_dir=%r, prefix="w8", dist=dist, ctx=Ctx(), streams=4,
                                  ingest_workers=max(1, min(4, plan_here[2])))
        peak.append(len(os.listdir("/proc/self/task")))
        open(os.path.join(%r, "rank%%d.json" %% r), "w").write(json.dumps({"plan16": plan16, "plan_here": plan_here,
                                                                             "threads": peak, "full": full if r == 0 else None}))
        dist.barrier()
        dist.destroy_process_group()
    ''' % (ROOT, fa, sam, str(tmp_path / "out"), str(tmp_path))))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                           "--master-addr", "127.0.0.1", "--master-port", "29547", str(script)], env=env,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    import json
    ranks = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(8)]
    assert all(tuple(x["plan16"]) == (1, 1, 2) for x in ranks)                    # 8 x (1 executor + 1 level server) = 16 CPUs
    assert sum(x["plan16"][0] + x["plan16"][1] for x in ranks) <= 16
    assert all(x["plan_here"][0] + x["plan_here"][1] <= max(2, (os.cpu_count() or 8) // 8 + 1) for x in ranks)
    # the native reader's threads end with sc_aln_open; what is left afterwards are the rank's own (torch, gloo, the pool)
    assert all(x["threads"][1] - x["threads"][0] <= 4 + 2 for x in ranks), [x["threads"] for x in ranks]
    full = ranks[0]["full"]
    names = [l[1:] for l in full.splitlines() if l.startswith(">")]
    assert len(names) == 12 and names == sorted(names)                            # every region once, in .fai order
    assert (tmp_path / "out" / "w8.fa").read_text() == full


def _two_gene_dataset(d):
    from rambl_amd import synth
    genes = [synth.make_gene(700, glen=300, n_strains=2, n_reads=150, rlen=110, name="full"),
             synth.make_gene(701, glen=300, n_strains=1, n_reads=0, rlen=110, name="bare"),
             synth.make_gene(702, glen=300, n_strains=1, n_reads=1200, rlen=110, name="deep")]
    return synth.write_dataset(d, genes)


def test_region_costs_follow_reads_after_thinning(tmp_path):
    """LPT cost = alignments that reach the graph after thinning to -D, plus the gene length."""
    from rambl_amd import samio
    fa, sam = _two_gene_dataset(str(tmp_path))
    fai = samio.read_fai(fa + ".fai")
    aln = samio.Alignments(sam)
    full, bare, deep = stage5.region_costs(fai, aln, {"max_depth": 100})
    assert bare == 300.0                                   # no alignment: only the levels
    assert abs(full - (150 + 300)) < 1                     # depth 55 < 100: every read counts
    assert 300 + 250 < deep < 300 + 330                    # depth 440 thinned to 100: ~1200 * 100 / 440 reads
    assert stage5.lpt_shards([full, bare, deep], 2) == [[2], [0, 1]] or stage5.lpt_shards([full, bare, deep], 2) == [[0, 1], [2]][::-1]


def test_failing_region_does_not_take_the_others_down(tmp_path, capsys):
    """rambl.py ignores StrainCall's exit status (rambl.py:159-166): a region that cannot be assembled leaves an empty
    <roi>.fa and the others are still concatenated.  Here: an ROI no read covers (ingest raises), and a region the
    device refuses -- with a stand-in context, this test runs without a GPU."""
    from rambl_amd import capi
    fa, sam = _two_gene_dataset(str(tmp_path))
    rois = stage5.roi_list(fa + ".fai")
    prepared = list(stage5.prepared_stream(rois, fa, sam, None, workers=2))
    assert isinstance(prepared[1], stage5.RegionFailure) and prepared[1].roi == "bare:1-300"
    assert not isinstance(prepared[0], stage5.RegionFailure) and len(prepared[0][1][0][1]) > 0

    class Res:
        seqs, abundance, stats = ["ACGT"], [1.0], {}

    class FakeCtx:
        def __init__(self):
            self.n = 0

        def submit(self, reads, params):
            self.n += 1
            return self.n

        def wait(self, h):
            if h == 2:
                raise capi.StrainCallError(-4, "more than 16 distinct symbols")
            return Res()

    errors = []
    texts, stats = stage5.run_regions(FakeCtx(), prepared, streams=2, params=object(), errors=errors)
    assert texts[0].startswith(">contigfull") and texts[1] == "" and texts[2] == ""
    assert [i for i, _ in errors] == [1, 2]
    err = capsys.readouterr().err
    assert "bare:1-300" in err and "deep" in err and "SC_ERR_UNSUPPORTED" in err


def test_fiber_pool_runs_regions_on_a_few_threads(tmp_path):
    """rambl_amd/csrc/sc_fiber.hpp on the CPU: 512 regions-as-fibers on 4 executor threads walk 300 levels each (park
    per level, a server thread makes them ready); never more than 4 run at once and the process never has more than
    4 + 2 threads -- the structure that replaces the thread per region in flight."""
    import json
    exe = str(tmp_path / "fiber_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", exe, os.path.join(ROOT, "tests", "native", "fiber_check.cpp")])
    for r, t, lv in ((512, 4, 300), (48, 8, 1500), (3, 1, 200)):
        rec = json.loads(subprocess.run([exe, str(r), str(t), str(lv)], stdout=subprocess.PIPE, check=True, timeout=300).stdout)
        assert rec["ok"] and rec["finished"] == r and rec["max_running"] <= t and rec["max_os_threads"] <= t + 2, rec
    # the way resident contexts run: nobody makes the fibers ready -- the "device" only stamps the levels, the executors find the
    # stamps between two fibers and while they spin (set_poll), somebody keeps watching while anything is waited for; with and
    # without separate set-up threads, few and many regions (several times each: what went wrong here went wrong rarely)
    for rep in range(3):
        for r, t, lv, tl in ((224, 8, 600, 3), (300, 6, 400, 0), (3, 8, 1000, 4), (512, 4, 200, 0)):
            rec = json.loads(subprocess.run([exe, str(r), str(t), str(lv), str(tl), "poll"], stdout=subprocess.PIPE, check=True, timeout=120).stdout)
            assert rec["ok"] and rec["finished"] == r and rec["max_running"] <= t and rec["max_os_threads"] <= t + 2, rec


def test_eight_ranks_fit_sixteen_cpus():
    """Eight ranks on one host (one per GPU, LOCAL_WORLD_SIZE = 8) with 128-512 regions in flight each: the host
    threads their contexts start (executors + level server) stay within the host's CPUs, whatever the number of
    regions in flight; rambl.py's Pool(cores) sized itself to the cores the same way (scripts/rambl.py:190-194)."""
    from rambl_amd import capi
    for streams in (1, 16, 128, 512):
        for cpus, world in ((16, 8), (16, 1), (256, 8), (2, 8)):
            ex, srv, ing = capi.host_plan(streams, world, float(cpus))
            share = max(cpus / world, 1)
            assert 1 <= ex <= min(streams, 32) and srv == (1 if streams > 1 else 0) and 1 <= ing <= 32
            assert ex + srv <= max(share, 2) and ing <= max(share, 1), (streams, cpus, world, ex, srv, ing)
            if world == 8:
                assert 8 * (ex + srv) <= max(cpus, 16)
    # the environment's LOCAL_WORLD_SIZE is what a rank under torch.distributed.run sees
    code = ("import os, sys; sys.path.insert(0, %r); from rambl_amd import capi; "
            "print(capi.host_plan(256, 0, 16.0))" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LOCAL_WORLD_SIZE="8"), stdout=subprocess.PIPE, check=True)
    assert out.stdout.decode().strip() == "(1, 1, 2)"
    out = subprocess.run([sys.executable, "-c", code], env={k: v for k, v in os.environ.items() if k != "LOCAL_WORLD_SIZE"},
                         stdout=subprocess.PIPE, check=True)
    assert out.stdout.decode().strip() == "(15, 1, 16)"


def test_a_rank_keeps_the_records_of_its_own_shard(tmp_path):
    """sc_aln_open_filtered: the per-reference statistics (what LPT prices regions with) cover every reference of the file
    whatever is kept; the windows of a kept reference are ingested exactly as from the unfiltered file; a reference that
    was not kept has no reads.  SAM text (parsed in stretches on several threads) and BAM."""
    import sys as _sys
    _sys.path.insert(0, os.path.join(ROOT, "tests"))
    import sc_testlib as T
    from rambl_amd import capi, samio, synth
    genes = [synth.make_gene(900 + k, glen=400, n_strains=2, n_reads=300 + 200 * k, rlen=110, name="g%d" % k, paired=(k == 1)) for k in range(4)]
    fa, sam = synth.write_dataset(str(tmp_path), genes)
    bam = str(tmp_path / "reads.bam")
    T.write_bam(sam, bam)
    fai = samio.read_fai(fa + ".fai")
    os.environ["SC_INGEST_THREADS"] = "3"
    os.environ["SC_INGEST_MIN_CHUNK"] = "4096"            # the small text is still cut into three stretches
    try:
        for path in (sam, bam):
            full = capi.NativeAln(path)
            none = capi.NativeAln(path, only=[])
            some = capi.NativeAln(path, only=["g1", "g3"])
            for name, _ in fai:
                assert none.ref_stats(name) == full.ref_stats(name) == some.ref_stats(name) and full.ref_stats(name)[0] > 0
            assert none.records() == full.records() == some.records()
            for name, ln in fai:
                ln = int(ln)
                want = full.load_reads("", name, 1, ln, 0, 70, 13, 800)
                got = some.load_reads("", name, 1, ln, 0, 70, 13, 800)
                if name in ("g1", "g3"):
                    assert (got.pos, got.cigar, got.seq, got.copies, got.mates, got.n_input) == (want.pos, want.cigar, want.seq, want.copies, want.mates, want.n_input)
                    assert some.pileup_flags(0, name, 1, ln) == full.pileup_flags(0, name, 1, ln)
                else:
                    assert len(got) == 0 and got.n_input == 0 and len(want) > 0
                assert len(none.load_reads("", name, 1, ln, 0, 70, 13, 800)) == 0
        # the shards of two ranks: every region once, each rank's file holds its own references
        for rank in (0, 1):
            mine, aln = stage5.shard_alignments(fai, sam, 2, rank)
            assert sorted(mine + stage5.shard_alignments(fai, sam, 2, 1 - rank)[0]) == [0, 1, 2, 3]
            for i, (name, ln) in enumerate(fai):
                n = len(aln.native.load_reads("", name, 1, int(ln), 0, 70, 13, 800))
                assert (n > 0) == (i in mine)
    finally:
        os.environ.pop("SC_INGEST_THREADS", None)
        os.environ.pop("SC_INGEST_MIN_CHUNK", None)
