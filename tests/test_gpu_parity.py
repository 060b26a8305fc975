"""-m gpu: the HIP path (through the C-ABI) against the oracle on seeded inputs."""
import os
import random

import pytest

import sc_testlib as T

pytestmark = pytest.mark.gpu


# 312: log-likelihoods below -745 for every candidate at a soft-update level (fp64 exp underflow); 465: NaN
# abundances in the reference itself; 744: every candidate pruned before the end of the gene (the reference
# prints nothing and exits 0); 7705: a single-character strain label against a two-character read label (the
# reference's never-set count -> log 0)
@pytest.mark.parametrize("seed", list(range(0, 16)) + [312, 465, 744, 7705])
def test_region_parity(seed, tmp_path, oracle_bin):
    d = str(tmp_path)
    args = T.make_case(seed, d)
    exp_fa, exp_tr = T.run_oracle(args, d, trace=True)
    exp_g, _ = T.run_oracle(args, d, graph=True)
    tf = os.path.join(d, "product.trace")
    got_fa = T.run_product(args, trace_file=tf)
    got_g = T.run_product(args, graph=True)
    assert got_g == exp_g                      # graph topology, levels, labels, read counts: bit-exact
    assert got_fa == exp_fa                    # consensus FASTA: bit-exact
    T.compare_traces(open(tf).read(), exp_tr)  # per-level candidates identical, abundances to 1e-9


@pytest.mark.parametrize("seed", list(range(0, 12)))
def test_region_parity_with_varied_parameters(seed, tmp_path, oracle_bin):
    """-e -t -d -l -I -D -w -o varied (sc_testlib.param_case)."""
    d = str(tmp_path)
    args, _ = T.param_case(seed, d)
    exp_fa, exp_tr = T.run_oracle(args, d, trace=True)
    tf = os.path.join(d, "trace.txt")
    got_fa = T.run_product(args, trace_file=tf)
    assert got_fa == exp_fa
    T.compare_traces(open(tf).read(), exp_tr)


def test_msa_kernel_random(oracle_bin):
    from rambl_amd import capi
    rng = random.Random(5)
    with capi.Context(0, 1) as ctx:
        for it in range(60):
            n = rng.randint(2, 40)
            alpha = "ACGT" if it % 5 else "ACGTacgtN-"
            seqs = ["".join(rng.choice(alpha) for _ in range(rng.randint(1, 12))) for _ in range(n)]
            seqs.sort(key=len, reverse=True)
            assert ctx.msa_align(seqs) == T.oracle_msa(seqs), seqs


def test_config2_full_size_matches_reference_fasta(tmp_path):
    """BASELINE.json configs[1] at full size (10 000 x 150 bp reads, 1 500 bp gene):
    the consensus FASTA must equal the reference's own output (tests/golden/
    config2_full, produced by oracle/_ref in the build container), and a second
    run must reproduce it bit for bit."""
    import hashlib
    import json
    from rambl_amd import synth
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config2_full")
    meta = json.load(open(os.path.join(gold, "meta.json")))
    fa, sam, _ = synth.config2(str(tmp_path))
    assert hashlib.sha256(open(fa, "rb").read()).hexdigest() == meta["fasta_sha256"]
    assert hashlib.sha256(open(sam, "rb").read()).hexdigest() == meta["sam_sha256"]
    args = meta["argv"] + [fa, sam]
    got = T.run_product(args)
    assert got == open(os.path.join(gold, "expected.fa")).read()
    assert T.run_product(args) == got
    # domain property: the abundant contigs are the planted strains (generator truth), up to the window ends
    assert got.count(">") >= 3


def test_config4_deep_many_candidates_matches_reference_fasta(tmp_path):
    """BASELINE.json configs[3] scaled to 100 000 reads (50 strains, depth 10 000 thinned by -D 800): tens
    of candidate strains per level, so the wide variants of the sampler run.  The consensus FASTA must
    equal the reference's own output (tests/golden/config4_deep, produced by oracle/_ref in the build
    container)."""
    import hashlib
    import json
    from rambl_amd import synth
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config4_deep")
    if not os.path.exists(os.path.join(gold, "meta.json")):
        pytest.skip("fixture not generated (tests/golden/make_golden_large.py config4_deep)")
    meta = json.load(open(os.path.join(gold, "meta.json")))
    gene = synth.make_gene(4, glen=1500, n_strains=50, n_reads=100000, name="deep4")
    fa, sam = synth.write_dataset(str(tmp_path), [gene])
    assert hashlib.sha256(open(fa, "rb").read()).hexdigest() == meta["fasta_sha256"]
    assert hashlib.sha256(open(sam, "rb").read()).hexdigest() == meta["sam_sha256"]
    got = T.run_product(meta["argv"] + [fa, sam])
    assert got == open(os.path.join(gold, "expected.fa")).read()


def test_equal_abundance_tie_resolved_like_the_reference(tmp_path):
    """tests/golden/tie_case525 (sc_testlib.big_case(525): paired reads, a shared insertion site): pairs of
    candidate strains reach the end of the gene with equal abundances, and the reference tells them apart
    by an increment of 3e-17 on 113 -- visible only in its long double bookkeeping.  The merged contig must
    be represented by the same sequence as in the reference's own output."""
    import hashlib
    import json
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tie_case525")
    if not os.path.exists(os.path.join(gold, "meta.json")):
        pytest.skip("fixture not generated")
    meta = json.load(open(os.path.join(gold, "meta.json")))
    args, _ = T.big_case(525, str(tmp_path))
    fa, sam = args[-2], args[-1]
    assert args[:-2] == meta["argv"]
    assert hashlib.sha256(open(fa, "rb").read()).hexdigest() == meta["fasta_sha256"]
    assert hashlib.sha256(open(sam, "rb").read()).hexdigest() == meta["sam_sha256"]
    assert T.run_product(args) == open(os.path.join(gold, "expected.fa")).read()


def _golden_cases():
    import json
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    return sorted(json.load(open(os.path.join(gold, "index.json"))))


@pytest.mark.parametrize("name", _golden_cases())
def test_golden_reference_outputs(name, tmp_path):
    """The HIP path against outputs of the REFERENCE itself (committed fixtures):
    FASTA and graph dump bit-exact, per-level trace to 1e-9 (fp64 vs x87)."""
    from test_oracle_golden import load_case
    args, exp_fa, exp_g, exp_tr = load_case(name, str(tmp_path))
    tf = os.path.join(str(tmp_path), "p.trace")
    assert T.run_product(args, trace_file=tf) == exp_fa
    assert T.run_product(args, graph=True) == exp_g
    T.compare_traces(open(tf).read(), exp_tr)


@pytest.mark.parametrize("seed", [1, 7, 9, 15])
def test_edge_support_kernel(seed, tmp_path):
    """k_edge_support (row a16) against the supports the -G dump prints."""
    from rambl_amd import capi, cli
    args = T.make_case(seed, str(tmp_path))
    pa = cli.parse_cmd_line(args)
    regs = cli.load_regions(pa)
    params = capi.default_params(float(pa.error_rate), float(pa.tau), float(pa.diff_rate), want_graph=True)
    with capi.Context(0, 1) as ctx:
        for window, reads in regs:
            if len(reads) == 0:
                continue
            h = ctx.submit(reads, params)
            res = ctx.wait(h, want_graph=True, release=False)
            sup = ctx.edge_support(h)
            ctx.lib.sc_roi_release(ctx.h, h)
            edges = [l.split("\t") for l in res.graph.splitlines() if not l.startswith("#")]
            assert [int(e[2]) for e in edges] == sup


def test_regions_in_flight_are_independent(tmp_path):
    """Several regions on separate streams give the same FASTA as one at a time."""
    from rambl_amd import capi, cli, stage5
    prepared, single = [], []
    for seed in (0, 2, 4, 9):
        d = os.path.join(str(tmp_path), "s%d" % seed)
        args = T.make_case(seed, d)
        pa = cli.parse_cmd_line(args)
        prepared.append((pa, cli.load_regions(pa)))
        single.append(T.run_product(args))
    with capi.Context(0, 4) as ctx:
        texts, _ = stage5.run_regions(ctx, prepared, streams=4)
    assert texts == single


def test_more_regions_set_up_than_mailboxes(tmp_path, monkeypatch):
    """A resident context whose regions outnumber its mailboxes (three workgroups for eight region slots: SC_RESIDENT_SLOTS):
    a region whose set-up is done waits in line for a mailbox, takes over the one a finished region hands on, and its
    levels carry that mailbox's stamps.  Sixteen regions through it, each equal to its FASTA from a run of its own."""
    from rambl_amd import capi, cli, stage5
    prepared, single = [], []
    for seed in (0, 2, 4, 9, 11, 13, 5, 8):
        d = os.path.join(str(tmp_path), "s%d" % seed)
        args = T.make_case(seed, d)
        pa = cli.parse_cmd_line(args)
        prepared.append((pa, cli.load_regions(pa)))
        single.append(T.run_product(args))
    monkeypatch.setenv("SC_RESIDENT", "1")
    monkeypatch.setenv("SC_RESIDENT_SLOTS", "3")
    errors = []
    with capi.Context(0, 8) as ctx:
        texts, stats = stage5.run_regions(ctx, prepared * 2, 8, None, errors)
    assert errors == []
    assert texts == single * 2
    assert sum(s["mailbox_ms"] for s in stats) > 0                      # somebody did wait for a mailbox


def test_unthinned_deep_coverage_no_sweeps(tmp_path, oracle_bin):
    """More than 40 000 read copies per level: the sweep count min(5000, 40000/copies) is 0
    (NonparametricClustering.cpp:160) and np_bayes_clustering degenerates; -D large keeps every read."""
    from rambl_amd import synth
    d = str(tmp_path)
    gene = synth.make_gene(77, glen=170, n_strains=2, n_reads=42000, rlen=150, err=0.0002, n_sub=3, n_ins=0, n_del=0,
                           name="deep")
    fa, sam = synth.write_dataset(d, [gene])
    args = ["-r", "deep:1-170", "-q", "0", "-D", "1000000", "-I", "13", "-l", "70", "-t", "0.02", "-d", "0.02", "-w", "5000", fa, sam]
    exp_fa, exp_tr = T.run_oracle(args, d, trace=True)
    tf = os.path.join(d, "p.trace")
    assert T.run_product(args, trace_file=tf) == exp_fa
    T.compare_traces(open(tf).read(), exp_tr)


def _with_codes_in_gene(args, codes, n, seed):
    """Overwrite n positions of the gene FASTA of a generated case with the given symbols."""
    fa = args[-2]
    lines = open(fa).read().split("\n")
    idx = [i for i, l in enumerate(lines) if l and not l.startswith(">")]
    seq = list("".join(lines[i] for i in idx))
    rng = random.Random(seed)
    for k in range(n):
        seq[rng.randrange(len(seq))] = codes[k % len(codes)]
    seq = "".join(seq)
    k = 0
    for i in idx:
        m = len(lines[i])
        lines[i] = seq[k:k + m]
        k += m
    open(fa, "w").write("\n".join(lines))


@pytest.mark.parametrize("seed", [2, 5, 9, 14])
def test_iupac_codes_in_the_gene(seed, tmp_path, oracle_bin):
    """GreenGenes references carry R Y K M S W N: symbols outside {A,C,G,T,-,=}.  The reference's Strain tables are
    std::map-keyed and take any byte (Strain.cpp:127-150: a never-set count is created as 0 -> log 0); the device
    tables are [K][K] over the symbols of the window, up to 16.  A gene with seven different codes (13 symbols in
    the labels); the oracle was checked against the reference itself on exactly these inputs (FASTA and 17-digit
    trace byte for byte).  Reads that carry such codes themselves crash the reference (segmentation fault): no parity
    to speak of, not tested."""
    d = str(tmp_path)
    args = T.make_case(seed, d)
    _with_codes_in_gene(args, "RYKMSWN", 14, seed)
    exp_fa, exp_tr = T.run_oracle(args, d, trace=True)
    exp_g, _ = T.run_oracle(args, d, graph=True)
    tf = os.path.join(d, "trace.txt")
    assert T.run_product(args, graph=True) == exp_g
    assert T.run_product(args, trace_file=tf) == exp_fa
    T.compare_traces(open(tf).read(), exp_tr)


def test_more_than_sixteen_symbols_is_reported_not_guessed(tmp_path):
    """Labels with more than 16 distinct symbols are outside the device model: SC_ERR_UNSUPPORTED."""
    from rambl_amd import capi
    d = str(tmp_path)
    args = T.make_case(3, d)
    _with_codes_in_gene(args, "RYKMSWBDHVN", 22, 3)       # 6 + 11 symbols
    with pytest.raises(capi.StrainCallError) as e:
        T.run_product(args)
    assert e.value.code == -4


def test_process_boundary_bin_straincall(tmp_path, oracle_bin):
    """bin/StrainCall as rambl.py launches it: argv in, FASTA on stdout, nothing else."""
    import subprocess
    import sys
    d = str(tmp_path)
    args = T.make_case(8, d)
    exp_fa, _ = T.run_oracle(args, d)
    p = subprocess.run([sys.executable, os.path.join(T.ROOT, "bin", "StrainCall")] + args, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-1000:]
    assert p.stdout.decode() == exp_fa


def test_stage5_many_regions_one_gpu(tmp_path, oracle_bin):
    """rambl.py stage 5 on one FASTA/SAM with several seed genes (BASELINE configs[2] in small):
    per-region FASTA, concatenation in .fai order -- equal to running the oracle per region."""
    from rambl_amd import stage5, synth
    d = str(tmp_path)
    genes = [synth.make_gene(300 + k, glen=260 + 20 * k, n_strains=1 + k % 3, n_reads=120 + 30 * k, rlen=110, err=0.004,
                             n_sub=5, n_ins=k % 2, n_del=(k + 1) % 2, name="otu%d" % k) for k in range(5)]
    fa, sam = synth.write_dataset(d, genes)
    expected = ""
    for roi in stage5.roi_list(fa + ".fai"):
        out, _ = T.run_oracle(stage5.straincall_argv(roi, fa, sam), d)
        expected += out
    got = stage5.strain_call(fa, sam, out_dir=os.path.join(d, "work"), prefix="rambl", streams=3)
    assert got == expected
    assert open(os.path.join(d, "work", "rambl.fa")).read() == expected
    for roi in stage5.roi_list(fa + ".fai"):
        assert os.path.exists(os.path.join(d, "work", "3_straincall_results", "%s.fa" % roi))


@pytest.mark.gpu
@pytest.mark.parametrize("stretch_words", [0, 512])
def test_thread_kernels_wide_classes(stretch_words, tmp_path):
    """Deep coverage: 60 000 reads over a 300-base gene, so the reads of a class span far more ids than a wavefront's bitmap
    holds (16 384) and the pools are put into read order by k_thread_sort_big -- with the full stretch of half a million ids,
    and, in a child process, with stretches of 16 384 ids so that a class needs several.  Expected pools: the per-base M loop
    (PartialOrderGraph.cpp:129-177) restated with numpy."""
    import subprocess
    import sys
    if stretch_words:
        env = dict(os.environ, SC_SORT_BIG_WORDS=str(stretch_words), SC_WIDE_CHILD="1")
        p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "%s::test_thread_kernels_wide_classes" % os.path.abspath(__file__),
                            "-k", "0]"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        assert p.returncode == 0, p.stdout.decode()[-3000:]
        return
    import numpy as np
    import py_ingest_mirror as mirror
    from rambl_amd import capi, cli, stage5, synth
    g = synth.make_gene(77, glen=300, n_strains=4, n_reads=60000, name="wide77")
    fa, sam = synth.write_dataset(str(tmp_path), [g])
    pa = cli.parse_cmd_line(["-r", "wide77:1-300", "-q", "0", "-D", "1000000", "-I", "13", "-l", "70", "-t", "0.02", "-d", "0.02", "-w", "5000", fa, sam])
    params = capi.default_params(float(pa.error_rate), float(pa.tau), float(pa.diff_rate), graph_only=True)
    (window, reads), = cli.load_regions(pa)
    assert len(reads) > 30000
    with capi.Context(0, 1) as ctx:
        h = ctx.submit(reads, params)
        ctx.wait(h, release=False)
        cnt, first, pool, sym = ctx.thread_tables(h)
        ctx.lib.sc_roi_release(ctx.h, h)
    code = np.full(256, 255, dtype=np.int64)
    for k, ch in enumerate(sym):
        if ch:
            code[ch] = k
    cls_parts, rid_parts = [], []
    for rid in range(len(reads)):
        i, j = reads.pos[rid], 0
        sq = np.frombuffer(reads.seq[rid].encode("ascii"), dtype=np.uint8)
        for op, ln in mirror.parse_cigar(reads.cigar[rid]):
            if op == "M":
                cls_parts.append((i + np.arange(ln)) * 8 + code[sq[j:j + ln]])
                rid_parts.append(np.full(ln, rid, dtype=np.int64))
                i += ln
                j += ln
            elif op == "I":
                j += ln
            elif op == "D":
                i += ln
    cls = np.concatenate(cls_parts)
    rids = np.concatenate(rid_parts)
    order = np.lexsort((rids, cls))
    glen = len(reads.gene_seq)
    exp_cnt = np.bincount(cls, minlength=glen * 8)
    got_cnt = np.array(cnt[:], dtype=np.int64)
    assert (got_cnt == exp_cnt).all()
    got_pool = np.array(pool[:], dtype=np.int64)
    assert got_pool.shape == rids.shape
    assert (got_pool == rids[order]).all()                        # every pool in ascending read order
    span = np.array([rids[order][a:b].max() - rids[order][a:b].min() for a, b in zip(np.cumsum(exp_cnt) - exp_cnt, np.cumsum(exp_cnt)) if b > a])
    assert (span >= 16384).sum() > 100                              # the wide path was the one that ran


@pytest.mark.parametrize("seed", [1, 3, 7, 13])
def test_thread_kernels_class_tables(seed, tmp_path):
    """k_thread_* (row a5) against a plain restatement of the per-base M loop
    (PartialOrderGraph.cpp:129-177): class sizes, first read per class, pools in read order."""
    import py_ingest_mirror as mirror
    from rambl_amd import capi, cli
    args = T.make_case(seed, str(tmp_path))
    pa = cli.parse_cmd_line(args)
    params = capi.default_params(float(pa.error_rate), float(pa.tau), float(pa.diff_rate), graph_only=True)
    with capi.Context(0, 1) as ctx:
        for window, reads in cli.load_regions(pa):
            if len(reads) == 0:
                continue
            h = ctx.submit(reads, params)
            ctx.wait(h, release=False)
            cnt, first, pool, sym = ctx.thread_tables(h)
            ctx.lib.sc_roi_release(ctx.h, h)
            code = {ch: k for k, ch in enumerate(sym) if ch}
            glen = len(reads.gene_seq)
            exp = {}
            for rid in range(len(reads)):
                i, j = reads.pos[rid], 0
                for op, ln in mirror.parse_cigar(reads.cigar[rid]):
                    if op == "M":
                        for t in range(ln):
                            exp.setdefault((i + t) * 8 + code[ord(reads.seq[rid][j + t])], []).append(rid)
                        i += ln
                        j += ln
                    elif op == "I":
                        j += ln
                    elif op == "D":
                        i += ln
            assert len(cnt) == glen * 8
            off = 0
            for cls in range(glen * 8):
                members = exp.get(cls, [])
                assert cnt[cls] == len(members)
                assert pool[off:off + cnt[cls]] == members          # ascending read ids
                if members:
                    assert first[cls] == members[0]
                off += cnt[cls]
            assert off == len(pool)


def test_msa_kernel_wider_than_lds(oracle_bin):
    """Hundreds of unrelated insertion strings at one site: the reference's scoring keeps opening columns,
    the alignment outgrows the 1 024 columns kept in LDS and k_msa<true> (state in HBM) takes over."""
    from rambl_amd import capi
    rng = random.Random(23)
    seqs = ["".join(rng.choice("ACGT") for _ in range(rng.randint(4, 11))) for _ in range(260)]
    seqs.sort(key=len, reverse=True)
    exp = T.oracle_msa(seqs)
    assert len(exp[0]) > 1024, len(exp[0])
    with capi.Context(0, 1) as ctx:
        assert ctx.msa_align(seqs) == exp


def test_msa_kernel_deep_and_long(oracle_bin):
    """k_msa with hundreds of rows (deep coverage at an indel hot spot) and with insertions shorter and longer than the
    64 DP columns one pass of the wavefront holds."""
    import time
    from rambl_amd import capi
    rng = random.Random(17)
    with capi.Context(0, 1) as ctx:
        # an indel hot spot at depth 600: three insertion alleles plus sequencing errors
        alleles = ["ACGTTGCA", "ACGTA", "GT"]

        def noisy(a):
            out = []
            for ch in a:
                r = rng.random()
                if r < 0.03:
                    continue
                out.append(rng.choice("ACGT") if r < 0.06 else ch)
                if rng.random() < 0.02:
                    out.append(rng.choice("ACGT"))
            return "".join(out) or "A"
        seqs = [noisy(rng.choice(alleles)) for _ in range(600)]
        seqs.sort(key=len, reverse=True)
        t0 = time.time()
        got = ctx.msa_align(seqs)
        t_gpu = time.time() - t0
        t0 = time.time()
        exp = T.oracle_msa(seqs)
        t_cpu = time.time() - t0
        assert got == exp
        print("msa 600 rows: gpu %.3f s, oracle %.3f s" % (t_gpu, t_cpu))
        seqs = ["".join(rng.choice("ACGT") for _ in range(rng.choice([63, 40, 17, 5, 2]))) for _ in range(12)]
        seqs.sort(key=len, reverse=True)
        assert ctx.msa_align(seqs) == T.oracle_msa(seqs)
        # insertions of 64 bases and more (-I is the user's to raise): the DP columns run in chunks of 64 lanes
        for lens in ([64, 10], [65, 64, 63, 3], [200, 130, 129, 128, 127, 64, 1], [300, 299, 7, 7, 2]):
            seqs = ["".join(rng.choice("ACGT") for _ in range(n)) for n in lens]
            assert ctx.msa_align(seqs) == T.oracle_msa(seqs), lens
        related = "".join(rng.choice("ACGT") for _ in range(150))
        seqs = sorted([related, related[:100] + "TT" + related[100:], related[5:140], related[:70] + related[75:]], key=len, reverse=True)
        assert ctx.msa_align(seqs) == T.oracle_msa(seqs)


@pytest.mark.parametrize("seed", [3, 16, 31])
def test_reference_with_ambiguity_codes(seed, tmp_path, oracle_bin):
    """N in the gene (16S references carry ambiguity codes): a seventh symbol in the node labels, and
    the 'N in the strain label matches anything' rule of NonparametricClustering.cpp:358,372,384."""
    import random
    d = str(tmp_path)
    args = T.make_case(seed, d)
    fa = args[-2]
    lines = open(fa).read().split("\n")
    idx = [i for i, l in enumerate(lines) if l and not l.startswith(">")]
    seq = list("".join(lines[i] for i in idx))
    rng = random.Random(seed)
    for _ in range(6):
        seq[rng.randrange(len(seq))] = "N"
    seq = "".join(seq)
    k = 0
    for i in idx:
        n = len(lines[i])
        lines[i] = seq[k:k + n]
        k += n
    open(fa, "w").write("\n".join(lines))
    exp_fa, exp_tr = T.run_oracle(args, d, trace=True)
    tf = os.path.join(d, "trace.txt")
    got = T.run_product(args, trace_file=tf)
    assert got == exp_fa
    T.compare_traces(open(tf).read(), exp_tr)


def _sha(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def test_config3_hundred_regions_match_reference(tmp_path):
    """BASELINE.json configs[2] at the shape SURVEY.md section 8(d) gives it: 100 seed genes (seeds 100-199, 2 000-10 000
    reads each, 575 103 alignments) in ONE FASTA + ONE SAM, run the way scripts/rambl.py:169-201 runs stage 5 -- one
    StrainCall per `name:1-len` of the .fai, outputs concatenated in .fai order.  Every region's FASTA must equal the
    reference's own stdout for that region (tests/golden/config3_regions, produced by oracle/_ref in the build
    container, 7 h of CPU), and so must the concatenation."""
    import gzip
    import json
    from rambl_amd import stage5, synth
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config3_regions")
    meta = json.load(open(os.path.join(gold, "meta.json")))
    expected = json.loads(gzip.open(os.path.join(gold, "expected.json.gz")).read())
    d = str(tmp_path)
    fa, sam, _ = synth.config3(d)
    assert _sha(fa) == meta["fasta_sha256"] and _sha(sam) == meta["sam_sha256"]
    rois = stage5.roi_list(fa + ".fai")
    assert rois == [r for r, _ in expected] and len(rois) == 100
    errors = []
    work = os.path.join(d, "work")
    full = stage5.strain_call(fa, sam, out_dir=work, prefix="rambl", streams=32, ingest_workers=4, errors=errors)
    assert errors == []
    bad = [r for r, text in expected if open(os.path.join(work, "3_straincall_results", "%s.fa" % r)).read() != text]
    assert bad == []
    assert full == "".join(text for _, text in expected)
    assert full.count(">") == meta["contigs"]


def test_mixed_kinds_in_one_batch_match_the_fixtures(tmp_path):
    """Regions of very different shapes in flight together, so that one launch of the level server carries levels that
    need different variants of the level kernel (k_level_any): the 50-strain region of config4_deep (up to 100+
    candidates per level: the widest sampler variants, weight rows in HBM) next to the small golden cases (a few
    candidates, levels without sampler, a gene with IUPAC codes).  Every region's FASTA must equal the reference's own
    output for it, whoever it shared its launches with."""
    import hashlib
    import json
    from rambl_amd import capi, cli, stage5, synth
    from test_oracle_golden import load_case
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    prepared, expected = [], []
    meta = json.load(open(os.path.join(gold, "config4_deep", "meta.json")))
    gene = synth.make_gene(4, glen=1500, n_strains=50, n_reads=100000, name="deep4")
    fa, sam = synth.write_dataset(str(tmp_path / "deep"), [gene])
    assert hashlib.sha256(open(sam, "rb").read()).hexdigest() == meta["sam_sha256"]
    for rep in range(2):                                       # twice: the wide variants also meet each other
        pa = cli.parse_cmd_line(meta["argv"] + [fa, sam])
        prepared.append((pa, cli.load_regions(pa)))
        expected.append(open(os.path.join(gold, "config4_deep", "expected.fa")).read())
    for k, name in enumerate(_golden_cases()):
        d = tmp_path / ("g%d" % k)
        d.mkdir()
        args, exp_fa, _, _ = load_case(name, str(d))
        pa = cli.parse_cmd_line(args)
        prepared.append((pa, cli.load_regions(pa)))
        expected.append(exp_fa)
    errors = []
    with capi.Context(0, len(prepared)) as ctx:
        texts, _ = stage5.run_regions(ctx, prepared, len(prepared), None, errors)
    assert errors == []
    assert [i for i, (g, e) in enumerate(zip(texts, expected)) if g != e] == []


@pytest.fixture(scope="module")
def million_reads(tmp_path_factory):
    """BASELINE.json configs[3] at its stated size: 1 000 000 x 150 bp reads, 50 strains, one 1 500 bp gene."""
    from rambl_amd import synth
    d = str(tmp_path_factory.mktemp("deep4m"))
    gene = synth.make_gene(4, glen=1500, n_strains=50, n_reads=1000000, name="deep4m")
    fa, sam = synth.write_dataset(d, [gene])
    return fa, sam


@pytest.mark.parametrize("depth", [800, 3000])
def test_config4_million_reads_match_reference(depth, million_reads):
    """The 10^6-read region through the native ingest (view, depth -> keep probability, mt19937 thinning of a million
    candidates, duplicate collapse) and the device path: -D 800 (rambl.py's value: ~8 000 reads reach the graph) and a
    raised -D 3000 (~30 000 reads, 3 000 read copies per level, 13 sweeps, weight rows in HBM).  FASTA equal to the
    reference's own stdout (tests/golden/config4_full_D*, oracle/_ref in the build container; -D 3000 is the largest
    depth the reference's quadratic edge support finishes in an hour -- tests/golden/make_golden_config4.py)."""
    import json
    fa, sam = million_reads
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config4_full_D%d" % depth)
    meta = json.load(open(os.path.join(gold, "meta.json")))
    assert _sha(fa) == meta["fasta_sha256"] and _sha(sam) == meta["sam_sha256"]
    got = T.run_product(meta["argv"] + [fa, sam])
    assert got == open(os.path.join(gold, "expected.fa")).read()


def test_grid_levels_on_resident_workers(million_reads):
    """Levels with more than 131 072 (candidate, read) items send their row copies and log-likelihood update to a grid of
    their own (k_level_copy, k_level_update) before the level's workgroup takes over.  With several regions in flight that
    workgroup is a RESIDENT one: it has to see what kernels on other CUs (other XCDs) have just written to rows it read at
    the level before.  The -D 3000 region (3 000 read copies per level: grid updates from ~45 candidates on) twice in a
    two-slot context, both against the reference's FASTA."""
    import json
    from rambl_amd import capi, cli
    fa, sam = million_reads
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config4_full_D3000")
    meta = json.load(open(os.path.join(gold, "meta.json")))
    pa = cli.parse_cmd_line(meta["argv"] + [fa, sam])
    regions = cli.load_regions(pa)
    params = capi.default_params(float(pa.error_rate), float(pa.tau), float(pa.diff_rate))
    exp = open(os.path.join(gold, "expected.fa")).read()
    with capi.Context(0, 2) as ctx:
        hs = [[(w, ctx.submit(r, params)) for w, r in regions] for _ in range(2)]
        for handles in hs:
            assert "".join(cli.format_fasta(w, ctx.wait(h), pa.tau) for w, h in handles) == exp


def test_config4_million_reads_unthinned_matches_oracle(million_reads):
    """configs[3] with no thinning at all (-D 100000): every one of the 10^6 reads reaches the graph (~590 000 distinct
    reads, ~100 000 read copies per level: the closed-form levels, grid updates, 600 MB of pools).  The reference cannot
    finish this case (quadratic edge support); the expected FASTA is the ORACLE's, run with its counting edge support
    (tests/golden/config4_full_D100000/meta.json says how)."""
    import json
    fa, sam = million_reads
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config4_full_D100000")
    meta = json.load(open(os.path.join(gold, "meta.json")))
    assert _sha(fa) == meta["fasta_sha256"] and _sha(sam) == meta["sam_sha256"]
    got = T.run_product(meta["argv"] + [fa, sam])
    assert got == open(os.path.join(gold, "expected.fa")).read()


@pytest.mark.parametrize("seed", [2, 6])
def test_region_from_bam_matches_oracle(seed, tmp_path, oracle_bin, monkeypatch):
    """BAM in (read by the library's own BGZF / BAM decoder: no samtools in the image), FASTA out: equal to the oracle
    on the SAM text the BAM was written from (paired reads; several scan windows)."""
    d = str(tmp_path)
    args = T.make_case(seed, d)
    exp_fa, _ = T.run_oracle(args, d)
    bam = os.path.join(d, "reads.bam")
    T.write_bam(args[-1], bam)
    assert T.run_product(args[:-1] + [bam]) == exp_fa


def test_msa_kernel_on_the_reference_vectors():
    """k_msa against the reference itself, with nothing in between: every case of tests/golden/msa_vectors.json.gz (rows
    printed by the reference's own MultipleSequenceAlignmentSP::align through oracle/_ref/msa_ref in the build
    container, MultipleSequenceAlignmentSP.cpp:10-301)."""
    import gzip
    import json
    from rambl_amd import capi
    cases = json.loads(gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "msa_vectors.json.gz")).read())
    assert len(cases) >= 100
    with capi.Context(0, 1) as ctx:
        for c in cases:
            rows = ctx.msa_align(c["seqs"])
            assert rows == c["rows"], c["seqs"]
            assert len(rows[0]) == c["ncol"]


def test_every_wide_sampler_variant_is_exercised(tmp_path):
    """The sampler kernel has one variant per 16 candidates (NB = 1..8) and per home of the weight rows (LDS / HBM).  Under
    the reference's cap of 80 candidates (NonparametricClustering.cpp:532-551) no committed data set goes beyond NB = 5
    (config4_deep: 72 candidates at most); tests/golden/wide_cap120 is a region of 100 strains walked with the cap raised
    to 120 (sc_params.max_candidates; SC_ORACLE_MAX_CANDIDATES for the oracle, whose output the fixture holds): up to 122
    candidates at a level.  The FASTA must equal the fixture; at every trace block the candidates (number, sequences in
    order) must be the oracle's and their abundances must add up to the oracle's sum to 1e-9; and the histogram of
    sc_stats must show every variant NB = 1..8 at work -- so the wide variants cannot lose their coverage unnoticed."""
    import gzip
    import hashlib
    import json
    from rambl_amd import capi, cli, synth
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wide_cap120")
    meta = json.load(open(os.path.join(gold, "meta.json")))
    gene = synth.make_gene(77, glen=600, n_strains=100, n_reads=30000, name="wide", n_sub=12, err=0.01)
    fa, sam = synth.write_dataset(str(tmp_path), [gene])
    assert hashlib.sha256(open(sam, "rb").read()).hexdigest() == meta["sam_sha256"]
    assert hashlib.sha256(open(fa, "rb").read()).hexdigest() == meta["fasta_sha256"]
    pa = cli.parse_cmd_line(meta["argv"] + [fa, sam])
    regions = cli.load_regions(pa)
    params = capi.default_params(float(pa.error_rate), float(pa.tau), float(pa.diff_rate), want_trace=True)
    params.max_candidates = meta["max_candidates"]
    hist = [0] * 17
    text, blocks = "", []
    with capi.Context(0, 1) as ctx:
        for w, r in regions:
            res = ctx.wait(ctx.submit(r, params), want_trace=True)
            text += cli.format_fasta(w, res, pa.tau)
            hist = [x + y for x, y in zip(hist, res.stats["kind_levels"])]
            for when, level, rows in T.parse_trace(res.trace):
                h = hashlib.sha1()
                for seq, _ in rows:
                    h.update(seq.encode())
                    h.update(b"\n")
                blocks.append([when, level, len(rows), sum(ab for _, ab in rows), h.hexdigest()[:16]])
    assert text == open(os.path.join(gold, "expected.fa")).read()
    exp = json.loads(gzip.open(os.path.join(gold, "expected_levels.json.gz")).read())
    assert len(blocks) == len(exp) == meta["trace_blocks"]
    for g, e in zip(blocks, exp):
        assert g[:3] == e[:3] and g[4] == e[4], (g, e)
        tot = float(e[3])
        assert abs(g[3] - tot) <= 1e-9 * max(abs(tot), 1e-300), (g, e)
    assert max(b[2] for b in blocks) == meta["max_candidates_at_a_level"] > 112
    by_nb = {nb: hist[1 + 2 * (nb - 1)] + hist[2 + 2 * (nb - 1)] for nb in range(1, 9)}
    assert all(by_nb[nb] > 0 for nb in range(1, 9)), by_nb


def test_launch_per_level_path_with_many_regions_in_flight():
    """Several regions in flight run on resident level workers by default; the other way of running their levels -- one launch
    per level through the level server, levels of different kinds leaving as one grid (k_level_any) -- stays in the library
    (SC_RESIDENT=0; a single region always uses a launch per level) and must keep giving the reference's results.  The
    tests that put regions of very different shapes in flight together are run again that way, in a child process with its
    own time limit (a hang of either path ends there, not in this process)."""
    import subprocess
    import sys
    if os.environ.get("SC_TEST_CHILD"):
        pytest.skip("already the child")
    env = dict(os.environ, SC_RESIDENT="0", SC_TEST_CHILD="1")
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-k",
                        "mixed_kinds or regions_in_flight_are_independent or stage5_many_regions"], env=env, timeout=900,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    tail = p.stdout.decode()[-1500:]
    assert p.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail
