"""CPU: host-side pieces of the product path against the oracle -- Python ingest
(rows a1-a4), the C++ graph builder (a5, a6, a9-a11; MSA callback = oracle), the
libstdc++ sort port, the mt19937 stream, the CLI surface."""
import ctypes
import io
import os
import random
import subprocess

import numpy as np
import pytest

import sc_testlib as T
import py_ingest_mirror as mirror          # the Python restatement of the read path (test infrastructure)
from rambl_amd import cli, ingest

ROOT = T.ROOT


def _dump_reads(args, d):
    base = os.path.join(d, "dump")
    T.run_oracle(args, d, graph=True, dump_reads=base)
    out = []
    i = 0
    while os.path.exists("%s.%d" % (base, i)):
        out.append(open("%s.%d" % (base, i)).read().splitlines())
        i += 1
    return out, base


def test_ingest_sorts_on_several_threads_match_oracle(tmp_path, oracle_bin):
    """sc_aln_load_reads sorts a deep region's reads on the ingest threads (stretches sorted side by side, merged pairwise):
    the cases of test_ingest_matches_oracle again in a child process whose stretches are 50 reads long and whose merges
    therefore all run."""
    import subprocess
    import sys
    env = dict(os.environ, SC_INGEST_SORT_MIN="50", SC_INGEST_THREADS="5")
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", os.path.abspath(__file__), "-k", "test_ingest_matches_oracle"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    assert p.returncode == 0, p.stdout.decode()[-3000:]
    assert b"12 passed" in p.stdout


@pytest.mark.parametrize("seed", list(range(0, 12)))
def test_ingest_matches_oracle(seed, tmp_path, oracle_bin):
    d = str(tmp_path)
    args = T.make_case(seed, d)
    dumps, _ = _dump_reads(args, d)
    pa = cli.parse_cmd_line(args)
    regs = cli.load_regions(pa)
    assert len(regs) == len(dumps)
    for (w, r), lines in zip(regs, dumps):
        assert lines[0].split("\t")[1:] == [w[0], str(w[1]), str(w[2])]
        assert lines[1].split("\t")[1] == r.gene_seq
        exp = [l.split("\t") for l in lines[2:]]
        assert len(exp) == len(r)
        for i, e in enumerate(exp):
            got = [str(r.pos[i]), r.cigar[i], r.seq[i], str(r.copies[i]), ",".join(map(str, r.mates[i]))]
            assert got == e[1:6]


@pytest.fixture(scope="session")
def graph_check_bin(oracle_bin, tmp_path_factory):
    out = str(tmp_path_factory.mktemp("native") / "graph_host_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", out, os.path.join(ROOT, "tests", "native", "graph_host_check.cpp"),
                           os.path.join(ROOT, "rambl_amd", "csrc", "sc_graph.cpp"), "-L" + os.path.join(ROOT, "oracle"),
                           "-loracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    return out


@pytest.mark.parametrize("seed", list(range(0, 16)))
def test_host_graph_matches_oracle(seed, tmp_path, oracle_bin, graph_check_bin):
    d = str(tmp_path)
    args = T.make_case(seed, d)
    base = os.path.join(d, "dump")
    exp_g, _ = T.run_oracle(args, d, graph=True, dump_reads=base)
    got = ""
    i = 0
    while os.path.exists("%s.%d" % (base, i)):
        lines = open("%s.%d" % (base, i)).read().splitlines()
        if len(lines) > 2:
            p = subprocess.run([graph_check_bin, "%s.%d" % (base, i)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert p.returncode == 0, p.stderr.decode()
            assert "unsupported ''" in p.stderr.decode()
            got += p.stdout.decode()
        i += 1
    assert got == exp_g


def test_sort_port_matches_libstdcxx(tmp_path, oracle_bin):
    """oracle's std::sort port vs the real libstdc++ std::sort on keys full of ties."""
    src = tmp_path / "s.cpp"
    src.write_text(r'''
#include <algorithm>
#include <cstdio>
#include <vector>
int main() { int n; while (scanf("%d", &n) == 1) { std::vector<double> k(n); for (auto& x : k) scanf("%lf", &x);
  std::vector<int> p(n); for (int i = 0; i < n; i++) p[i] = i;
  std::sort(p.begin(), p.end(), [&](int a, int b) { return k[a] > k[b]; });
  for (int i = 0; i < n; i++) printf("%d ", p[i]); printf("\n"); } }
''')
    exe = str(tmp_path / "s")
    subprocess.check_call(["g++", "-O1", "-o", exe, str(src)])
    lib = T.oracle_lib()
    rng = random.Random(3)
    text, cases = [], []
    for _ in range(300):
        n = rng.choice([1, 2, 5, 16, 17, 18, 33, 80, 81, 200, 1000])
        keys = [float(rng.randint(0, max(1, n // rng.choice([1, 2, 8])))) for _ in range(n)]
        cases.append(keys)
        text.append("%d %s" % (n, " ".join(map(repr, keys))))
    out = subprocess.run([exe], input="\n".join(text).encode(), stdout=subprocess.PIPE).stdout.decode().splitlines()
    for keys, line in zip(cases, out):
        n = len(keys)
        perm = (ctypes.c_int * n)()
        lib.oracle_sort_desc_perm((ctypes.c_double * n)(*keys), n, perm)
        assert list(perm) == [int(x) for x in line.split()]


def test_mt19937_canonical_stream(oracle_bin):
    lib = T.oracle_lib()
    n = 3000
    out = (ctypes.c_double * n)()
    lib.oracle_mt_canonical(1234, n, out)
    gen = mirror.MT19937(1234)
    assert [gen.canonical() for _ in range(n)] == list(out)
    # std::mt19937 known answer: the 10000th output of the default-seeded engine is 4123659995
    g = mirror.MT19937(5489)
    v = 0
    for _ in range(10000):
        v = g.next_u32()
    assert v == 4123659995


def test_cli_argv_forms():
    pa = cli.parse_cmd_line(["-r", "g:1-10", "--window", "77", "-overlap", "5", "-e", "0.5", "-q", "7", "-D", "9",
                             "-I", "3", "-l", "44", "-t", "0.25", "--diff-rate", "0.125", "-G", "--bogus", "a.fa", "b.bam", "c.bam"])
    assert (pa.roi, pa.window_size, pa.overlap_size, pa.mapping_qual, pa.max_depth, pa.max_ins, pa.read_len) == \
        ("g:1-10", 77, 5, 7, 9, 3, 44)
    assert float(pa.error_rate) == 0.5 and float(pa.tau) == 0.25 and float(pa.diff_rate) == 0.125
    assert pa.plot_graph and pa.gene_file == "a.fa" and pa.mapping_file == "c.bam"
    assert float(cli.parse_cmd_line(["-t", "0.02"]).tau) == float(np.float32(0.02))
    assert ingest.stoi("12abc") == 12
    with pytest.raises(ValueError):
        ingest.stoi("abc")
    assert ingest.gene_roi_end_pos("g:1-9") == 9
    with pytest.raises(ValueError):            # first '-' of the WHOLE roi (StrainCall.cpp:210-220): stoi("b:1-9") throws
        ingest.gene_roi_end_pos("a-b:1-9")


def test_cli_help_goes_to_stderr():
    out, err = io.StringIO(), io.StringIO()
    assert cli.main([], out=out, err=err) == 0
    assert out.getvalue() == "" and err.getvalue().startswith("StrainCall marker_gene read_mapping")
    out, err = io.StringIO(), io.StringIO()
    assert cli.main(["-h", "x.fa", "y.sam"], out=out, err=err) == 0
    assert "-G,--plot-graph    print graph" in err.getvalue()


def test_crop_and_cigar_helpers():
    assert mirror.parse_cigar("3S10M2I4D5=6X") == [("S", 3), ("M", 10), ("I", 2), ("D", 4), ("M", 5), ("M", 6)]
    cig = mirror.parse_cigar("10M")
    assert mirror.crop_read_within_window(5, 8, "ACGTACGTAC", "IIIIIIIIII", cig, 3, 12) == ("GTAC", "4M")
    assert mirror.crop_read_within_window(1, 100, "ACGTACGTAC", "IIIIIIIIII", cig, 3, 12) == ("ACGTACGTAC", "10M")
    assert mirror.max_insert_size("5M3I2M7I1M") == 7


@pytest.mark.parametrize("seed", [0, 1, 2, 5, 6, 13])
def test_pileup_flags_equal_pileup_text(seed, tmp_path):
    """The array-based pileup summary equals the summary of the emulated pileup text."""
    from rambl_amd import samio
    args = T.make_case(seed, str(tmp_path))
    aln = samio.Alignments(args[-1])
    text = mirror.SamText(args[-1])
    name = "g%d" % seed
    for region in ("%s:1-5000" % name, "%s:40-180" % name, "%s:200-260" % name):
        for mq in (0, 50):
            assert aln.pileup_flags(mq, region) == mirror.flags_from_pileup_text(text.mpileup(mq, region)) == text.pileup_flags(mq, region)


@pytest.mark.parametrize("seed", [2, 4])
def test_native_bam_reader_gives_the_same_regions(seed, tmp_path, monkeypatch):
    """BAM input read natively (no samtools in the image) yields the same ingest as the SAM text."""
    d = str(tmp_path)
    args = T.make_case(seed, d)
    bam = os.path.join(d, "reads.bam")
    T.write_bam(args[-1], bam)
    from rambl_amd import samio
    assert samio.is_bam(bam)
    a = cli.load_regions(cli.parse_cmd_line(args))
    b = cli.load_regions(cli.parse_cmd_line(args[:-1] + [bam]))
    assert len(a) == len(b)
    for (wa, ra), (wb, rb) in zip(a, b):
        assert wa == wb and ra.gene_seq == rb.gene_seq
        assert (ra.pos, ra.cigar, ra.seq, ra.copies, ra.mates) == (rb.pos, rb.cigar, rb.seq, rb.copies, rb.mates)


def _random_sam(rng, path, genes=("gA", "gB"), n=400, glen=400):
    """SAM records with every CIGAR shape the ingest has a rule for: soft / hard clips, insertions and deletions at
    and across window edges, = and X, N skips, reads hanging over both ends of a window, duplicates, mates by flag
    and by name, repeated names, filtered flags, low mapping qualities, short reads, N bases."""
    lines = ["@SQ\tSN:%s\tLN:%d" % (g, glen) for g in genes]
    recs = []
    for k in range(n):
        g = rng.choice(genes)
        pos = rng.randint(1, glen - 30)
        ops = []
        if rng.random() < 0.15:
            ops.append((rng.randint(1, 12), "H"))
        if rng.random() < 0.3:
            ops.append((rng.randint(1, 15), "S"))
        body = []
        for _ in range(rng.randint(1, 6)):
            body.append((rng.randint(1, 60), rng.choice("MMMM=X")))
            r = rng.random()
            if r < 0.25:
                body.append((rng.randint(1, 14), "I"))
            elif r < 0.5:
                body.append((rng.randint(1, 9), "D"))
            elif r < 0.55:
                body.append((rng.randint(1, 20), "N"))
        while body and body[-1][1] in "IDN":
            body.pop()
        ops += body
        if rng.random() < 0.3:
            ops.append((rng.randint(1, 15), "S"))
        if rng.random() < 0.1:
            ops.append((rng.randint(1, 12), "H"))
        qlen = sum(l for l, o in ops if o in "MIS=X")
        seq = "".join(rng.choice("ACGT") for _ in range(qlen))
        if rng.random() < 0.04:
            seq = seq[:3] + rng.choice("Nn") + seq[4:]
        flag = rng.choice([0, 0, 0, 16, 65, 129, 81, 161, 4, 256, 512, 1024, 2048, 99, 147])
        name = "r%d" % rng.randint(0, n // 2)
        if rng.random() < 0.1:
            name += rng.choice(["/1", "/2"])
        mapq = rng.choice([0, 1, 3, 9, 10, 12, 30, 42, 60])
        recs.append((g, pos, "\t".join([name, str(flag), g, str(pos), str(mapq), "".join("%d%s" % x for x in ops), "*", "0", "0", seq,
                                        "I" * len(seq)])))
        if rng.random() < 0.2:      # an exact duplicate under another name
            f = recs[-1][2].split("\t")
            f[0] = "d%d" % k
            recs.append((g, pos, "\t".join(f)))
    recs.sort(key=lambda r: (r[0], r[1]))
    with open(path, "w") as f:
        f.write("\n".join(lines + [r[2] for r in recs]) + "\n")


@pytest.mark.parametrize("seed", list(range(8)))
def test_native_ingest_equals_python_mirror(seed, tmp_path):
    """sc_aln_* (C++, the product's reader) against the Python mirror of StrainCall.cpp:480-736 on records with every
    CIGAR shape, over many windows and option values: reads, copies, mates, depth thinning, pileup flags."""
    from rambl_amd import capi, samio
    rng = random.Random(1000 + seed)
    sam = str(tmp_path / "x.sam")
    _random_sam(rng, sam)
    nat = capi.NativeAln(sam)
    py = mirror.SamText(sam)

    class PyAln:
        native = None
        view = staticmethod(lambda mq, region: py.view(mq, 1804, region))
    n_checked = 0
    for _ in range(40):
        g = rng.choice(["gA", "gB"])
        p0 = rng.randint(1, 300)
        p1 = rng.randint(p0 + 5, 400)
        mq = rng.choice([0, 3, 10])
        rl = rng.choice([0, 20, 70])
        max_ins = rng.choice([3, 10, 13])
        max_depth = rng.choice([2, 10, 800])
        roi = "%s:%d-%d" % (g, p0, p1)
        assert nat.pileup_flags(mq, g, p0, p1) == py.pileup_flags(mq, roi)
        try:
            exp = mirror.load_mapping_reads("", PyAln, mq, rl, max_ins, max_depth, roi)
        except (ValueError, IndexError):
            with pytest.raises(capi.StrainCallError):
                nat.load_reads("", g, p0, p1, mq, rl, max_ins, max_depth)
            continue
        got = nat.load_reads("", g, p0, p1, mq, rl, max_ins, max_depth)
        assert (got.pos, got.cigar, got.seq, got.copies, got.mates) == (exp.pos, exp.cigar, exp.seq, exp.copies, exp.mates), roi
        assert got.n_input == len(py.view(mq, 1804, roi))
        n_checked += len(exp)
    assert n_checked > 200


def test_add_ones_equals_the_literal_loop(tmp_path):
    """The urn update `a[c] += 1` per draw (NonparametricClustering.cpp:195) in O(log k): tests/native/add_ones_check.cpp
    compares sc_add_ones with the literal long double loop around every binade border and at random."""
    exe = str(tmp_path / "add_ones_check")
    subprocess.check_call(["g++", "-O1", "-o", exe, os.path.join(ROOT, "tests", "native", "add_ones_check.cpp"),
                           "-L" + os.path.join(ROOT, "rambl_amd"), "-lstraincall_hip", "-Wl,-rpath," + os.path.join(ROOT, "rambl_amd")])
    p = subprocess.run([exe], stdout=subprocess.PIPE)
    assert p.returncode == 0, p.stdout.decode()
    assert b" 0 differences" in p.stdout


def _bgzf_block(payload):
    """One BGZF block exactly as SAM specification section 4.1 lays it out (gzip member, FEXTRA with the BC subfield)."""
    import struct
    import zlib
    comp = zlib.compressobj(9, zlib.DEFLATED, -15)
    cdata = comp.compress(payload) + comp.flush()
    bsize = len(cdata) + 25                                   # total block size - 1
    return (b"\x1f\x8b\x08\x04" + b"\x00\x00\x00\x00" + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<H", 2) +
            struct.pack("<H", bsize) + cdata + struct.pack("<I", zlib.crc32(payload) & 0xFFFFFFFF) + struct.pack("<I", len(payload)))


@pytest.mark.parametrize("threads", ["1", "5"])
def test_bam_of_many_bgzf_members_decodes_like_the_sam(threads, tmp_path, monkeypatch):
    """A BAM of a few hundred BGZF members (inflated side by side by SC_INGEST_THREADS host threads) gives the records of
    the SAM text it was written from, in the same order; a flipped bit in one member is reported, not decoded."""
    from rambl_amd import capi, synth
    monkeypatch.setenv("SC_INGEST_THREADS", threads)
    gene = synth.make_gene(7, glen=1500, n_strains=3, n_reads=60000, name="g7")
    fa, sam = synth.write_dataset(str(tmp_path), [gene])
    bam = str(tmp_path / "reads.bam")
    T.write_bam(sam, bam)
    assert os.path.getsize(bam) > 300000                 # ~230 members of 60 000 bytes before compression
    a, b = capi.NativeAln(sam), capi.NativeAln(bam)
    assert a.records() == b.records() == 60000
    seq = open(fa).read().split("\n", 1)[1].replace("\n", "")
    ra, rb = a.load_reads(seq, "g7", 1, 1500, 0, 70, 13, 800), b.load_reads(seq, "g7", 1, 1500, 0, 70, 13, 800)
    assert (ra.pos, ra.cigar, ra.seq, ra.copies, ra.mates) == (rb.pos, rb.cigar, rb.seq, rb.copies, rb.mates)
    raw = bytearray(open(bam, "rb").read())
    raw[len(raw) // 2] ^= 0x40
    bad = str(tmp_path / "bad.bam")
    open(bad, "wb").write(bytes(raw))
    with pytest.raises(capi.StrainCallError):
        capi.NativeAln(bad)


def test_bam_decoder_on_hand_assembled_bytes(tmp_path):
    """The library's BGZF/BAM decoder (sc_aln_open) on bytes assembled field by field from the SAM specification (section
    4.2: block_size, refID, pos, l_read_name, mapq, bin, n_cigar_op, flag, l_seq, next_refID, next_pos, tlen, read_name,
    cigar as oplen<<4|op, 4-bit packed seq over '=ACMGRSVTWYHKDBN', qual) -- not by any BAM writer of this repository:
    records that straddle BGZF block borders, an empty block in the middle, = and X and N and H and P operations, an odd
    l_seq, a 0xff quality string, an unmapped mate (flag 73) and a fully unmapped read (refID -1).  The expected SAM fields
    are written out by hand below; the Python reader (samio.bam_records, gzip module) must agree too."""
    import struct
    from rambl_amd import capi, samio

    def rec(ref_id, pos0, name, mapq, cigar_ops, flag, seq_nibbles, l_seq, qual, next_ref=-1, next_pos=-1, tlen=0):
        body = struct.pack("<iiBBHHHiiii", ref_id, pos0, len(name) + 1, mapq, 4680, len(cigar_ops), flag, l_seq, next_ref, next_pos, tlen)
        body += name + b"\x00" + b"".join(struct.pack("<I", c) for c in cigar_ops) + seq_nibbles + qual
        return struct.pack("<i", len(body)) + body

    hdr_text = b"@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:otuA\tLN:50\n@SQ\tSN:otuB\tLN:40\n"
    head = b"BAM\x01" + struct.pack("<i", len(hdr_text)) + hdr_text + struct.pack("<i", 2)
    head += struct.pack("<i", 5) + b"otuA\x00" + struct.pack("<i", 50) + struct.pack("<i", 5) + b"otuB\x00" + struct.pack("<i", 40)
    # r1: otuA pos 3, 2S 3= 1X 2I 1D 2M (7 aligned read bases + 2 clipped = 10 bases "TTACGTAACG"), flag 99, mate at otuA:30
    r1 = rec(0, 2, b"r1", 42, [2 << 4 | 4, 3 << 4 | 7, 1 << 4 | 8, 2 << 4 | 1, 1 << 4 | 2, 2 << 4 | 0], 99,
             bytes([0x88, 0x12, 0x48, 0x11, 0x24]), 10, bytes([40] * 10), 0, 29, 37)
    # r2: otuA pos 10, 5M 4N 3M with hard clips and a pad: 2H 5M 4N 1P 3M 3H; 8 bases "GGGGGCCC"; odd length next: see r3
    r2 = rec(0, 9, b"read/2", 0, [2 << 4 | 5, 5 << 4 | 0, 4 << 4 | 3, 1 << 4 | 6, 3 << 4 | 0, 3 << 4 | 5], 16,
             bytes([0x44, 0x44, 0x42, 0x22]), 8, bytes([0xff] * 8))
    # r3: otuB pos 1, 7M, 7 bases "ACGTNRY" (odd l_seq: the last low nibble is padding), mate unmapped (flag 73)
    r3 = rec(1, 0, b"r3", 255, [7 << 4 | 0], 73, bytes([0x12, 0x48, 0xF5, 0xA0]), 7, bytes(range(7)), 1, 0, 0)
    # r4: unmapped, no reference, no cigar, 4 bases "TTTT"
    r4 = rec(-1, -1, b"r4", 0, [], 77, bytes([0x88, 0x88]), 4, bytes([30] * 4))
    stream = head + r1 + r2 + r3 + r4
    # block borders inside the header, inside r1's cigar, inside r2's name; one empty block; EOF marker block
    cuts = [0, 17, len(head) + 44, len(head) + len(r1) + 40, len(stream)]
    blob = b""
    for a, b in zip(cuts, cuts[1:]):
        blob += _bgzf_block(stream[a:b])
        if a == 17:
            blob += _bgzf_block(b"")
    blob += bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    path = str(tmp_path / "hand.bam")
    open(path, "wb").write(blob)

    expected = [
        ["r1", "99", "otuA", "3", "42", "2S3=1X2I1D2M", "=", "30", "37", "TTACGTAACG", "I" * 10],
        ["read/2", "16", "otuA", "10", "0", "2H5M4N1P3M3H", "*", "0", "0", "GGGGGCCC", "*"],
        ["r3", "73", "otuB", "1", "255", "7M", "=", "1", "0", "ACGTNRY", "!\"#$%&'"],
        ["r4", "77", "*", "0", "0", "*", "*", "0", "0", "TTTT", "????"],
    ]
    assert list(mirror.bam_records(path)) == expected
    aln = capi.NativeAln(path)
    assert aln.records() == 4
    assert aln.ref_stats("otuA") == (2, (3 + 1 + 1 + 2) + (5 + 4 + 3)) and aln.ref_stats("otuB") == (1, 7) and aln.ref_stats("*") == (1, 1)
    # view semantics on them: -F 1804 drops r4 (unmapped) and keeps r1 (99), r2 (16), r3 (73: mate unmapped is bit 8 = filtered)
    got = aln.load_reads("", "otuA", 1, 50, 0, 0, 13, 800)
    assert got.n_input == 2
    assert (got.pos, got.cigar, got.seq, got.copies) == ([2, 9], ["3M1M2I1D2M", "2H5M4N1P3M3H"], ["ACGTAACG", "GGGGGCCC"], [1, 1])
    assert got.mates == [[-1], [-1]]                         # r1 -> "r1/1" (flag 99 has 65 set? 99 = 1+2+32+64: yes), mate "r1/2" absent
    assert aln.load_reads("", "otuB", 1, 40, 0, 0, 13, 800).n_input == 0      # flag 73 & 1804 = 8: mate unmapped is filtered by -F 1804
    # pileup: r1 covers 3-9 (insertion marked on the base before it, 6; deleted base 7), r2 covers 10-21; its reference
    # skip 15-18 prints '<' (reverse strand), which window_adjust (StrainCall.cpp:712-735) does not read as a deletion
    assert aln.pileup_flags(0, "otuA", 1, 50) == {p: (p == 6, p == 7) for p in range(3, 22)}


def test_reference_skip_is_coverage_not_deletion(tmp_path):
    """CIGAR N (reference skip): mpileup prints '>' / '<' there, not '*'; window_adjust looks for '+', '-' and '*' only
    (StrainCall.cpp:712-735), so the skipped span counts as covered and carries no deletion mark -- in the native
    reader, in the Python mirror and in the pileup text of the stand-in alike."""
    from rambl_amd import capi, samio
    sam = tmp_path / "n.sam"
    sam.write_text("@SQ\tSN:g\tLN:60\n"
                   "a\t0\tg\t5\t30\t4M3N4M1D2M\t*\t0\t0\tACGTACGTAC\tIIIIIIIIII\n"
                   "b\t16\tg\t20\t30\t3M2N3M\t*\t0\t0\tACGTAC\tIIIIII\n")
    want = {p: (False, p in (15, 16)) for p in list(range(5, 19)) + list(range(20, 28))}     # 15: "-1N" on the base before; 16: '*'
    nat = capi.NativeAln(str(sam))
    assert nat.pileup_flags(0, "g", 1, 60) == want
    text = mirror.SamText(str(sam))
    assert text.pileup_flags(0, "g:1-60") == want
    lines = text.mpileup(0, "g:1-60")
    assert mirror.flags_from_pileup_text(lines) == want
    assert ">" in lines[4].split("\t")[4] and "<" in [ln for ln in lines if ln.split("\t")[1] == "23"][0].split("\t")[4]
