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
    gen = ingest.MT19937(1234)
    assert [gen.canonical() for _ in range(n)] == list(out)
    # std::mt19937 known answer: the 10000th output of the default-seeded engine is 4123659995
    g = ingest.MT19937(5489)
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
    assert ingest.parse_cigar("3S10M2I4D5=6X") == [("S", 3), ("M", 10), ("I", 2), ("D", 4), ("M", 5), ("M", 6)]
    cig = ingest.parse_cigar("10M")
    assert ingest.crop_read_within_window(5, 8, "ACGTACGTAC", "IIIIIIIIII", cig, 3, 12) == ("GTAC", "4M")
    assert ingest.crop_read_within_window(1, 100, "ACGTACGTAC", "IIIIIIIIII", cig, 3, 12) == ("ACGTACGTAC", "10M")
    assert ingest.max_insert_size("5M3I2M7I1M") == 7


@pytest.mark.parametrize("seed", [0, 1, 2, 5, 6, 13])
def test_pileup_flags_equal_pileup_text(seed, tmp_path):
    """The array-based pileup summary equals the summary of the emulated pileup text."""
    from rambl_amd import samio
    args = T.make_case(seed, str(tmp_path))
    aln = samio.Alignments(args[-1])
    name = "g%d" % seed
    for region in ("%s:1-5000" % name, "%s:40-180" % name, "%s:200-260" % name):
        for mq in (0, 50):
            assert aln.pileup_flags(mq, region) == samio.flags_from_pileup_text(aln.mpileup(mq, region))


def _write_bam(sam_path, bam_path):
    """Minimal BAM writer (SAM spec section 4) used only to exercise the native reader."""
    import struct
    import zlib
    hdr, recs, refs = [], [], []
    for line in open(sam_path):
        if line.startswith("@"):
            hdr.append(line)
            if line.startswith("@SQ"):
                f = dict(x.split(":", 1) for x in line.rstrip().split("\t")[1:])
                refs.append((f["SN"], int(f["LN"])))
        elif line.strip():
            recs.append(line.rstrip("\n").split("\t"))
    ref_id = {n: i for i, (n, _) in enumerate(refs)}
    text = "".join(hdr).encode()
    out = bytearray(b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(refs)))
    for n, l in refs:
        out += struct.pack("<i", len(n) + 1) + n.encode() + b"\x00" + struct.pack("<i", l)
    seq_code = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
    cig_code = {c: i for i, c in enumerate("MIDNSHP=X")}
    import re
    for f in recs:
        qn = f[0].encode() + b"\x00"
        cig = [(int(n), op) for n, op in re.findall(r"(\d+)([MIDNSHP=X])", f[5])]
        seq = f[9]
        sb = bytearray()
        for k in range(0, len(seq), 2):
            hi = seq_code[seq[k]]
            lo = seq_code[seq[k + 1]] if k + 1 < len(seq) else 0
            sb.append(hi << 4 | lo)
        qual = bytes(ord(c) - 33 for c in f[10]) if f[10] != "*" else b"\xff" * len(seq)
        body = struct.pack("<iiBBHHHiiii", ref_id.get(f[2], -1), int(f[3]) - 1, len(qn), int(f[4]), 0, len(cig), int(f[1]),
                           len(seq), -1, -1, 0)
        body += qn + b"".join(struct.pack("<I", n << 4 | cig_code[op]) for n, op in cig) + bytes(sb) + qual
        out += struct.pack("<i", len(body)) + body
    with open(bam_path, "wb") as g:
        for k in range(0, len(out), 60000):          # BGZF: gzip members with the BC extra field
            chunk = bytes(out[k:k + 60000])
            comp = zlib.compressobj(6, zlib.DEFLATED, -15)
            cdata = comp.compress(chunk) + comp.flush()
            bsize = len(cdata) + 25
            g.write(b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize)
                    + cdata + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
        g.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))


@pytest.mark.parametrize("seed", [2, 4])
def test_native_bam_reader_gives_the_same_regions(seed, tmp_path, monkeypatch):
    """BAM input read natively (no samtools in the image) yields the same ingest as the SAM text."""
    monkeypatch.setenv("SC_NATIVE_BAM", "1")
    d = str(tmp_path)
    args = T.make_case(seed, d)
    bam = os.path.join(d, "reads.bam")
    _write_bam(args[-1], bam)
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
    py = samio.SamText(sam)

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
            exp = ingest.load_mapping_reads("", PyAln, mq, rl, max_ins, max_depth, roi)
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
