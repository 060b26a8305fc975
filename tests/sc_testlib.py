"""Shared helpers of the test-suite: seeded scenarios, running the oracle
(oracle/straincall_oracle, test infrastructure) and the product (rambl_amd.cli,
through the C-ABI), comparing traces."""
import ctypes
import io
import os
import random
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from rambl_amd import synth  # noqa: E402

ORACLE = os.path.join(ROOT, "oracle", "straincall_oracle")
ORACLE_LIB = os.path.join(ROOT, "oracle", "liboracle.so")
TOOLS = os.path.join(ROOT, "oracle", "tools")
RAMBL_ARGS = ["-q", "0", "-D", "800", "-I", "13", "-l", "70", "-t", "0.02", "-d", "0.02", "-w", "5000"]


def scenario(seed):
    """Seed -> (generator kwargs, option overrides); same families as
    oracle/tools/difftest.py: SNPs, indels, MSA sites, pairs, thinning, partial
    ROI, several windows, noisy deep data."""
    rng = random.Random(seed * 104729 + 7)
    kind = seed % 8
    kw = dict(glen=rng.randint(260, 520), n_strains=rng.randint(1, 4), n_reads=rng.randint(40, 260),
              rlen=rng.choice([100, 120, 150]), err=rng.choice([0.0, 0.003, 0.01]),
              n_sub=rng.randint(2, 10), n_ins=rng.randint(0, 2), n_del=rng.randint(0, 2))
    opts = dict(q=0, D=800, I=13, l=70, t=0.02, d=0.02, w=5000, roi="full")
    if kind == 1:
        kw.update(shared_ins_site=True, n_strains=rng.randint(2, 4))
    elif kind == 2:
        kw.update(paired=True, n_reads=rng.randint(80, 300))
    elif kind == 3:
        opts.update(D=rng.choice([5, 10, 20]))
    elif kind == 4:
        opts.update(roi="part")
    elif kind == 5:
        kw.update(ins_len=(1, 6), n_ins=2, shared_ins_site=True, err=0.01)
        opts.update(I=rng.choice([5, 13]))
    elif kind == 6:
        opts.update(w=rng.choice([200, 250]), o=rng.choice([50, 100]), l=rng.choice([40, 70]))
    elif kind == 7:
        kw.update(n_reads=rng.randint(300, 600), err=0.02, n_strains=rng.randint(2, 5))
    return kw, opts


def make_case(seed, outdir):
    kw, opts = scenario(seed)
    gene = synth.make_gene(seed, name="g%d" % seed, **kw)
    fa, sam = synth.write_dataset(outdir, [gene])
    rng = random.Random(seed)
    glen = len(gene["ref"])
    args = []
    if opts["roi"] == "full":
        args += ["-r", "%s:1-%d" % (gene["name"], glen)]
    elif opts["roi"] == "part":
        a = rng.randint(1, glen // 3)
        b = rng.randint(2 * glen // 3, glen)
        args += ["-r", "%s:%d-%d" % (gene["name"], a, b)]
    args += ["-q", str(opts["q"]), "-D", str(opts["D"]), "-I", str(opts["I"]), "-l", str(opts["l"]),
             "-t", str(opts["t"]), "-d", str(opts["d"]), "-w", str(opts["w"])]
    if "o" in opts:
        args += ["-o", str(opts["o"])]
    return args + [fa, sam]


def big_case(seed, outdir):
    """Larger regions than `scenario` makes: more strains, noisier reads -> many candidate strains
    (tools/parity_sweep.py --big; tests/golden/tie_case525)."""
    rng = random.Random(seed * 7919 + 3)
    kw = dict(glen=rng.randint(500, 900), n_strains=rng.randint(3, 7), n_reads=rng.randint(1200, 3500),
              rlen=rng.choice([100, 150]), err=rng.choice([0.005, 0.01, 0.02, 0.03]),
              n_sub=rng.randint(6, 20), n_ins=rng.randint(0, 3), n_del=rng.randint(0, 3),
              paired=rng.random() < 0.3, shared_ins_site=rng.random() < 0.3)
    gene = synth.make_gene(seed, name="b%d" % seed, **kw)
    fa, sam = synth.write_dataset(outdir, [gene])
    D = rng.choice([200, 400, 800])
    t = rng.choice([0.02, 0.01, 0.005])
    return ["-r", "%s:1-%d" % (gene["name"], len(gene["ref"])), "-q", "0", "-D", str(D), "-I", "13", "-l", "70",
            "-t", str(t), "-d", "0.02", "-w", "5000", fa, sam], kw


def param_case(seed, outdir):
    """Small regions with the command-line parameters varied (-e -t -d -l -I -D -w -o)."""
    rng = random.Random(seed * 31337 + 11)
    kw = dict(glen=rng.randint(240, 480), n_strains=rng.randint(1, 4), n_reads=rng.randint(60, 320),
              rlen=rng.choice([100, 120, 150]), err=rng.choice([0.0, 0.005, 0.02]),
              n_sub=rng.randint(2, 12), n_ins=rng.randint(0, 2), n_del=rng.randint(0, 2),
              paired=rng.random() < 0.25, shared_ins_site=rng.random() < 0.25)
    gene = synth.make_gene(seed, name="p%d" % seed, **kw)
    fa, sam = synth.write_dataset(outdir, [gene])
    glen = len(gene["ref"])
    opts = ["-r", "%s:1-%d" % (gene["name"], glen), "-q", "0",
            "-D", str(rng.choice([10, 50, 800])), "-I", str(rng.choice([3, 13])), "-l", str(rng.choice([40, 70, 100])),
            "-t", rng.choice(["0.005", "0.02", "0.1"]), "-d", rng.choice(["0.005", "0.02", "0.05"]),
            "-e", rng.choice(["0.001", "0.01", "0.05"])]
    if rng.random() < 0.3:
        opts += ["-w", str(rng.choice([150, 200, 300])), "-o", str(rng.choice([30, 50, 100]))]
    else:
        opts += ["-w", "5000"]
    return opts + [fa, sam], kw


def run_oracle(args, cwd, trace=False, graph=False, dump_reads=None, timeout=3000, check=True):
    env = dict(os.environ)
    env["PATH"] = TOOLS + os.pathsep + env.get("PATH", "")
    env["TMPDIR"] = cwd
    if trace:
        env["SC_TRACE"] = "1"
        env["SC_TRACE_PREC"] = "17"
    if dump_reads:
        env["SC_ORACLE_DUMP_READS"] = dump_reads
    a = [ORACLE] + (["-G"] if graph else []) + list(args)
    p = subprocess.run(a, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    if not check and p.returncode < 0:
        return None, None                                  # killed by a signal: the reference crashes on this input too
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return p.stdout.decode(), p.stderr.decode()


def run_product(args, trace_file=None, graph=False):
    """The product path, in-process: rambl_amd.cli.main -> ctypes -> HIP."""
    from rambl_amd import cli
    out, err = io.StringIO(), io.StringIO()
    old = os.environ.get("SC_TRACE_FILE")
    if trace_file:
        os.environ["SC_TRACE_FILE"] = trace_file
    else:
        os.environ.pop("SC_TRACE_FILE", None)
    try:
        rc = cli.main((["-G"] if graph else []) + list(args), out=out, err=err)
    finally:
        if old is None:
            os.environ.pop("SC_TRACE_FILE", None)
        else:
            os.environ["SC_TRACE_FILE"] = old
    assert rc == 0
    return out.getvalue()


def parse_trace(text):
    """-> list of (when, level, [(strain_seq, abundance)])."""
    blocks = []
    lines = text.splitlines()
    i = 0
    while i < len(lines):
        if lines[i].startswith("------"):
            when = lines[i + 1]
            level = int(lines[i + 2].split(":")[1])
            i += 3
            rows = []
            while i < len(lines) and not lines[i].startswith("------"):
                seq, _, ab = lines[i].rpartition("\t")
                rows.append((seq, float(ab)))
                i += 1
            blocks.append((when, level, rows))
        else:
            i += 1
    return blocks


def compare_traces(got, exp, rel=1e-9):
    g, e = parse_trace(got), parse_trace(exp)
    assert len(g) == len(e), "trace block count %d vs %d" % (len(g), len(e))
    for bi, (bg, be) in enumerate(zip(g, e)):
        assert bg[0] == be[0] and bg[1] == be[1], (bi, bg[:2], be[:2])
        assert len(bg[2]) == len(be[2]), "block %d (%s level %d): %d vs %d strains" % (bi, bg[0], bg[1], len(bg[2]), len(be[2]))
        for (sg, ag), (se, ae) in zip(bg[2], be[2]):
            assert sg == se, "block %d level %d: strain sequence differs" % (bi, bg[1])
            if ag != ag and ae != ae:
                continue                                   # NaN in the reference too (e.g. 0/0 responsibilities)
            assert abs(ag - ae) <= rel * max(abs(ae), 1e-300), "block %d level %d: abundance %r vs %r" % (bi, bg[1], ag, ae)


def oracle_lib():
    lib = ctypes.CDLL(ORACLE_LIB)
    lib.oracle_msa_align.argtypes = [ctypes.POINTER(ctypes.c_char_p), ctypes.c_int, ctypes.c_char_p, ctypes.c_int,
                                     ctypes.POINTER(ctypes.c_int)]
    lib.oracle_sort_desc_perm.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    lib.oracle_mt_canonical.argtypes = [ctypes.c_uint, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    return lib


def oracle_msa(seqs):
    lib = oracle_lib()
    arr = (ctypes.c_char_p * len(seqs))(*[s.encode() for s in seqs])
    cap = (sum(len(s) for s in seqs) + 2) * len(seqs) + 16
    out = ctypes.create_string_buffer(cap)
    ncol = ctypes.c_int()
    rc = lib.oracle_msa_align(arr, len(seqs), out, cap, ctypes.byref(ncol))
    assert rc == 0
    w = ncol.value + 1
    return [out.raw[i * w:i * w + ncol.value].decode() for i in range(len(seqs))]


def write_bam(sam_path, bam_path):
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
