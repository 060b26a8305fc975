"""Text that crosses the samtools boundary of the StrainCall path.

The reference shells out to samtools 0.1.19 four times per window
(/root/reference/StrainCall/StrainCall.cpp:167 faidx, :496 view, :696 mpileup,
:229/:256 faidx for the index).  Only a few whitespace-separated fields of that
text are read.  This module produces the same text either

* natively, when the alignment file is SAM *text* (so benchmarks and tests do not
  depend on an external tool), or
* by running the real `samtools` with the reference's exact command lines when the
  alignment file is BAM (drop-in behaviour under scripts/rambl.py).

FASTA access is always native (a FASTA record is plain text).
"""
import re
import shutil
import subprocess

_CIG = re.compile(r"(\d+)([MIDNSHP=X])")


class SamtoolsMissing(RuntimeError):
    pass


def is_bam(path):
    try:
        with open(path, "rb") as f:
            return f.read(2) == b"\x1f\x8b"
    except OSError:
        return False


def _run_samtools(args):
    exe = shutil.which("samtools")
    if exe is None:
        raise SamtoolsMissing("`samtools` is needed to read BAM input (%s) and is not on PATH; "
                              "pass SAM text instead" % " ".join(args))
    p = subprocess.run([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    return p.stdout.decode("ascii", "replace").splitlines()


def parse_region(reg):
    if ":" in reg:
        name, span = reg.rsplit(":", 1)
        a, b = span.split("-")
        return name, int(a), int(b)
    return reg, None, None


class Fasta:
    """Whole-file FASTA reader (seed-gene files are a few MB)."""

    def __init__(self, path):
        self.path = path
        self.seqs = {}
        self.order = []
        name = None
        chunks = None
        with open(path) as f:
            for line in f:
                line = line.rstrip("\r\n")
                if line.startswith(">"):
                    if name is not None:
                        self.seqs[name] = "".join(chunks)
                    name = line[1:].split()[0] if len(line) > 1 else ""
                    chunks = []
                    self.order.append(name)
                elif name is not None:
                    chunks.append(line)
        if name is not None:
            self.seqs[name] = "".join(chunks)

    def fetch(self, region):
        """Sequence text `samtools faidx fa region` would print (header removed)."""
        name, a, b = parse_region(region)
        if name not in self.seqs and region in self.seqs:
            name, a, b = region, None, None
        s = self.seqs.get(name, "")
        if a is not None:
            s = s[max(0, a - 1):b]
        return s


def read_fai(path):
    """[(name, length)] from <fasta>.fai: fields 1-2 (StrainCall.cpp:233-270)."""
    out = []
    with open(path) as f:
        for line in f:
            fld = line.split()
            if fld:
                out.append((fld[0], fld[1] if len(fld) > 1 else ""))
    return out


def _ref_span(pos, cigar):
    n = 0
    for ln, op in _CIG.findall(cigar):
        if op in "MDN=X":
            n += int(ln)
    return pos, pos + max(n, 1) - 1


def bam_records(path):
    """Native BAM reader: BGZF is a series of gzip
    members, the payload is the BAM record stream of the SAM specification.  Yields
    the 11 mandatory SAM fields of every alignment as text (optional tags are not
    needed by the path).  The whole file is inflated; there is no .bai random access."""
    import gzip
    import struct
    with gzip.open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"BAM\x01":
        raise ValueError("%s: not a BAM file" % path)
    (l_text,) = struct.unpack_from("<i", data, 4)
    o = 8 + l_text
    (n_ref,) = struct.unpack_from("<i", data, o)
    o += 4
    refs = []
    for _ in range(n_ref):
        (l_name,) = struct.unpack_from("<i", data, o)
        o += 4
        refs.append(data[o:o + l_name - 1].decode("ascii"))
        o += l_name + 4
    seq_code = "=ACMGRSVTWYHKDBN"
    cig_code = "MIDNSHP=X"
    n = len(data)
    while o + 4 <= n:
        (block_size,) = struct.unpack_from("<i", data, o)
        o += 4
        ref_id, pos, l_read_name, mapq, _bin, n_cigar, flag, l_seq, next_ref, next_pos, tlen = struct.unpack_from(
            "<iiBBHHHiiii", data, o)
        p = o + 32
        qname = data[p:p + l_read_name - 1].decode("ascii")
        p += l_read_name
        cig = struct.unpack_from("<%dI" % n_cigar, data, p) if n_cigar else ()
        p += 4 * n_cigar
        cigar = "".join("%d%s" % (c >> 4, cig_code[c & 15]) for c in cig) or "*"
        nb = (l_seq + 1) // 2
        sb = data[p:p + nb]
        p += nb
        seq = "".join(seq_code[b >> 4] + seq_code[b & 15] for b in sb)[:l_seq] or "*"
        q = data[p:p + l_seq]
        qual = "*" if (l_seq == 0 or q[:1] == b"\xff") else bytes(c + 33 for c in q).decode("ascii")
        rname = refs[ref_id] if 0 <= ref_id < n_ref else "*"
        rnext = "*" if next_ref < 0 else ("=" if next_ref == ref_id else refs[next_ref])
        yield [qname, str(flag), rname, str(pos + 1), str(mapq), cigar, rnext, str(next_pos + 1), str(tlen), seq, qual]
        o += block_size


class SamText:
    """Alignments held in memory, indexed by reference name: a SAM text file, or a
    BAM file read natively (`bam=True`)."""

    def __init__(self, path, bam=False):
        self.path = path
        self.by_ref = {}
        if bam:
            for fld in bam_records(path):
                self.by_ref.setdefault(fld[2], []).append(("\t".join(fld), fld))
            return
        with open(path) as f:
            for line in f:
                if line.startswith("@") or not line.strip():
                    continue
                line = line.rstrip("\r\n")
                fld = line.split("\t")
                if len(fld) < 11:
                    continue
                self.by_ref.setdefault(fld[2], []).append((line, fld))

    def view(self, mq, fmask, region):
        name, a0, b0 = parse_region(region)
        out = []
        for line, f in self.by_ref.get(name, ()):
            if int(f[1]) & fmask:
                continue
            if int(f[4]) < mq:
                continue
            if a0 is not None:
                s, e = _ref_span(int(f[3]), f[5])
                if e < a0 or s > b0:
                    continue
            out.append(line)
        return out

    def mpileup(self, mq, region):
        """Lines of `samtools mpileup -q mq -Q0 -A -r region`; only fields 2 and 5
        are consumed downstream (StrainCall.cpp:712-735)."""
        name, a0, b0 = parse_region(region)
        cols = {}
        for line, f in self.by_ref.get(name, ()):
            flag = int(f[1])
            if flag & 1796 or int(f[4]) < mq:
                continue
            pos, seq = int(f[3]), f[9]
            s, e = _ref_span(pos, f[5])
            if a0 is not None and (e < a0 or s > b0):
                continue
            rev = bool(flag & 16)
            ops = [(int(n), op) for n, op in _CIG.findall(f[5]) if op not in "HP"]
            mapq_ch = chr(33 + min(int(f[4]), 93))
            j, p, first = 0, pos, True
            for k, (n, op) in enumerate(ops):
                if op == "S":
                    j += n
                elif op in "M=X":
                    for t in range(n):
                        if a0 is None or a0 <= p <= b0:
                            b = seq[j].lower() if rev else seq[j].upper()
                            txt = ("^" + mapq_ch if first else "") + b
                            if t == n - 1 and k + 1 < len(ops):
                                n2, op2 = ops[k + 1]
                                if op2 == "I":
                                    ins = seq[j + 1:j + 1 + n2]
                                    txt += "+%d%s" % (n2, ins.lower() if rev else ins.upper())
                                elif op2 == "D":
                                    txt += "-%d%s" % (n2, ("n" if rev else "N") * n2)
                            if p == e:
                                txt += "$"
                            cols.setdefault(p, []).append(txt)
                        first = False
                        j += 1
                        p += 1
                elif op == "I":
                    j += n
                elif op in "DN":
                    ch = "*" if op == "D" else ("<" if rev else ">")      # mpileup: deleted base / reference skip
                    for t in range(n):
                        if a0 is None or a0 <= p <= b0:
                            cols.setdefault(p, []).append(ch + ("$" if p == e else ""))
                        p += 1
        return ["%s\t%d\tN\t%d\t%s\t%s" % (name, p, len(cols[p]), "".join(cols[p]), "I" * len(cols[p]))
                for p in sorted(cols)]


def flags_from_pileup_text(lines):
    """{pos: (has_insert, has_delete)} exactly as StrainCall.cpp:705-736 reads pileup text:
    '+' anywhere in field 5 -> insert, '-' or '*' anywhere -> delete (which also catches
    the '^'+mapq characters '+', '-', '*')."""
    info = {}
    for line in lines:
        f = line.split()
        if len(f) < 2:
            continue
        f5 = f[4] if len(f) > 4 else ""
        info[int(f[1])] = ("+" in f5, ("-" in f5) or ("*" in f5))
    return info


def _sam_pileup_flags(self, mq, region):
    """Same result as flags_from_pileup_text(self.mpileup(mq, region)) without building the text."""
    import numpy as np
    name, a0, b0 = parse_region(region)
    lo = a0 if a0 is not None else 1
    recs = self.by_ref.get(name, ())
    hi = b0
    if hi is None:
        hi = max([_ref_span(int(f[3]), f[5])[1] for _, f in recs] + [1])
    n = hi - lo + 3
    cover = np.zeros(n + 1, dtype=np.int64)
    ins = np.zeros(n + 1, dtype=bool)
    dele = np.zeros(n + 1, dtype=bool)

    def mark(arr, p):
        if lo <= p <= hi:
            arr[p - lo] = True

    for line, f in recs:
        flag = int(f[1])
        if flag & 1796 or int(f[4]) < mq:
            continue
        pos = int(f[3])
        s, e = _ref_span(pos, f[5])
        if e < lo or s > hi:
            continue
        ops = [(int(k), op) for k, op in _CIG.findall(f[5]) if op not in "HP"]
        mapq_ch = chr(33 + min(int(f[4]), 93))
        p, first = pos, True
        for k, (ln, op) in enumerate(ops):
            if op in "M=X":
                a, b = max(p, lo), min(p + ln - 1, hi)
                if a <= b:
                    cover[a - lo] += 1
                    cover[b - lo + 1] -= 1
                if first:
                    if mapq_ch == "+":
                        mark(ins, p)
                    elif mapq_ch in "-*":
                        mark(dele, p)
                    first = False
                if k + 1 < len(ops):
                    op2 = ops[k + 1][1]
                    if op2 == "I":
                        mark(ins, p + ln - 1)
                    elif op2 == "D":
                        mark(dele, p + ln - 1)
                p += ln
            elif op in "DN":
                a, b = max(p, lo), min(p + ln - 1, hi)
                if a <= b:
                    cover[a - lo] += 1
                    cover[b - lo + 1] -= 1
                    if op == "D":                         # a reference skip prints '>' / '<': no deletion mark
                        dele[a - lo:b - lo + 1] = True
                p += ln
    depth = np.cumsum(cover[:n])
    out = {}
    for k in np.nonzero(depth > 0)[0]:
        out[int(k) + lo] = (bool(ins[k]), bool(dele[k]))
    return out


SamText.pileup_flags = _sam_pileup_flags


class Alignments:
    """view/mpileup provider for a mapping file (SAM text or BAM).

    Default: the library reads and indexes the file itself (capi.NativeAln: sc_aln_*), and a window's reads never
    become Python objects.  BAM with a `samtools` on PATH (and no SC_NATIVE_BAM=1): the reference's exact command
    lines, parsed by the Python mirror in ingest.py.  SC_PY_INGEST=1 forces the Python mirror for SAM text / BAM
    too (kept as the second implementation the tests compare the native one with)."""

    def __init__(self, path, only=None):
        """only: reference names whose records the native reader keeps ([]: none, the per-reference statistics only;
        None: all) -- a rank of a multi-GPU run prices every region, then keeps its own shard."""
        self.path = path
        import os
        self.bam = is_bam(path) and shutil.which("samtools") is not None and not os.environ.get("SC_NATIVE_BAM")
        self.native = None
        self.sam = None
        if not self.bam:
            if os.environ.get("SC_PY_INGEST"):
                self.sam = SamText(path, bam=is_bam(path))
            else:
                from . import capi
                self.native = capi.NativeAln(path, only)

    def _text(self):
        """The Python mirror of the file (tests and tools ask for pileup / view TEXT; the product path does not)."""
        if self.sam is None:
            self.sam = SamText(self.path, bam=is_bam(self.path))
        return self.sam

    def view(self, mq, region):
        if self.bam:   # StrainCall.cpp:496
            return _run_samtools(["view", self.path, "-q", str(mq), "-F", "1804", region])
        return self._text().view(mq, 1804, region)

    def mpileup(self, mq, region):
        if self.bam:   # StrainCall.cpp:696
            return _run_samtools(["mpileup", "-q", str(mq), "-Q0", "-A", "-r", region, self.path])
        return self._text().mpileup(mq, region)

    def pileup_flags(self, mq, region):
        """What window_adjust extracts from the pileup (StrainCall.cpp:702-736)."""
        if self.bam:
            return flags_from_pileup_text(self.mpileup(mq, region))
        if self.native is not None:
            name, a0, b0 = parse_region(region)
            return self.native.pileup_flags(mq, name, a0, b0)
        return self.sam.pileup_flags(mq, region)
