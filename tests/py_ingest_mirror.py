"""Second implementation of the host ingest (rows a2-a4), in Python -- TEST INFRASTRUCTURE.

The product reads alignment files and ingests windows in C++ (rambl_amd/csrc/sc_ingest.cpp: sc_aln_*).  This module is the
independent restatement the tests compare it with: the records of a SAM text / BAM file as Python objects, `samtools view`
and `samtools mpileup` emulated as TEXT (what /root/reference/StrainCall/StrainCall.cpp:496,696 parses), the pileup summary
read from that text as window_adjust reads it (:702-736), and load_mapping_reads (:480-670) with crop_read_within_window
(:291-414), read_align_end_pos (:276-289), max_insert_size (:427-440), parse_cigar (PartialOrderGraph.cpp:13-59) and
libstdc++'s mt19937 + generate_canonical.  Until round 3 it lived in the product package (rambl_amd/ingest.py, samio.py).
"""
import re

import numpy as np

from rambl_amd.ingest import gene_roi_end_pos, gene_roi_name, gene_roi_start_pos, stoi      # noqa: F401
from rambl_amd.samio import parse_region

_CIGAR_OK = re.compile(r"(?:\d+[MIDNSHP=X])+\Z")
_CIGAR_OP = re.compile(r"(\d+)([MIDNSHP=X])")


def parse_cigar(cigar):
    """PartialOrderGraph.cpp:13-59: [(op, len)], '=' and 'X' become 'M'."""
    if _CIGAR_OK.match(cigar):                     # well-formed: same result as the character loop below
        return [("M" if op in "=X" else op, int(n)) for n, op in _CIGAR_OP.findall(cigar)]
    out = []
    num = ""
    for ch in cigar:
        if ch in "MIDNSHP":
            out.append((ch, stoi(num)))
            num = ""
        elif ch in "=X":
            out.append(("M", stoi(num)))
            num = ""
        else:
            num += ch
    return out


class MT19937:
    """std::mt19937 + generate_canonical<double,53> (libstdc++), vectorised twist."""

    def __init__(self, seed):
        x = np.zeros(624, dtype=np.uint64)
        x[0] = seed
        for i in range(1, 624):
            x[i] = (1812433253 * (int(x[i - 1]) ^ (int(x[i - 1]) >> 30)) + i) & 0xFFFFFFFF
        self.x = x.astype(np.uint32)
        self.buf = np.zeros(0, dtype=np.uint32)
        self.p = 0

    def _twist(self):
        x = self.x
        UP, LO, A = np.uint32(0x80000000), np.uint32(0x7FFFFFFF), np.uint32(0x9908B0DF)

        def f(hi, lo):
            y = (hi & UP) | (lo & LO)
            return (y >> np.uint32(1)) ^ np.where(y & np.uint32(1), A, np.uint32(0))

        x[0:227] = x[397:624] ^ f(x[0:227], x[1:228])
        x[227:454] = x[0:227] ^ f(x[227:454], x[228:455])
        x[454:623] = x[227:396] ^ f(x[454:623], x[455:624])
        x[623] = x[396] ^ f(x[623:624], x[0:1])[0]
        z = x.copy()
        z ^= z >> np.uint32(11)
        z ^= (z << np.uint32(7)) & np.uint32(0x9D2C5680)
        z ^= (z << np.uint32(15)) & np.uint32(0xEFC60000)
        z ^= z >> np.uint32(18)
        self.buf = z
        self.p = 0

    def next_u32(self):
        if self.p >= len(self.buf):
            self._twist()
        v = int(self.buf[self.p])
        self.p += 1
        return v

    def canonical(self):
        x0 = self.next_u32()
        x1 = self.next_u32()
        r = float(x0 + x1 * 4294967296) / 18446744073709551616.0   # int -> double rounds to nearest even
        if r >= 1.0:
            r = float(np.nextafter(1.0, 0.0))
        return r


def read_align_end_pos(p0, cigars):
    for op, ln in cigars:
        if op == "M" or op == "D":
            p0 += ln
    return p0 - 1


def crop_read_within_window(w0, w1, seq, qual, ops, r0, r1):
    """The read inside the window [w0, w1] (StrainCall.cpp:291-414) -> (bases, CIGAR text).

    Soft clips go; the front of a read that starts before the window and the back of one that ends after it are cut
    away along a reference coordinate map of the operations: M and D are clipped position by position, an insertion
    in front of the first / behind the last kept reference position leaves with its bases, operations the reference
    does not know to consume anything (N, H, P) stay as they are.  Same rules as `crop_to_window` in
    rambl_amd/csrc/sc_ingest.cpp (the product's reader); raises where the C++ of the reference would run past a vector
    or a string."""
    n = len(ops)
    lead = trail = 0
    first = 0
    if ops[0][0] == "S":
        lead, first = ops[0][1], 1
    if first >= n:
        raise IndexError("crop: nothing but a soft clip")
    kept = []
    k = first
    if r0 < w0 and r0 < w1:
        # walk the reference cursor up to the window start
        cur, last_op, used_last = r0, None, 0
        while cur < w0 and cur < w1:
            op, ln = ops[k]                      # IndexError: the read never reaches the window
            k += 1
            used = 0
            if op in "MD":
                used = max(0, min(ln, w0 - cur))
                cur += used
                if op == "M":
                    lead += used
            elif op == "I":
                lead += ln
            last_op, used_last = (op, ln), used
        if used_last < last_op[1]:
            kept.append([last_op[0], last_op[1] - used_last])        # the operation the window starts in
    else:
        if ops[k][1] > 0:
            kept.append(list(ops[k]))
        k += 1
    kept.extend(list(o) for o in ops[k:])
    # the same from the other end, on the operations still held
    last = n - 1
    if ops[last][0] == "S":
        trail = ops[last][1]
        last -= 1
        kept.pop()
    cur = r1
    while cur > w1 and cur > w0:
        if last < 0:
            raise IndexError("crop: the read never comes back into the window")
        op, ln = ops[last]
        last -= 1
        used = 0
        if op in "MD":
            used = max(0, min(ln, cur - w1))
            cur -= used
            if op == "M":
                trail += used
        elif op == "I":
            trail += ln
        if used == ln or op == "I":
            kept.pop()
        else:
            kept[-1][1] -= used
    if lead > len(seq) or trail > len(qual):
        raise ValueError("crop_read_within_window: substr out of range")
    cnt = len(seq) - lead - trail
    return (seq[lead:] if cnt < 0 else seq[lead:lead + cnt]), "".join("%d%s" % (ln, o) for o, ln in kept)


def max_insert_size(cigar):
    ins = 0
    for op, ln in parse_cigar(cigar):
        if ln > ins and op == "I":
            ins = ln
    return ins


class RegionReads:
    """What load_gene_seq + load_mapping_reads hand to the graph stage."""

    def __init__(self, gene_seq, pos, cigar, seq, copies, mates):
        self.gene_seq = gene_seq
        self.pos = pos          # list[int]
        self.cigar = cigar      # list[str]
        self.seq = seq          # list[str]
        self.copies = copies    # list[int]
        self.mates = mates      # list[list[int]]  ReadPairs[uid]

    def __len__(self):
        return len(self.pos)


def load_mapping_reads(gene_seq, aln, mq, rl, max_ins, max_depth, gene_roi):
    """StrainCall.cpp:480-670 on view text (`aln.view`): the second implementation the native reader is compared with."""
    lines = aln.view(mq, gene_roi)
    p0 = gene_roi_start_pos(gene_roi)
    p1 = gene_roi_end_pos(gene_roi)
    depth = 0
    for line in lines:
        f = line.split()
        f += [""] * (11 - len(f))
        ln = 0
        for op, n in parse_cigar(f[5]):
            if op == "M" or op == "D":
                ln += n
        r0 = stoi(f[3])
        r1 = r0 + ln - 1
        if p0 <= r0 and p1 > r1:
            depth += r1 - r0 + 1
        elif p0 <= r0 and p1 <= r1:
            depth += p1 - r0 + 1
        elif p0 > r0 and p1 <= r1:
            depth += p1 - p0 + 1
        elif p0 > r0 and p1 > r1:
            depth += r1 - p0 + 1
    depth = int(depth / (p1 - p0 + 1))                 # C++ int division truncates toward zero
    rho = min(1.0, max_depth / (depth + 0.0)) if depth != 0 else 1.0
    gen = MT19937(1234)

    dups = {}
    for line in lines:
        f = line.split()
        f += [""] * (11 - len(f))
        if len(f[9]) < rl:
            continue
        if "N" in f[9] or "n" in f[9]:
            continue
        rn = f[0]
        flag = stoi(f[1])
        if (flag & 65) == 65:
            rn += "/1"
        elif (flag & 129) == 129:
            rn += "/2"
        cigars = parse_cigar(f[5])
        read_p0 = stoi(f[3])
        read_p1 = read_align_end_pos(read_p0, cigars)
        relative_pos = read_p0 - p0
        if relative_pos < 0:
            relative_pos = 0
        seq, cigar = crop_read_within_window(p0, p1, f[9], f[10], cigars, read_p0, read_p1)
        maxins = max_insert_size(cigar)
        if len(seq) > rl and maxins < max_ins:
            if gen.canonical() > rho:
                continue
            dups.setdefault((relative_pos, cigar, seq), []).append(rn)

    keys = sorted(dups)
    pos, cig, sq, cn = [], [], [], []
    uids = {}
    for uid, k in enumerate(keys):
        pos.append(k[0]); cig.append(k[1]); sq.append(k[2]); cn.append(len(dups[k]))
        for name in dups[k]:
            uids[name] = uid                           # later assignment wins
    mates = [[] for _ in keys]
    for rn1 in sorted(uids):
        uid = uids[rn1]
        rn2 = None
        if rn1[-2:] == "/1":
            rn2 = rn1[:-2] + "/2"
        elif rn1[-2:] == "/2":
            rn2 = rn1[:-2] + "/1"
        mates[uid].append(uids.get(rn2, -1) if rn2 is not None else -1)
    return RegionReads(gene_seq, pos, cig, sq, cn, mates)


# ---------------------------------------------------------------------------------------------------------------------
# the alignment file as Python objects; view / mpileup text
_CIG = re.compile(r"(\d+)([MIDNSHP=X])")


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




class PyAlignments:
    """view / mpileup / pileup_flags of a mapping file through the Python mirror (SAM text, or BAM decoded by bam_records):
    what rambl_amd.samio.Alignments offers, without the library."""

    native = None

    def __init__(self, path):
        from rambl_amd.samio import is_bam
        self.path = path
        self.sam = SamText(path, bam=is_bam(path))

    def view(self, mq, region):
        return self.sam.view(mq, 1804, region)

    def mpileup(self, mq, region):
        return self.sam.mpileup(mq, region)

    def pileup_flags(self, mq, region):
        return self.sam.pileup_flags(mq, region)
