"""Scan windows and read ingest of the StrainCall path.

Host-side mirror, in Python 3, of /root/reference/StrainCall/StrainCall.cpp:
make_scan_window (:798-848), window_adjust (:673-783), load_gene_seq (:157-185)
and load_mapping_reads (:480-670) with crop_read_within_window (:291-414),
read_align_end_pos (:276-289), max_insert_size (:427-440) and parse_cigar
(PartialOrderGraph.cpp:13-59).  The result is exactly what the reference hands to
`new PartialOrderGraph(gene_seq, reads)`: the window sequence, the sorted unique
alignments with copy numbers, and the mate table.
"""
import re

import numpy as np

_LEAD_INT = re.compile(r"\s*([+-]?\d+)")


def stoi(s):
    """std::stoi: leading integer, junk after it ignored; no digits -> error."""
    if s.isdigit():                                # the usual field: plain decimal digits
        return int(s)
    m = _LEAD_INT.match(s)
    if not m:
        raise ValueError("stoi: no conversion for %r" % (s,))
    return int(m.group(1))


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


def gene_roi_name(roi):
    x = roi.find(":")
    return roi[:x] if x >= 0 else roi


def gene_roi_start_pos(roi):
    x = roi.find(":")
    s = roi[x + 1:]
    k = s.find("-")
    return stoi(s if k < 0 else s[:k])


def gene_roi_end_pos(roi):
    x = roi.find("-")         # first '-' anywhere in the roi, StrainCall.cpp:210-220
    return stoi(roi[x + 1:])


def window_adjust(aln, mq, gn, p0, p1, z, L):
    """StrainCall.cpp:673-783 -> (d0, d1)."""
    if p0 - z < 1:
        z = p0 - 1
    P = p0 - z
    Q = p1 + z
    if Q > L:
        Q = L
    info = aln.pileup_flags(mq, "%s:%d-%d" % (gn, P, Q))
    if not info:
        raise RuntimeError("no pileup for %s:%d-%d: the reference dereferences an empty map here" % (gn, p0, p1))
    keys = sorted(info)
    index = {k: n for n, k in enumerate(keys)}
    if p0 not in info:
        P = keys[0]
    else:
        P = p0
        k = index[p0]
        while info[keys[k]][0] or info[keys[k]][1]:
            k -= 1
            if k < 0:
                break
            P -= 1
    if p1 not in info:
        Q = keys[-1]
    else:
        Q = p1
        k = index[p1]
        while info[keys[k]][0] or info[keys[k]][1]:
            k += 1
            if k >= len(keys):
                break
            Q += 1
    return p0 - P, Q - p1


def make_scan_window(params, fai, aln):
    """StrainCall.cpp:798-848 -> [(gn, p0, p1)]."""
    z = 50
    if params.roi == "":
        gn = ""
        for name, _ in fai:          # gene_name(): last record of the .fai
            if name:
                gn = name
        l = 1
        L = LL = _gene_length(fai, gn)
    else:
        gn = gene_roi_name(params.roi)
        l = gene_roi_start_pos(params.roi)
        L = gene_roi_end_pos(params.roi)
        LL = _gene_length(fai, gn)
    windows = []
    visited = set()
    p0 = p1 = l
    d0 = d1 = 0
    while p1 < L:
        p1 = p0 + params.window_size - 1
        if p1 > L:
            p1 = L
        d0, d1 = window_adjust(aln, params.mapping_qual, gn, p0, p1, z, LL)
        if p0 == l:
            params.d0 = d0
        if (p1 + d1) not in visited:
            windows.append((gn, p0 - d0, p1 + d1))
            visited.add(p1 + d1)
        p0 += params.window_size - params.overlap_size
    params.d1 = d1
    return windows


def _gene_length(fai, name):
    ln = 0
    for n, l in fai:
        if n == name:
            ln = stoi(l)
    return ln


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
    """StrainCall.cpp:480-670.  With the library's own reader (aln.native) the whole function runs there
    (sc_aln_load_reads) and the result stays in the packed arrays sc_roi_submit takes."""
    if getattr(aln, "native", None) is not None:
        return aln.native.load_reads(gene_seq, gene_roi_name(gene_roi), gene_roi_start_pos(gene_roi), gene_roi_end_pos(gene_roi),
                                     mq, rl, max_ins, max_depth)
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
