"""Scan windows and read ingest of the StrainCall path.

Host side, in Python 3, of /root/reference/StrainCall/StrainCall.cpp: make_scan_window
(:798-848) and window_adjust (:673-783) over the pileup summary the library computes
(sc_aln_pileup_flags), load_gene_seq (:157-185, samio.Fasta) and load_mapping_reads
(:480-670), which runs in the library (sc_aln_load_reads, rambl_amd/csrc/sc_ingest.cpp) and
leaves exactly what the reference hands to `new PartialOrderGraph(gene_seq, reads)` -- the
window sequence, the sorted unique alignments with copy numbers, the mate table -- in the
packed arrays sc_roi_submit takes.  (The Python restatement of the read path that used to live
here is test infrastructure now: tests/py_ingest_mirror.py.)
"""
import re

_LEAD_INT = re.compile(r"\s*([+-]?\d+)")


def stoi(s):
    """std::stoi: leading integer, junk after it ignored; no digits -> error."""
    if s.isdigit():                                # the usual field: plain decimal digits
        return int(s)
    m = _LEAD_INT.match(s)
    if not m:
        raise ValueError("stoi: no conversion for %r" % (s,))
    return int(m.group(1))


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


def load_mapping_reads(gene_seq, aln, mq, rl, max_ins, max_depth, gene_roi):
    """StrainCall.cpp:480-670, in the library (sc_aln_load_reads): view filter, depth -> keep probability, length / N /
    insertion filters, crop to the window, mt19937(1234) thinning, duplicate collapse, mate table.  `aln`: samio.Alignments
    (anything with a `load_reads` of its own -- the tests' Python mirror -- is asked directly)."""
    reader = aln.native if getattr(aln, "native", None) is not None else aln
    return reader.load_reads(gene_seq, gene_roi_name(gene_roi), gene_roi_start_pos(gene_roi), gene_roi_end_pos(gene_roi),
                             mq, rl, max_ins, max_depth)
