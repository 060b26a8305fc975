"""Stage 1 of rambl.py (depth and breadth of the marker genes across samples) on MI355X.

Mirror of /root/reference/scripts/coverage_all_samples.py (`coverage_all_samples.py <fofn> <gi> -c cores`, launched
by rambl.py:82-102): the per-base depth of every sample's BAM, summed over the samples, merged into intervals
(gaps of at most 10 uncovered bases) with their mean depth -- printed as `chrom <TAB> start <TAB> end <TAB> mean`, the
stdout of the reference's final `bedtools merge -c 4 -o mean -d 10`.  The alignment files are read by the library
(sc_aln_open), the depths and the interval reduction run on the device in one kernel (sc_depth_scan, k_depth_fused in
rambl_amd/csrc/sc_depth.hip: a wavefront per reference, its difference array in LDS).

Parity at the samtools / bedtools boundary is unpinned (neither tool is in the image, the reference holds no fixture):
samtools' per-file depth cap is not applied, and the mean is printed with bedtools' default precision as documented
(5 significant digits), `start` / `end` as the reference forms them (pos and pos + 1 of samtools' 1-based positions).
"""
import ctypes as C
import re

from . import capi, samio

_LEAD = re.compile(r"\s*([+-]?\d+)")


def _numeric_key(name):
    """`sort -k1,1n`: the leading number of the key, 0 when there is none."""
    m = _LEAD.match(name)
    return int(m.group(1)) if m else 0


class DepthStats(C.Structure):
    _fields_ = [("cells", C.c_long), ("runs", C.c_long), ("extract_ms", C.c_double), ("prepare_ms", C.c_double),
                ("upload_ms", C.c_double), ("kernel_ms", C.c_double)]


def depth_intervals(paths, fai_path, max_gap=10, device=0, alns=None):
    """-> ([(name, start, end, depth sum, covered positions)], stats dict), in the reference's output order
    (references by `sort -k1,1n`, then by name; intervals by start)."""
    lib = capi.lib()
    refs = sorted(((n, int(l)) for n, l in samio.read_fai(fai_path)), key=lambda r: (_numeric_key(r[0]), r[0]))
    alns = alns if alns is not None else [capi.NativeAln(p) for p in paths]
    n_refs = len(refs)
    names = (C.c_char_p * max(n_refs, 1))(*[n.encode() for n, _ in refs])
    lens = (C.c_int * max(n_refs, 1))(*[l for _, l in refs])
    handles = (C.c_void_p * max(len(alns), 1))(*[a._h for a in alns])
    lib.sc_depth_scan.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_int, C.c_int,
                                  C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_long), C.POINTER(C.c_int),
                                  C.c_int, C.POINTER(C.c_int), C.POINTER(DepthStats)]
    lib.sc_depth_scan.restype = C.c_int
    cap = max(4 * n_refs, 1024)
    while True:
        iv = [(C.c_int * cap)() for _ in range(3)]
        sm = (C.c_long * cap)()
        cn = (C.c_int * cap)()
        n = C.c_int()
        st = DepthStats()
        rc = lib.sc_depth_scan(device, handles, len(alns), names, lens, n_refs, max_gap, iv[0], iv[1], iv[2], sm, cn, cap, C.byref(n),
                               C.byref(st))
        if rc == -5 and n.value > cap:
            cap = n.value
            continue
        if rc != capi.SC_OK:
            raise capi.StrainCallError(rc, "sc_depth_scan")
        break
    out = [(refs[iv[0][i]][0], iv[1][i], iv[2][i], sm[i], cn[i]) for i in range(n.value)]
    return out, {k: getattr(st, k) for k, _ in DepthStats._fields_}


def bed_text(intervals):
    """The reference's stdout: chrom, start (= first position), end (= last position + 1), mean depth."""
    return "".join("%s\t%d\t%d\t%.5g\n" % (name, s, e + 1, sm / n) for name, s, e, sm, n in intervals)


def main(argv=None):
    """`python -m rambl_amd.stage1 <fofn> <gi>`: list of BAM files (one per line), gene index (.fai) -> BED on stdout."""
    import argparse
    import sys
    ap = argparse.ArgumentParser(description="rambl.py stage 1 (depth and breadth of marker genes across samples) on MI355X")
    ap.add_argument("fofn", help="list of bam filenames, one filename by one line")
    ap.add_argument("gi", help="gene/genome index file (.fai)")
    ap.add_argument("-c", "--cores", type=int, default=1, help="accepted for compatibility; the depth runs on the GPU")
    ap.add_argument("-v", action="store_true", dest="verbose")
    a = ap.parse_args(argv)
    paths = [x.rstrip() for x in open(a.fofn) if x.strip()]
    iv, st = depth_intervals(paths, a.gi)
    sys.stdout.write(bed_text(iv))
    if a.verbose:
        sys.stderr.write("stage1: %d intervals, %s\n" % (len(iv), st))
    return 0


if __name__ == "__main__":
    import sys
    sys.exit(main())
