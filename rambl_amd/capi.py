"""ctypes binding of libstraincall_hip.so (include/straincall_hip.h).

There is no CPU fallback: importing this module fails loudly when the HIP
library has not been built, and `Context()` fails when no gfx950 device exists.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libstraincall_hip.so")

SC_OK = 0
ERRORS = {-1: "SC_ERR_NO_DEVICE", -2: "SC_ERR_HIP", -3: "SC_ERR_ARG", -4: "SC_ERR_UNSUPPORTED",
          -5: "SC_ERR_CAPACITY", -6: "SC_ERR_INTERNAL"}


class ScParams(C.Structure):
    _fields_ = [("error_rate", C.c_float), ("tau", C.c_float), ("diff_rate", C.c_float),
                ("sweeps_cap", C.c_int), ("draw_budget", C.c_int), ("max_candidates", C.c_int),
                ("graph_only", C.c_int), ("want_trace", C.c_int), ("want_timing", C.c_int), ("want_graph", C.c_int)]


class ScStats(C.Structure):
    _fields_ = [("graph_ms", C.c_double), ("cluster_ms", C.c_double), ("sampler_kernel_ms", C.c_double),
                ("sampler_launches", C.c_long), ("sampler_read_copies", C.c_long), ("level_launches", C.c_long), ("draws", C.c_long),
                ("slow_draws", C.c_long), ("exact_draws", C.c_long), ("sampler_strains", C.c_long), ("chain_passes", C.c_long), ("chain_cycles", C.c_long), ("chain_wall_ticks", C.c_long), ("level_kernel_ticks", C.c_long), ("sampler_level_ticks", C.c_long), ("xcd_levels", C.c_long * 8), ("msa_calls", C.c_long), ("n_nodes", C.c_int), ("n_levels", C.c_int),
                ("n_unique_reads", C.c_int), ("n_read_copies", C.c_long), ("setup_ms", C.c_double), ("queue_ms", C.c_double), ("place_ms", C.c_double), ("mailbox_ms", C.c_double), ("host_us", C.c_double * 3), ("wake_us", C.c_double * 2), ("kind_levels", C.c_long * 17)]

    def as_dict(self):
        return {k: (list(getattr(self, k)) if k in ("xcd_levels", "kind_levels", "host_us", "wake_us") else getattr(self, k)) for k, _ in self._fields_}


class StrainCallError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__("%s (%d)%s" % (ERRORS.get(code, "error"), code, (": " + msg) if msg else ""))
        self.code = code


def load_library():
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback for the StrainCall path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, ip, cp = C.c_void_p, C.POINTER(C.c_int), C.c_char_p
    lib.sc_ctx_create.argtypes = [C.c_int, C.c_int, C.POINTER(vp)]
    lib.sc_ctx_destroy.argtypes = [vp]
    lib.sc_ctx_destroy.restype = None
    lib.sc_last_error.argtypes = [vp]
    lib.sc_last_error.restype = cp
    lib.sc_roi_error.argtypes = [vp, C.c_int]
    lib.sc_roi_error.restype = cp
    lib.sc_host_plan.argtypes = [C.c_int, C.c_int, C.c_double, ip]
    lib.sc_host_plan.restype = C.c_int
    lib.sc_host_bind.argtypes = [C.c_int]
    lib.sc_host_bind.restype = C.c_int
    lib.sc_roi_submit.argtypes = [vp, cp, C.c_int, ip, cp, ip, cp, ip, ip, ip, ip, C.c_int, C.POINTER(ScParams), ip]
    lib.sc_roi_wait.argtypes = [vp, C.c_int]
    lib.sc_roi_result.argtypes = [vp, C.c_int, C.c_char_p, C.c_long, ip, C.POINTER(C.c_double), C.c_int, ip]
    lib.sc_roi_graph_dump.argtypes = [vp, C.c_int, C.c_char_p, C.c_long, C.POINTER(C.c_long)]
    lib.sc_roi_trace.argtypes = [vp, C.c_int, C.c_char_p, C.c_long, C.POINTER(C.c_long)]
    lib.sc_roi_stats.argtypes = [vp, C.c_int, C.POINTER(ScStats)]
    lib.sc_roi_release.argtypes = [vp, C.c_int]
    lib.sc_roi_edge_support.argtypes = [vp, C.c_int, ip, C.c_int, ip]
    lib.sc_msa_align.argtypes = [vp, cp, ip, C.c_int, C.c_char_p, C.c_long, ip]
    lib.sc_roi_thread_tables.argtypes = [vp, C.c_int, ip, ip, C.c_int, ip, C.c_long, C.c_char_p, ip, C.POINTER(C.c_long)]
    pi, pc, pu = C.POINTER(ip), C.POINTER(C.c_char_p), C.POINTER(C.c_ubyte)
    lib.sc_aln_open.argtypes = [cp, C.POINTER(vp)]
    lib.sc_aln_open_filtered.argtypes = [cp, C.POINTER(C.c_char_p), C.c_int, C.POINTER(vp)]
    lib.sc_aln_open_filtered.restype = C.c_int
    lib.sc_aln_close.argtypes = [vp]
    lib.sc_aln_close.restype = None
    lib.sc_aln_error.argtypes = [vp]
    lib.sc_aln_error.restype = cp
    lib.sc_aln_records.argtypes = [vp]
    lib.sc_aln_records.restype = C.c_long
    lib.sc_aln_ref_stats.argtypes = [vp, cp, C.POINTER(C.c_long), C.POINTER(C.c_long)]
    lib.sc_aln_ref_stats.restype = C.c_int
    lib.sc_aln_pileup_flags.argtypes = [vp, cp, C.c_int, C.c_int, C.c_int, pu, pu, pu]
    lib.sc_aln_load_reads.argtypes = [vp, cp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    lib.sc_reads_get.argtypes = [vp, ip, pi, pc, pi, pc, pi, pi, pi, pi, C.POINTER(C.c_long), ip]
    lib.sc_reads_free.argtypes = [vp]
    lib.sc_reads_free.restype = None
    for f in ("sc_aln_open", "sc_aln_pileup_flags", "sc_aln_load_reads", "sc_reads_get"):
        getattr(lib, f).restype = C.c_int
    for f in ("sc_ctx_create", "sc_roi_submit", "sc_roi_wait", "sc_roi_result", "sc_roi_graph_dump", "sc_roi_trace",
              "sc_roi_stats", "sc_roi_release", "sc_roi_edge_support", "sc_msa_align", "sc_roi_thread_tables"):
        getattr(lib, f).restype = C.c_int
    return lib


EXPORTS = ["sc_ctx_create", "sc_ctx_destroy", "sc_last_error", "sc_roi_error", "sc_host_plan", "sc_host_bind", "sc_roi_submit", "sc_roi_wait", "sc_roi_result",
           "sc_roi_graph_dump", "sc_roi_trace", "sc_roi_stats", "sc_roi_release", "sc_roi_edge_support", "sc_msa_align",
           "sc_roi_thread_tables", "sc_aln_open", "sc_aln_open_filtered", "sc_aln_close", "sc_aln_error", "sc_aln_records", "sc_aln_ref_stats", "sc_aln_pileup_flags",
           "sc_aln_load_reads", "sc_reads_get", "sc_reads_free", "sc_depth_scan", "sc_depth_scan_runs"]


def default_params(error_rate=0.01, tau=0.02, diff_rate=0.01, graph_only=False, want_trace=False, want_timing=False, want_graph=False):
    return ScParams(error_rate, tau, diff_rate, 5000, 40000, 80, int(graph_only), int(want_trace), int(want_timing), int(want_graph))


def _pack(strings):
    off = (C.c_int * (len(strings) + 1))()
    n = 0
    for i, s in enumerate(strings):
        off[i] = n
        n += len(s)
    off[len(strings)] = n
    return "".join(strings).encode("ascii"), off


def host_bind(device=0):
    """This process's main thread (and every thread started after the call) onto the CPUs next to GPU `device`; the number of
    CPUs, 0 when nothing was changed."""
    return int(lib().sc_host_bind(int(device)))


def host_plan(streams, local_world=0, cpus=0.0):
    """(executor threads, level-server threads, ingest threads) a Context(streams) starts on this host share."""
    out = (C.c_int * 3)()
    rc = lib().sc_host_plan(streams, local_world, cpus, out)
    if rc != SC_OK:
        raise StrainCallError(rc)
    return tuple(out)


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = load_library()
    return _LIB


class NativeReads:
    """A window's reads as load_mapping_reads leaves them (rows a2-a4), held by the library as the packed arrays
    sc_roi_submit takes.  The list views (pos, cigar, seq, copies, mates) are built on demand, for tests and tools."""

    def __init__(self, handle, gene_seq):
        self._h = handle
        self.gene_seq = gene_seq
        n, depth, n_in = C.c_int(), C.c_int(), C.c_long()
        ip = C.POINTER(C.c_int)
        self._pos, self._cig_off, self._seq_off, self._cn, self._mate_idx, self._mate_off = ip(), ip(), ip(), ip(), ip(), ip()
        self._cig, self._seq = C.c_char_p(), C.c_char_p()
        rc = lib().sc_reads_get(handle, C.byref(n), C.byref(self._pos), C.byref(self._cig), C.byref(self._cig_off), C.byref(self._seq),
                                C.byref(self._seq_off), C.byref(self._cn), C.byref(self._mate_idx), C.byref(self._mate_off),
                                C.byref(n_in), C.byref(depth))
        if rc != SC_OK:
            raise StrainCallError(rc)
        self.n = n.value
        self.n_input = n_in.value          # alignments the view returned for the window
        self.depth = depth.value

    def __len__(self):
        return self.n

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().sc_reads_free(self._h)
            except TypeError:          # interpreter shutdown: the module globals are gone, the process frees the memory
                pass
            self._h = None

    def _texts(self, text, off):
        raw = C.string_at(text, off[self.n]) if self.n else b""
        return [raw[off[i]:off[i + 1]].decode("ascii") for i in range(self.n)]

    def _view(self, name, build):
        """The list views are built once (an access inside a loop would otherwise rebuild the whole list every time)."""
        cache = self.__dict__.setdefault("_views", {})
        if name not in cache:
            cache[name] = build()
        return cache[name]

    @property
    def pos(self):
        return self._view("pos", lambda: [self._pos[i] for i in range(self.n)])

    @property
    def copies(self):
        return self._view("copies", lambda: [self._cn[i] for i in range(self.n)])

    @property
    def cigar(self):
        return self._view("cigar", lambda: self._texts(self._cig, self._cig_off))

    @property
    def seq(self):
        return self._view("seq", lambda: self._texts(self._seq, self._seq_off))

    @property
    def mates(self):
        return self._view("mates", lambda: [[self._mate_idx[k] for k in range(self._mate_off[i], self._mate_off[i + 1])] for i in range(self.n)])


class NativeAln:
    """An alignment file (SAM text or BAM) read and indexed by the library (sc_aln_*)."""

    def __init__(self, path, only=None):
        """only: keep the records of these references only ([]: of none -- statistics for pricing the regions; None: all)."""
        self.path = path
        self._h = C.c_void_p()
        if only is None:
            rc = lib().sc_aln_open(path.encode(), C.byref(self._h))
        else:
            names = (C.c_char_p * max(len(only), 1))(*[n.encode() for n in only])
            rc = lib().sc_aln_open_filtered(path.encode(), names, len(only), C.byref(self._h))
        if rc != SC_OK:
            msg = lib().sc_aln_error(self._h).decode() if self._h else "cannot open"
            self.close()
            raise StrainCallError(rc, "%s: %s" % (path, msg))

    def close(self):
        if getattr(self, "_h", None):
            try:
                lib().sc_aln_close(self._h)
            except TypeError:          # interpreter shutdown
                pass
            self._h = C.c_void_p()

    def __del__(self):
        self.close()

    def records(self):
        return lib().sc_aln_records(self._h)

    def ref_stats(self, gene):
        """(alignments of the reference, reference bases they cover)."""
        n, b = C.c_long(), C.c_long()
        lib().sc_aln_ref_stats(self._h, gene.encode(), C.byref(n), C.byref(b))
        return n.value, b.value

    def pileup_flags(self, mq, gene, P, Q):
        """{position: (has_insert, has_delete)} for the covered positions of gene:P-Q."""
        import numpy as np
        n = Q - P + 1
        if n <= 0:
            return {}
        cov, ins, dele = (np.zeros(n, dtype=np.uint8) for _ in range(3))
        pu = C.POINTER(C.c_ubyte)
        rc = lib().sc_aln_pileup_flags(self._h, gene.encode(), P, Q, mq, cov.ctypes.data_as(pu), ins.ctypes.data_as(pu),
                                       dele.ctypes.data_as(pu))
        if rc != SC_OK:
            raise StrainCallError(rc)
        idx = np.nonzero(cov)[0]
        return {int(k) + P: (bool(ins[k]), bool(dele[k])) for k in idx}

    def load_reads(self, gene_seq, gene, p0, p1, mq, rl, max_ins, max_depth):
        h = C.c_void_p()
        rc = lib().sc_aln_load_reads(self._h, gene.encode(), p0, p1, mq, rl, max_ins, max_depth, C.byref(h))
        if rc != SC_OK:
            raise StrainCallError(rc, lib().sc_aln_error(self._h).decode())
        return NativeReads(h, gene_seq)


class RegionResult:
    def __init__(self, seqs, abundance, graph, trace, stats):
        self.seqs = seqs
        self.abundance = abundance
        self.graph = graph
        self.trace = trace
        self.stats = stats


class Context:
    """One HIP device, `streams` regions in flight."""

    def __init__(self, device=0, streams=1):
        self.lib = load_library()
        self.h = C.c_void_p()
        rc = self.lib.sc_ctx_create(device, streams, C.byref(self.h))
        if rc != SC_OK:
            raise StrainCallError(rc, "no gfx950 HIP device %d (the StrainCall path has no CPU fallback)" % device)

    def close(self):
        if self.h:
            self.lib.sc_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _err(self, rc):
        msg = self.lib.sc_last_error(self.h)
        return StrainCallError(rc, msg.decode("utf-8", "replace") if msg else "")

    def submit(self, region, params):
        """region: ingest.RegionReads or NativeReads.  Returns a handle."""
        n = len(region)
        if isinstance(region, NativeReads):
            ref = region.gene_seq.encode("ascii")
            handle = C.c_int()
            rc = self.lib.sc_roi_submit(self.h, ref, len(ref), region._pos, region._cig, region._cig_off, region._seq, region._seq_off,
                                        region._cn, region._mate_idx, region._mate_off, n, C.byref(params), C.byref(handle))
            if rc != SC_OK:
                raise self._err(rc)
            return handle.value
        pos = (C.c_int * max(n, 1))(*region.pos)
        cn = (C.c_int * max(n, 1))(*region.copies)
        cig, cig_off = _pack(region.cigar)
        seq, seq_off = _pack(region.seq)
        mate_off = (C.c_int * (n + 1))()
        flat = []
        for i, m in enumerate(region.mates):
            mate_off[i] = len(flat)
            flat.extend(m)
        mate_off[n] = len(flat)
        mate_idx = (C.c_int * max(len(flat), 1))(*flat)
        ref = region.gene_seq.encode("ascii")
        handle = C.c_int()
        rc = self.lib.sc_roi_submit(self.h, ref, len(ref), pos, cig, cig_off, seq, seq_off, cn, mate_idx, mate_off, n,
                                    C.byref(params), C.byref(handle))
        if rc != SC_OK:
            raise self._err(rc)
        return handle.value

    def wait(self, handle, want_graph=False, want_trace=False, release=True):
        rc = self.lib.sc_roi_wait(self.h, handle)
        graph = trace = None
        stats = ScStats()
        self.lib.sc_roi_stats(self.h, handle, C.byref(stats))
        if want_graph or rc != SC_OK:
            ln = C.c_long()
            self.lib.sc_roi_graph_dump(self.h, handle, None, 0, C.byref(ln))
            buf = C.create_string_buffer(ln.value + 1)
            if self.lib.sc_roi_graph_dump(self.h, handle, buf, ln.value, C.byref(ln)) == SC_OK:
                graph = buf.raw[:ln.value].decode("ascii")
        if rc != SC_OK:
            msg = self.lib.sc_roi_error(self.h, handle)            # this region's own message
            err = StrainCallError(rc, msg.decode("utf-8", "replace") if msg else "")
            if release:
                self.lib.sc_roi_release(self.h, handle)
            err.graph = graph
            raise err
        nst = C.c_int()
        cap, maxs = 1 << 16, 256
        while True:
            buf = C.create_string_buffer(cap)
            off = (C.c_int * (maxs + 1))()
            ab = (C.c_double * maxs)()
            rc = self.lib.sc_roi_result(self.h, handle, buf, cap, off, ab, maxs, C.byref(nst))
            if rc == -5 and cap < (1 << 30):
                cap *= 8
                maxs *= 4
                continue
            break
        if rc != SC_OK:
            raise self._err(rc)
        seqs = [buf.raw[off[i]:off[i + 1]].decode("ascii") for i in range(nst.value)]
        abundance = [ab[i] for i in range(nst.value)]
        if want_trace:
            ln = C.c_long()
            self.lib.sc_roi_trace(self.h, handle, None, 0, C.byref(ln))
            tb = C.create_string_buffer(ln.value + 1)
            if self.lib.sc_roi_trace(self.h, handle, tb, ln.value, C.byref(ln)) == SC_OK:
                trace = tb.raw[:ln.value].decode("ascii")
        if release:
            self.lib.sc_roi_release(self.h, handle)
        return RegionResult(seqs, abundance, graph, trace, stats.as_dict())

    def run(self, region, params, want_graph=False, want_trace=False):
        return self.wait(self.submit(region, params), want_graph, want_trace)

    def edge_support(self, handle):
        n = C.c_int()
        self.lib.sc_roi_edge_support(self.h, handle, None, 0, C.byref(n))
        arr = (C.c_int * max(n.value, 1))()
        rc = self.lib.sc_roi_edge_support(self.h, handle, arr, n.value, C.byref(n))
        if rc != SC_OK:
            raise self._err(rc)
        return list(arr[:n.value])

    def thread_tables(self, handle):
        """Row a5: (count[glen*8], first_read[glen*8], pool[], symbols) of a finished region."""
        ncls, npool = C.c_int(), C.c_long()
        self.lib.sc_roi_thread_tables(self.h, handle, None, None, 0, None, 0, None, C.byref(ncls), C.byref(npool))
        cnt = (C.c_int * max(ncls.value, 1))()
        first = (C.c_int * max(ncls.value, 1))()
        pool = (C.c_int * max(npool.value, 1))()
        sym = C.create_string_buffer(8)
        rc = self.lib.sc_roi_thread_tables(self.h, handle, cnt, first, ncls.value, pool, npool.value, sym, C.byref(ncls),
                                           C.byref(npool))
        if rc != SC_OK:
            raise self._err(rc)
        return list(cnt[:ncls.value]), list(first[:ncls.value]), list(pool[:npool.value]), sym.raw

    def msa_align(self, seqs):
        """Row a7: rows of the progressive sum-of-pairs MSA of `seqs` (in the given order)."""
        txt, off = _pack(seqs)
        cap = (len(txt) + 2) * max(len(seqs), 1) + 16
        out = C.create_string_buffer(cap)
        ncol = C.c_int()
        rc = self.lib.sc_msa_align(self.h, txt, off, len(seqs), out, cap, C.byref(ncol))
        if rc != SC_OK:
            raise self._err(rc)
        w = ncol.value + 1
        return [out.raw[i * w:i * w + ncol.value].decode("ascii") for i in range(len(seqs))]
