#!/usr/bin/env python3
"""Stage-1 depth scan on a synthetic 10^8-base input (66 667 genes of ~1 500 bases, reads of 150 bases at depth ~20):
times of k_depth_mark and k_depth_segments (HIP events inside sc_depth_scan_runs) and the segment kernel's streaming
rate (4 algorithmic bytes per cell of the difference array) against the 8 TB/s HBM peak.
usage: python3 tools/depth_bench.py [bases] [depth] [repeats]"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    bases = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
    depth = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    from rambl_amd import capi, stage1
    lib = capi.lib()
    rng = np.random.default_rng(3)
    n_refs = bases // 1500
    ref_len = rng.integers(1400, 1601, n_refs).astype(np.int32)
    n_runs = int(ref_len.sum() * depth / 150)
    run_ref = np.sort(rng.integers(0, n_refs, n_runs)).astype(np.int32)          # coordinate-sorted, as a BAM is
    start = np.maximum((rng.random(n_runs) * (ref_len[run_ref] - 150)).astype(np.int64) + 1, 1).astype(np.int32)
    end = np.minimum(start + 149, ref_len[run_ref]).astype(np.int32)
    ip = C.POINTER(C.c_int)
    cap = 4 * n_refs
    iv = [np.zeros(cap, dtype=np.int32) for _ in range(3)]
    sm = np.zeros(cap, dtype=np.int64)
    cn = np.zeros(cap, dtype=np.int32)
    lib.sc_depth_scan_runs.argtypes = [C.c_int, ip, C.c_int, ip, ip, ip, C.c_long, C.c_int, ip, ip, ip, C.POINTER(C.c_long), ip, C.c_int, ip,
                                       C.POINTER(stage1.DepthStats)]
    best = None
    for _ in range(reps):
        n = C.c_int()
        st = stage1.DepthStats()
        rc = lib.sc_depth_scan_runs(0, ref_len.ctypes.data_as(ip), n_refs, run_ref.ctypes.data_as(ip), start.ctypes.data_as(ip),
                                    end.ctypes.data_as(ip), n_runs, 10, iv[0].ctypes.data_as(ip), iv[1].ctypes.data_as(ip),
                                    iv[2].ctypes.data_as(ip), sm.ctypes.data_as(C.POINTER(C.c_long)), cn.ctypes.data_as(ip), cap, C.byref(n),
                                    C.byref(st))
        assert rc == 0, rc
        if best is None or st.segments_ms < best["segments_ms"]:
            best = dict(cells=st.cells, runs=st.runs, mark_ms=st.mark_ms, segments_ms=st.segments_ms, intervals=n.value)
    assert int(sm[:best["intervals"]].sum()) == int((end - start + 1).sum())
    gbs = 4.0 * best["cells"] / (best["segments_ms"] * 1e-3) / 1e9
    best.update(kernel="k_depth_segments<4>", algorithmic_bytes=4 * best["cells"], achieved_GBps=gbs, peak_GBps=8000.0, frac=gbs / 8000.0,
                mark_GBps_algorithmic=(8.0 * best["runs"] + 4.0 * best["cells"]) / (best["mark_ms"] * 1e-3) / 1e9)
    print(json.dumps(best))


if __name__ == "__main__":
    main()
