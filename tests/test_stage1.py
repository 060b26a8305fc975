"""rambl.py stage 1 (depth and breadth of marker genes across samples): the oracle restatement on a hand-computed case
(CPU), the device path (sc_depth_scan: k_depth_fused) against the oracle on random records (-m gpu)."""
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import depth_oracle  # noqa: E402  (test infrastructure)


def test_oracle_on_a_hand_computed_case():
    # two samples on a 40-base gene.  Sample 1: 5M at 3 (3-7), 3M2D2M at 20 (20-22, 25-26; 23-24 deleted: no depth).
    # Sample 2: 4M at 6 (6-9), a duplicate-flagged read at 1 (ignored), 2S3M at 38 (38-40).
    f1 = {"g": [(0, 3, "5M"), (16, 20, "3M2D2M")]}
    f2 = {"g": [(0, 6, "4M"), (1024, 1, "30M"), (0, 38, "2S3M")]}
    d = depth_oracle.depth_of_records(f1["g"], 40)
    assert d[3:8] == [1] * 5 and d[20:27] == [1, 1, 1, 0, 0, 1, 1] and sum(d) == 10
    iv = depth_oracle.stage1([f1, f2], [("g", 40)], max_gap=10)
    # covered: 3-9 (depths 1 1 1 2 2 1 1), 20-22, 25-26, 38-40; gaps 10-19 (10 bases: merges), 23-24, 27-37 (11 bases: splits)
    assert iv == [(0, 3, 26, 9 + 5, 7 + 5), (0, 38, 40, 3, 3)]
    assert depth_oracle.stage1([f1, f2], [("g", 40)], max_gap=9) == [(0, 3, 9, 9, 7), (0, 20, 26, 5, 5), (0, 38, 40, 3, 3)]
    assert depth_oracle.stage1([f1, f2], [("g", 40)], max_gap=0) == [(0, 3, 9, 9, 7), (0, 20, 22, 3, 3), (0, 25, 26, 2, 2), (0, 38, 40, 3, 3)]


def test_bed_text_and_reference_order():
    from rambl_amd import stage1
    assert stage1.bed_text([("4479944", 3, 26, 14, 12), ("g", 38, 40, 3, 3)]) == "4479944\t3\t27\t1.1667\ng\t38\t41\t1\n"
    assert sorted(["10", "9", "abc", "2x"], key=lambda n: (stage1._numeric_key(n), n)) == ["abc", "2x", "9", "10"]


def _random_files(rng, n_files, refs, n_reads):
    files, texts = [], []
    for _ in range(n_files):
        recs = {}
        lines = ["@SQ\tSN:%s\tLN:%d" % r for r in refs]
        rows = []
        for k in range(n_reads):
            name, ln = rng.choice(refs)
            pos = rng.randint(1, max(1, ln - 5))
            ops = []
            if rng.random() < 0.2:
                ops.append((rng.randint(1, 9), "S"))
            for _ in range(rng.randint(1, 4)):
                ops.append((rng.randint(1, 70), rng.choice("MMM=X")))
                r = rng.random()
                if r < 0.2:
                    ops.append((rng.randint(1, 6), "I"))
                elif r < 0.4:
                    ops.append((rng.randint(1, 14), rng.choice("DN")))
            while ops and ops[-1][1] in "IDN":
                ops.pop()
            cigar = "".join("%d%s" % o for o in ops)
            flag = rng.choice([0, 0, 0, 16, 99, 147, 4, 256, 512, 1024, 2048])
            qlen = sum(l for l, o in ops if o in "MIS=X")
            rows.append((name, pos, "\t".join(["r%d" % k, str(flag), name, str(pos), "30", cigar, "*", "0", "0", "A" * qlen, "I" * qlen])))
            recs.setdefault(name, []).append((flag, pos, cigar))
        rows.sort(key=lambda r: (r[0], r[1]))
        files.append(recs)
        texts.append("\n".join(lines + [r[2] for r in rows]) + "\n")
    return files, texts


@pytest.mark.gpu
@pytest.mark.parametrize("seed,max_gap", [(1, 10), (2, 10), (3, 0), (4, 2), (5, 3), (6, 25)])
def test_depth_scan_matches_oracle(seed, max_gap, tmp_path):
    """sc_depth_scan (three samples, references of 1 to 3 000 bases, reads hanging over reference ends, deletions, skips,
    clips, filtered flags) against the oracle: every interval with its depth sum and covered positions, exactly."""
    from rambl_amd import capi, stage1
    rng = random.Random(seed)
    # (2 047 / 2 048 / 2 049: either side of the LDS tile of a wavefront; 3 000 and 9 000: two and five tiles)
    refs = [("%d" % (1000 + k), rng.choice([1, 2, 5, 63, 64, 65, 255, 256, 257, 700, 1500, 2047, 2048, 2049, 3000, 9000])) for k in range(40)]
    files, texts = _random_files(rng, 3, refs, 900)
    paths = []
    for i, t in enumerate(texts):
        p = str(tmp_path / ("s%d.sam" % i))
        open(p, "w").write(t)
        paths.append(p)
    fai = str(tmp_path / "genes.fai")
    open(fai, "w").write("".join("%s\t%d\t0\t60\t61\n" % r for r in refs))
    got, st = stage1.depth_intervals(paths, fai, max_gap=max_gap)
    order = sorted(refs, key=lambda r: (stage1._numeric_key(r[0]), r[0]))
    exp = depth_oracle.stage1(files, order, max_gap=max_gap)
    assert got == [(order[ri][0], s, e, sm, n) for ri, s, e, sm, n in exp]
    assert st["runs"] > 1000 and st["cells"] == sum(l for _, l in refs) and st["kernel_ms"] > 0
    text = stage1.bed_text(got)
    assert text.count("\n") == len(exp)


@pytest.mark.gpu
def test_depth_scan_large_property(tmp_path):
    """10^7 reference bases, 10^6 runs through sc_depth_scan_runs: the intervals partition exactly the covered bases and
    their depth sums add up to the total run length (size-independent properties; the oracle is not run at this size)."""
    import ctypes as C
    import numpy as np
    from rambl_amd import capi, stage1
    lib = capi.lib()
    rng = np.random.default_rng(9)
    n_refs, n_runs = 6667, 1_000_000
    ref_len = rng.integers(1200, 1800, n_refs).astype(np.int32)
    run_ref = rng.integers(0, n_refs, n_runs).astype(np.int32)
    ln = rng.integers(30, 151, n_runs)
    start = (rng.random(n_runs) * (ref_len[run_ref] - ln)).astype(np.int64) + 1
    start = np.maximum(start, 1).astype(np.int32)
    end = np.minimum(start + ln - 1, ref_len[run_ref]).astype(np.int32)
    cap = 4 * n_refs
    ip = C.POINTER(C.c_int)
    iv = [np.zeros(cap, dtype=np.int32) for _ in range(3)]
    sm = np.zeros(cap, dtype=np.int64)
    cn = np.zeros(cap, dtype=np.int32)
    n = C.c_int()
    st = stage1.DepthStats()
    lib.sc_depth_scan_runs.argtypes = [C.c_int, ip, C.c_int, ip, ip, ip, C.c_long, C.c_int, ip, ip, ip, C.POINTER(C.c_long), ip, C.c_int, ip,
                                       C.POINTER(stage1.DepthStats)]
    rc = lib.sc_depth_scan_runs(0, ref_len.ctypes.data_as(ip), n_refs, run_ref.ctypes.data_as(ip), start.ctypes.data_as(ip),
                                end.ctypes.data_as(ip), n_runs, 10, iv[0].ctypes.data_as(ip), iv[1].ctypes.data_as(ip),
                                iv[2].ctypes.data_as(ip), sm.ctypes.data_as(C.POINTER(C.c_long)), cn.ctypes.data_as(ip), cap, C.byref(n),
                                C.byref(st))
    assert rc == 0
    k = n.value
    assert int(sm[:k].sum()) == int((end - start + 1).sum())          # every aligned base is counted once in some interval
    # covered positions: from a difference array on the host
    cells = np.zeros(int(ref_len.sum()) + n_refs + 1, dtype=np.int32)
    off = np.concatenate([[0], np.cumsum(ref_len + 1)])[:-1]
    np.add.at(cells, off[run_ref] + start - 1, 1)
    np.add.at(cells, off[run_ref] + end, -1)
    assert int(cn[:k].sum()) == int((np.cumsum(cells) > 0).sum())
    assert np.all(iv[1][:k] >= 1) and np.all(iv[2][:k] <= ref_len[iv[0][:k]]) and np.all(iv[1][:k] <= iv[2][:k])
    key = iv[0][:k].astype(np.int64) * 10000 + iv[1][:k]
    assert np.all(np.diff(key) > 0)                                   # sorted by (reference, start), no overlap
