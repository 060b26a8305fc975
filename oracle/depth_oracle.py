"""TEST INFRASTRUCTURE ONLY (oracle) -- parity unpinned.

Plain-Python restatement of rambl.py stage 1, /root/reference/scripts/coverage_all_samples.py:21-186, which is a
pipeline of external tools: `samtools depth <bams>` (:29), awk summing the per-file columns into `chrom_pos_pos+1 <TAB>
sum` (:37), joins of the batches (:131-139), per-position sums with bc (:62-66), `sort -k1,1n -k2,2n` and
`bedtools merge -c 4 -o mean -d 10` (:84-85, :173).  samtools (0.1.19) and bedtools are not in the image and the
reference holds no fixture for this stage, so the tools' behaviour is restated from their documentation:

  samtools depth   per-base depth of the reads whose flag has none of UNMAP, SECONDARY, QCFAIL, DUP (0x704); a read
                   counts at the reference positions of its M / = / X operations (a deleted or skipped base does
                   not count); positions of depth 0 in every file are not printed.  Its per-file depth cap (8000) is
                   not modelled.
  bedtools merge   -d 10: records [p, p+1) of one chromosome are merged while the next starts at most 10 bases behind
                   the end of the merged block; -c 4 -o mean: mean of column 4 over the merged records.

Only tests/ and tools/ import this module; the product path (rambl_amd/stage1.py -> sc_depth_scan) never does.
"""
import re

_CIG = re.compile(r"(\d+)([MIDNSHP=X])")


def depth_of_records(records, ref_len):
    """records: iterable of (flag, pos, cigar) of ONE reference; returns the list depth[1..ref_len] (index 0 unused)."""
    depth = [0] * (ref_len + 2)
    for flag, pos, cigar in records:
        if flag & 0x704:
            continue
        p = pos
        for n, op in _CIG.findall(cigar):
            n = int(n)
            if op in "M=X":
                for q in range(max(p, 1), min(p + n - 1, ref_len) + 1):
                    depth[q] += 1
                p += n
            elif op in "DN":
                p += n
    return depth


def merge_mean(depth, ref_len, max_gap=10):
    """bedtools merge -d max_gap -c 4 -o mean over the covered one-base records: [(start, end, sum, n)], 1-based inclusive."""
    out = []
    cur = None
    for p in range(1, ref_len + 1):
        if depth[p] <= 0:
            continue
        if cur is not None and p - cur[1] - 1 <= max_gap:
            cur[1] = p
            cur[2] += depth[p]
            cur[3] += 1
        else:
            if cur is not None:
                out.append(tuple(cur))
            cur = [p, p, depth[p], 1]
    if cur is not None:
        out.append(tuple(cur))
    return out


def stage1(files, refs, max_gap=10):
    """files: list of {ref name: [(flag, pos, cigar), ...]}; refs: [(name, length)] in output order.
    -> [(ref index, start, end, depth sum, covered positions)]."""
    out = []
    for ri, (name, ln) in enumerate(refs):
        total = [0] * (ln + 2)
        for f in files:
            d = depth_of_records(f.get(name, ()), ln)
            for p in range(1, ln + 1):
                total[p] += d[p]
        for s, e, sm, n in merge_mean(total, ln, max_gap):
            out.append((ri, s, e, sm, n))
    return out
