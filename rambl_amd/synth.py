"""Seeded synthetic inputs for the StrainCall path.

A data set is a FASTA file (one record per seed gene) + its .fai + one SAM
*text* file of reads already aligned to those records (the reference consumes
alignments made upstream by bowtie2, /root/reference/StrainCall/
PartialOrderGraph.cpp:94-255 threads them by CIGAR).  Everything is derived
from `random.Random(seed)` so fixtures and bench inputs are reproducible.

Generator: reference = `glen` uniform ACGT;
K strains = reference + `n_sub` substitutions + `n_ins` insertions of 1-3 bp +
`n_del` one-base deletions; strain weights ~ U(0.3,1.3); reads: uniform start,
`rlen` bp, `err` substitution error, forward strand, MAPQ 42, exact CIGAR; SAM
sorted by position.
"""
import os
import random

BASES = "ACGT"


def _mutate_strain(rng, ref, n_sub, n_ins, n_del, ins_len=(1, 3)):
    """Return list of (ref_pos, op, payload) edits, sorted, non-overlapping."""
    glen = len(ref)
    pos = rng.sample(range(20, glen - 20), n_sub + n_ins + n_del)
    edits = []
    for p in pos[:n_sub]:
        alt = rng.choice([b for b in BASES if b != ref[p]])
        edits.append((p, "S", alt))
    for p in pos[n_sub:n_sub + n_ins]:
        ln = rng.randint(*ins_len)
        edits.append((p, "I", "".join(rng.choice(BASES) for _ in range(ln))))
    for p in pos[n_sub + n_ins:]:
        edits.append((p, "D", ""))
    edits.sort()
    return edits


def _strain_columns(ref, edits):
    """Per strain: list of (ref_pos, kind, base) in strain order.
    kind: 'M' aligned base (ref_pos consumed), 'I' inserted base after ref_pos-1,
    'D' deleted ref base (no strain base)."""
    by_pos = {}
    for p, op, payload in edits:
        by_pos.setdefault(p, []).append((op, payload))
    cols = []
    for p, b in enumerate(ref):
        ops = by_pos.get(p, [])
        kind, base = "M", b
        ins_after = ""
        for op, payload in ops:
            if op == "S":
                base = payload
            elif op == "D":
                kind = "D"
            elif op == "I":
                ins_after = payload
        cols.append((p, kind, base))
        for c in ins_after:
            cols.append((p, "I", c))
    return cols


def _read_from_cols(rng, cols, start, rlen, err):
    """Cut `rlen` strain bases starting at strain-column index `start` (which
    must be an 'M' column); returns (ref_pos0, cigar, seq) or None."""
    seq = []
    ops = []  # run-length (op, len)
    i = start
    n = len(cols)
    ref_pos0 = cols[start][0]

    def push(op):
        if ops and ops[-1][0] == op:
            ops[-1][1] += 1
        else:
            ops.append([op, 1])

    while len(seq) < rlen and i < n:
        p, kind, base = cols[i]
        if kind == "D":
            push("D")
        else:
            if err > 0 and rng.random() < err:
                base = rng.choice([b for b in BASES if b != base])
            seq.append(base)
            push("M" if kind == "M" else "I")
        i += 1
    # an alignment may not end in I or D: trim trailing non-M ops
    while ops and ops[-1][0] != "M":
        op, ln = ops.pop()
        if op == "I":
            del seq[-ln:]
    if not ops or len(seq) == 0:
        return None
    cigar = "".join("%d%s" % (ln, op) for op, ln in ops)
    return ref_pos0, cigar, "".join(seq)


def make_gene(seed, glen=1500, n_strains=3, n_reads=10000, rlen=150, err=0.005,
              n_sub=45, n_ins=2, n_del=2, name=None, paired=False, mapq=42,
              ins_len=(1, 3), shared_ins_site=False):
    """One seed gene + its reads.  Returns dict(name, ref, sam_lines, strains).

    `shared_ins_site=True` forces every strain to carry an insertion of a
    different length after the same reference position, which makes the
    insertion canoniser run the sum-of-pairs MSA
    (/root/reference/StrainCall/PartialOrderGraph.cpp:500-549)."""
    rng = random.Random(seed)
    name = name or ("gene%d" % seed)
    ref = "".join(rng.choice(BASES) for _ in range(glen))
    strains = []
    for k in range(n_strains):
        edits = _mutate_strain(rng, ref, n_sub, n_ins, n_del, ins_len)
        if shared_ins_site:
            site = glen // 2
            edits = [e for e in edits if abs(e[0] - site) > 3]
            payload = "".join(rng.choice(BASES) for _ in range(1 + 2 * k))
            edits.append((site, "I", payload))
            edits.sort()
        strains.append(edits)
    weights = [rng.uniform(0.3, 1.3) for _ in range(n_strains)]
    cols = [_strain_columns(ref, e) for e in strains]
    recs = []
    serial = 0
    n_frag = n_reads // 2 if paired else n_reads
    for _ in range(n_frag):
        k = rng.choices(range(n_strains), weights)[0]
        c = cols[k]
        mstarts = None
        if paired:
            frag = rng.randint(int(rlen * 1.6), int(rlen * 2.6))
            hi = max(1, len(c) - frag)
            s0 = rng.randrange(0, hi)
            starts = [s0, min(len(c) - 1, s0 + frag - rlen)]
        else:
            hi = max(1, len(c) - rlen + 1)
            starts = [rng.randrange(0, hi)]
        mates = []
        for s0 in starts:
            while s0 < len(c) and c[s0][1] != "M":
                s0 += 1
            if s0 >= len(c):
                continue
            r = _read_from_cols(rng, c, s0, rlen, err)
            if r is not None:
                mates.append(r)
        qn = "r%s_%d" % (name, serial)
        serial += 1
        for mi, (p0, cigar, seq) in enumerate(mates):
            if paired and len(mates) == 2:
                flag = 65 if mi == 0 else 129
            else:
                flag = 0
            recs.append((p0, qn, flag, cigar, seq))
    recs.sort(key=lambda r: (r[0], r[1], r[2]))
    lines = []
    for p0, qn, flag, cigar, seq in recs:
        lines.append("\t".join([qn, str(flag), name, str(p0 + 1), str(mapq), cigar,
                                "*", "0", "0", seq, "I" * len(seq)]))
    return dict(name=name, ref=ref, sam_lines=lines, strains=strains, weights=weights)


def write_dataset(outdir, genes, prefix="seed_otus", line_width=60):
    """Write <prefix>.fasta, <prefix>.fasta.fai and reads.sam under outdir.
    Returns (fasta_path, sam_path)."""
    os.makedirs(outdir, exist_ok=True)
    fa = os.path.join(outdir, prefix + ".fasta")
    sam = os.path.join(outdir, "reads.sam")
    off = 0
    with open(fa, "w") as f, open(fa + ".fai", "w") as fi:
        for g in genes:
            hdr = ">%s\n" % g["name"]
            f.write(hdr)
            off += len(hdr)
            ref = g["ref"]
            fi.write("%s\t%d\t%d\t%d\t%d\n" % (g["name"], len(ref), off, line_width, line_width + 1))
            for i in range(0, len(ref), line_width):
                chunk = ref[i:i + line_width] + "\n"
                f.write(chunk)
                off += len(chunk)
    with open(sam, "w") as f:
        for g in genes:
            f.write("@SQ\tSN:%s\tLN:%d\n" % (g["name"], len(g["ref"])))
        for g in genes:
            for ln in g["sam_lines"]:
                f.write(ln + "\n")
    return fa, sam


def config2(outdir, seed=21, n_reads=10000, glen=1500, n_strains=3, **kw):
    """BASELINE.json configs[1]: 10k x 150 bp reads vs one 1500 bp gene."""
    g = make_gene(seed, glen=glen, n_strains=n_strains, n_reads=n_reads, name="gene%d" % seed, **kw)
    return write_dataset(outdir, [g]) + (g,)


def config3(outdir, seeds=range(100, 200), n_reads=(2000, 10000), **kw):
    """BASELINE.json configs[2]: many independent seed genes in one FASTA/SAM."""
    genes = []
    for s in seeds:
        rng = random.Random(s * 7919 + 1)
        n = rng.randint(*n_reads)
        genes.append(make_gene(s, n_reads=n, name="gene%d" % s, **kw))
    return write_dataset(outdir, genes) + (genes,)
