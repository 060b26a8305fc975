#!/usr/bin/env python3
"""BASELINE.json configs[3] with NO thinning (-D 100000: every one of the 10^6 reads reaches the graph; ~590 000
distinct reads, ~100 000 read copies per level).  The reference cannot finish this case (its edge support is
quadratic in the pool sizes, tests/golden/make_golden_config4.py), so there is no expected output: the probe reports
time, memory and the contigs, and checks only what can be checked without one -- two runs agree -- and counts the
contigs of the -D 3000 reference output (tests/golden/config4_full_D3000) that are found again.
usage: python3 tools/unthinned_probe.py [depth] [workdir]"""
import io
import json
import os
import resource
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def contigs(text):
    out, name = [], None
    for line in text.splitlines():
        if line.startswith(">"):
            name = line
        elif name is not None:
            out.append((name, line))
            name = None
    return out


def main():
    depth = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    work = sys.argv[2] if len(sys.argv) > 2 else "/tmp/unthinned"
    os.makedirs(work, exist_ok=True)
    from rambl_amd import cli, synth
    t0 = time.time()
    gene = synth.make_gene(4, glen=1500, n_strains=50, n_reads=1000000, name="deep4m")
    fa, sam = synth.write_dataset(work, [gene])
    print("dataset %.1f s" % (time.time() - t0), flush=True)
    argv = ["-r", "deep4m:1-1500", "-q", "0", "-D", str(depth), "-I", "13", "-l", "70", "-t", "0.02", "-d", "0.02", "-w", "5000", fa, sam]
    runs = []
    os.environ["SC_STATS"] = "1"
    os.environ["SC_SYNC_LOG"] = "1"
    for rep in range(2):
        out, err = io.StringIO(), io.StringIO()
        t0 = time.time()
        rc = cli.main(argv, out=out, err=err)
        dt = time.time() - t0
        print("run %d: rc %d, %.1f s, max rss %.1f GB" % (rep, rc, dt, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1048576.0), flush=True)
        for line in err.getvalue().splitlines():
            if line.startswith("sc_stats"):
                print(line[:1500], flush=True)
        if rc != 0:
            print(err.getvalue()[-2000:])
            return 1
        runs.append(out.getvalue())
    cs = contigs(runs[0])
    print(json.dumps({"depth": depth, "contigs": len(cs), "names": [n for n, _ in cs][:60], "runs_agree": runs[0] == runs[1]}))
    gold = os.path.join(ROOT, "tests", "golden", "config4_full_D3000", "expected.fa")
    if os.path.exists(gold):
        have = {s for _, s in cs}
        ref = contigs(open(gold).read())
        print("contigs of the -D 3000 reference output found again: %d of %d" % (sum(1 for _, s in ref if s in have), len(ref)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
