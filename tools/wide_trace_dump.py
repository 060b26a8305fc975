#!/usr/bin/env python3
"""Diagnostics: the per-level trace of the wide_cap120 region (tests/golden/wide_cap120) between two levels, as the product
walks it.  usage: wide_trace_dump.py OUT.gz LO HI [max_candidates]"""
import gzip
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    from rambl_amd import capi, cli, synth
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "wide_cap120", "meta.json")))
    cap = int(sys.argv[4]) if len(sys.argv) > 4 else meta["max_candidates"]
    d = tempfile.mkdtemp(prefix="wide_")
    gene = synth.make_gene(77, glen=600, n_strains=100, n_reads=30000, name="wide", n_sub=12, err=0.01)
    fa, sam = synth.write_dataset(d, [gene])
    pa = cli.parse_cmd_line(meta["argv"] + [fa, sam])
    params = capi.default_params(float(pa.error_rate), float(pa.tau), float(pa.diff_rate), want_trace=True)
    params.max_candidates = cap
    with capi.Context(0, 1) as ctx, gzip.open(out, "wt") as f:
        for w, r in cli.load_regions(pa):
            res = ctx.wait(ctx.submit(r, params), want_trace=True)
            keep = False
            lines = iter(res.trace.splitlines(True))
            for line in lines:
                if line.startswith("------"):
                    when, lvl = next(lines), next(lines)
                    keep = lo <= int(lvl.split(":")[1]) <= hi
                    if keep:
                        f.write(line + when + lvl)
                elif keep:
                    f.write(line)
            print(json.dumps({"kind_levels": res.stats["kind_levels"], "slow": res.stats["slow_draws"], "exact": res.stats["exact_draws"],
                              "draws": res.stats["draws"]}))


if __name__ == "__main__":
    main()
