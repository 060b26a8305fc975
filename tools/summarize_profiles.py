#!/usr/bin/env python3
"""Condense rocprofv3 outputs of `bench.py` into the files kept under profiles/rNN/.

    python3 tools/summarize_profiles.py STATS_DIR FETCH_DIR WRITE_DIR OUT_DIR

STATS_DIR: `rocprofv3 --kernel-trace --stats --output-format csv` run; FETCH_DIR / WRITE_DIR: separate
`--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs (`--kernel-trace --output-format csv`).  Writes
kernel_stats.csv, kernel_trace.csv.gz, pmc_*.csv.gz and pmc_summary.json (per-kernel averages; FETCH_SIZE
doubled as MI355X_MICROARCH.md prescribes for gfx950, KiB -> bytes).
"""
import csv
import glob
import gzip
import json
import os
import shutil
import sys


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
    if not hits:
        raise SystemExit("no *%s under %s" % (suffix, d))
    return hits[-1]


def counter_avgs(path, counter):
    per = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"]
            key = "k_chain_w" if "k_chain_w" in name else ("k_level" if "sc::k_level(" in name else None)
            if key is None:
                continue
            n, tot = per.get(key, (0, 0.0))
            per[key] = (n + 1, tot + float(row["Counter_Value"]))
    return {k: (n, tot / n) for k, (n, tot) in per.items()}


def main():
    stats_dir, fetch_dir, write_dir, out = sys.argv[1:5]
    os.makedirs(out, exist_ok=True)
    shutil.copy(find(stats_dir, "_kernel_stats.csv"), os.path.join(out, "bench_config2_kernel_stats.csv"))
    with open(find(stats_dir, "_kernel_trace.csv"), "rb") as f, gzip.open(os.path.join(out, "bench_config2_kernel_trace.csv.gz"), "wb") as g:
        shutil.copyfileobj(f, g)
    fpath, wpath = find(fetch_dir, "_counter_collection.csv"), find(write_dir, "_counter_collection.csv")
    for src, name in ((fpath, "pmc_fetch_size.csv.gz"), (wpath, "pmc_write_size.csv.gz")):
        with open(src, "rb") as f, gzip.open(os.path.join(out, name), "wb") as g:
            shutil.copyfileobj(f, g)
    fe, wr = counter_avgs(fpath, "FETCH_SIZE"), counter_avgs(wpath, "WRITE_SIZE")
    summ = {
        "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu (two separate passes)",
        "unit": "KiB as reported; bytes = value*1024; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide "
                "coalesced reads; other widths uncalibrated, so the doubled figure is an upper bound)",
    }
    for k in ("k_chain_w", "k_level"):
        if k in fe and k in wr:
            summ[k + "_launches"] = fe[k][0]
            summ[k + "_fetch_kib_avg"] = fe[k][1]
            summ[k + "_write_kib_avg"] = wr[k][1]
            summ[k + "_traffic_bytes_per_launch"] = (2.0 * fe[k][1] + wr[k][1]) * 1024.0
    summ["sample_traffic_bytes_per_launch"] = summ.get("k_chain_w_traffic_bytes_per_launch")
    json.dump(summ, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
    print(json.dumps(summ, indent=1))


if __name__ == "__main__":
    main()
