#!/usr/bin/env python3
"""Condense rocprofv3 outputs of `bench.py` into the files kept under profiles/rNN/.

    python3 tools/summarize_profiles.py STATS_DIR FETCH_DIR WRITE_DIR OUT_DIR [TAG]

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
            key = None
            for fam in ("k_level_sample", "k_level_resident", "k_level_any", "k_depth_fused", "k_level_copy", "k_level_update"):
                if fam in name:
                    key = fam
                    break
            if key is None and "sc::k_level(" in name:
                key = "k_level"
            if key is None:
                continue
            n, tot = per.get(key, (0, 0.0))
            per[key] = (n + 1, tot + float(row["Counter_Value"]))
    return {k: (n, tot / n) for k, (n, tot) in per.items()}


def main():
    stats_dir, fetch_dir, write_dir, out = sys.argv[1:5]
    tag = sys.argv[5] if len(sys.argv) > 5 else "bench_config2"
    os.makedirs(out, exist_ok=True)
    shutil.copy(find(stats_dir, "_kernel_stats.csv"), os.path.join(out, tag + "_kernel_stats.csv"))
    # the spread of the level kernels' durations (a stall of the queues -- round 2's pageable copies -- shows as a maximum
    # tens of times the average)
    spread = {}
    with open(find(stats_dir, "_kernel_stats.csv"), newline="") as f:
        for row in csv.DictReader(f):
            if "k_level" in row["Name"]:
                spread[row["Name"]] = {"calls": int(row["Calls"]), "avg_us": float(row["AverageNs"]) / 1e3, "max_us": float(row["MaxNs"]) / 1e3,
                                       "stddev_us": float(row["StdDev"]) / 1e3}
    json.dump(spread, open(os.path.join(out, tag + "_level_kernel_spread.json"), "w"), indent=1)
    with open(find(stats_dir, "_kernel_trace.csv"), "rb") as f, gzip.open(os.path.join(out, tag + "_kernel_trace.csv.gz"), "wb") as g:
        shutil.copyfileobj(f, g)
    fpath, wpath = find(fetch_dir, "_counter_collection.csv"), find(write_dir, "_counter_collection.csv")
    for src, name in ((fpath, tag + "_pmc_fetch_size.csv.gz"), (wpath, tag + "_pmc_write_size.csv.gz")):
        with open(src, "rb") as f, gzip.open(os.path.join(out, name), "wb") as g:
            shutil.copyfileobj(f, g)
    fe, wr = counter_avgs(fpath, "FETCH_SIZE"), counter_avgs(wpath, "WRITE_SIZE")
    summ = {
        "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- <the profiled command> (two separate passes)",
        "unit": "KiB as reported; bytes = value*1024; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide "
                "coalesced reads; other widths uncalibrated, so the doubled figure is an upper bound)",
    }
    for k in sorted(set(fe) & set(wr)):
        summ[k + "_launches"] = fe[k][0]
        summ[k + "_fetch_kib_avg"] = fe[k][1]
        summ[k + "_write_kib_avg"] = wr[k][1]
        summ[k + "_traffic_bytes_per_launch"] = (2.0 * fe[k][1] + wr[k][1]) * 1024.0
    summ["sample_traffic_bytes_per_launch"] = summ.get("k_level_sample_traffic_bytes_per_launch")
    name = "pmc_summary.json" if tag == "bench_config2" else tag + "_pmc_summary.json"
    json.dump(summ, open(os.path.join(out, name), "w"), indent=1)
    print(json.dumps(summ, indent=1))


if __name__ == "__main__":
    main()
