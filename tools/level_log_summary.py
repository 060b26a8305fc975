#!/usr/bin/env python3
"""Summarises SC_LEVEL_LOG files (one per worker slot): per level, time inside the kernel, time between the
request and the observed stamp, host time between levels.  usage: level_log_summary.py PREFIX"""
import glob
import sys


def rows(prefix):
    for fn in glob.glob(prefix + ".*"):
        for l in open(fn):
            f = l.split()
            if not f or f[0] != "h":
                continue
            d = {}
            i = 0
            try:
                while i < len(f):
                    if f[i] == "ph":
                        d["ph"] = [float(x) for x in f[i + 1:i + 6]]
                        i += 6
                    else:
                        d[f[i]] = float(f[i + 1])
                        i += 2
            except (ValueError, IndexError):
                continue
            yield d


def avg(xs):
    xs = list(xs)
    return sum(xs) / max(len(xs), 1)


def main():
    rs = list(rows(sys.argv[1]))
    first = min(r["h"] for r in rs)
    warm = max(r["h"] for r in rs) // 2
    rs = [r for r in rs if r["h"] > warm] or rs
    ch = [r for r in rs if r["chain_us"] > 0]
    pl = [r for r in rs if r["chain_us"] == 0]
    for name, g in (("sampler", ch), ("plain", pl)):
        print("%-8s n %6d  kernel %.1f us (chain %.1f: stage %.1f copies %.1f update %.1f slots %.1f table %.1f)  request->stamp seen %.1f"
              "  (outside kernel %.1f; request->launched %.1f, batch %.1f)  host between levels %.1f" % (
                  name, len(g), avg(r["level_us"] for r in g), avg(r["chain_us"] for r in g), avg(r["ph"][0] for r in g),
                  avg(r["ph"][1] - r["ph"][0] for r in g), avg(r["ph"][2] - r["ph"][1] for r in g), avg(r["ph"][3] - r["ph"][2] for r in g),
                  avg(r["ph"][4] - r["ph"][3] for r in g), avg(r["wait_us"] for r in g), avg(r["wait_us"] - r["level_us"] for r in g),
                  avg(r.get("pend_us", 0) for r in g), avg(r.get("batch", 0) for r in g), avg(r["host_us"] for r in g)))


if __name__ == "__main__":
    main()
