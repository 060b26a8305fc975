#!/usr/bin/env python3
"""Stage-1 depth scan on a synthetic 10^8-base input (66 667 genes of ~1 500 bases, reads of 150 bases at depth ~20):
times of k_depth_mark and k_depth_segments (HIP events inside sc_depth_scan_runs) and the segment kernel's streaming
rate (4 algorithmic bytes per cell of the difference array) against the 8 TB/s HBM peak.  The same leg bench.py
reports as "hbm_bound_kernel".
usage: python3 tools/depth_bench.py [bases] [depth] [repeats]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    bases = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
    depth = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    import bench
    print(json.dumps(bench.depth_scan_leg(0, bases, depth, reps)))


if __name__ == "__main__":
    main()
