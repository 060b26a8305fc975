#!/bin/bash
out=$(pwd)/gpurun_out/r03l
mkdir -p $out
cd tools/_exp/r02tree && timeout -k 10 300 python3 tools/wide_trace_dump.py $out/trace_r02.txt.gz 345 349 > $out/wide_r02.txt 2>&1; tail -n 2 $out/wide_r02.txt | cut -c1-300
cd ../../.. && python3 - <<'PY'
import gzip, sys
sys.path.insert(0, "tests")
import sc_testlib as T
b = [x for x in T.parse_trace(gzip.open("gpurun_out/r03l/trace_r02.txt.gz", "rt").read()) if x[0].startswith("after") and x[1] == 347]
print("r02 build", len(b[0][2]), [round(v, 4) for _, v in b[0][2][8:13]])
PY
