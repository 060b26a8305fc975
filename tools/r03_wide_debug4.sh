#!/bin/bash
out=gpurun_out/r03o
mkdir -p $out
run() { name=$1; shift; env "$@" timeout -k 10 200 python3 tools/wide_trace_dump.py $out/trace_$name.txt.gz 345 349 > $out/wide_$name.txt 2>&1; tail -n 1 $out/wide_$name.txt | cut -c1-200; }
run full1 SC_FULL_ROW_COPY=1
run full2 SC_FULL_ROW_COPY=1
run poison SC_POISON_ROWS=1
python3 - <<'PY'
import gzip, sys
sys.path.insert(0, "tests")
import sc_testlib as T
for n in ("full1", "full2", "poison"):
    b = [x for x in T.parse_trace(gzip.open("gpurun_out/r03o/trace_%s.txt.gz" % n, "rt").read()) if x[0].startswith("after") and x[1] == 347]
    print(n, len(b[0][2]), [round(v, 4) for _, v in b[0][2][8:13]])
PY
