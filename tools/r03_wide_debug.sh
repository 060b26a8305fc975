#!/bin/bash
out=gpurun_out/r03k
mkdir -p $out
run() { name=$1; shift; env "$@" timeout -k 10 200 python3 tools/wide_trace_dump.py $out/trace_$name.txt.gz 345 349 > $out/wide_$name.txt 2>&1; tail -n 1 $out/wide_$name.txt | cut -c1-300; }
run default SC_X=0
run fullLDS SC_LDS_EXACT=0
run resident SC_RESIDENT=1
run cap80 SC_X=0
SC_LEVEL_LOG=$out/lvl timeout -k 10 200 python3 tools/wide_trace_dump.py $out/trace_log.txt.gz 345 349 > /dev/null 2>&1
sed -n 340,352p $out/lvl.0 | cut -c1-260
python3 - <<'PY'
import gzip, sys
sys.path.insert(0, "tests")
import sc_testlib as T
for n in ("default", "fullLDS", "resident"):
    b = [x for x in T.parse_trace(gzip.open("gpurun_out/r03k/trace_%s.txt.gz" % n, "rt").read()) if x[0].startswith("after") and x[1] == 347]
    print(n, len(b[0][2]), [round(v, 4) for _, v in b[0][2][8:13]])
PY
