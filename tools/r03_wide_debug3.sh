#!/bin/bash
out=gpurun_out/r03n
mkdir -p $out
for lv in 346 347 348; do SC_DEBUG_LEVEL=$lv timeout -k 10 200 python3 tools/wide_trace_dump.py $out/t.gz 347 347 2> $out/dbg_$lv.txt > /dev/null; cat $out/dbg_$lv.txt | cut -c1-900; done
