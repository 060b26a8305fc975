#!/bin/bash
out=gpurun_out/r03q
mkdir -p $out
SC_POISON_ROWS=1 timeout -k 10 200 python3 tools/wide_trace_dump.py $out/t.gz 345 349 > $out/o.txt 2> $out/e.txt; grep poison $out/e.txt | head -14 | cut -c1-400; tail -n 2 $out/o.txt | cut -c1-200
