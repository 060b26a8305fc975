#!/bin/bash
# The unthinned region's levels by phase.
out=gpurun_out/r03v
mkdir -p $out; rm -f $out/dlv.*
SC_LEVEL_LOG=$out/dlv timeout -k 10 600 python3 tools/unthinned_probe.py 100000 /tmp/unthinned > $out/deep.txt 2> $out/deep.err || { echo "probe failed"; tail $out/deep.err; exit 1; }
python3 tools/level_log_summary.py $out/dlv
python3 - <<'PY'
import glob
rows=[]
for fn in glob.glob("gpurun_out/r03v/dlv.*"):
    for l in open(fn):
        f=l.split()
        if f and f[0]=="h":
            d={}
            i=0
            while i+1<len(f):
                if f[i]=="ph": d["ph"]=[float(x) for x in f[i+1:i+6]]; i+=6
                else:
                    try: d[f[i]]=float(f[i+1])
                    except ValueError: pass
                    i+=2
            rows.append(d)
h=max(r["h"] for r in rows); rows=[r for r in rows if r["h"]==h]
import collections
by=collections.defaultdict(list)
for r in rows: by[(int(r["mode"]), r["chain_us"]>0)].append(r)
for k,v in sorted(by.items()):
    tot=sum(r["level_us"] for r in v)
    print("mode %d chain %s: n %d total %.1f ms avg %.1f us; phases avg: stage %.1f copies %.1f update %.1f slots %.1f table %.1f ; S avg %.1f Q avg %.0f; wait-level %.1f host %.1f" % (k[0],k[1],len(v),tot/1e3,tot/len(v),
      sum(r["ph"][0] for r in v)/len(v), sum(r["ph"][1]-r["ph"][0] for r in v)/len(v), sum(r["ph"][2]-r["ph"][1] for r in v)/len(v), sum(r["ph"][3]-r["ph"][2] for r in v)/len(v), sum(r["ph"][4]-r["ph"][3] for r in v)/len(v),
      sum(r["S"] for r in v)/len(v), sum(r["Q"] for r in v)/len(v), sum(r["wait_us"]-r["level_us"] for r in v)/len(v), sum(r["host_us"] for r in v)/len(v)))
big=sorted(rows,key=lambda r:-r["level_us"])[:8]
for r in big: print({k:r[k] for k in ("mode","S","Q","n","level_us","ncopy","ph")})
PY
rm -f $out/dlv.*
