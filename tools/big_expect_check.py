#!/usr/bin/env python3
"""GPU side of tools/big_expect_make.py: the product on every precomputed large scenario, FASTA byte for
byte and the per-level signature (strain count, abundance sum to 1e-9).
Usage: python3 tools/big_expect_check.py [--inflight R]   (with --inflight: all scenarios through one context, R at a time, FASTA only)"""
import sys, os, json, gzip, tempfile, glob, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sc_testlib as T
bad = n = 0
t0 = time.time()
if "--inflight" in sys.argv:
    # every scenario in one context, R regions in flight: their levels share the level server's batches (mixed kinds)
    R = int(sys.argv[sys.argv.index("--inflight") + 1])
    from rambl_amd import capi, cli, stage5
    recs = [json.load(gzip.open(p, "rt")) for p in sorted(glob.glob(os.path.join(ROOT, "tools", "_big_expect", "*.json.gz")))]
    recs = [r for r in recs if not r.get("crashed")]
    prepared = []
    for rec in recs:
        d = tempfile.mkdtemp(prefix="bci%d_" % rec["seed"])
        args, kw = T.big_case(rec["seed"], d)
        pa = cli.parse_cmd_line(args)
        prepared.append((pa, cli.load_regions(pa)))
    print("prepared %d scenarios %.0fs" % (len(prepared), time.time() - t0), flush=True)
    errors = []
    with capi.Context(0, R) as ctx:
        t1 = time.time()
        texts, _ = stage5.run_regions(ctx, prepared, R, None, errors)
        dt = time.time() - t1
    for rec, got in zip(recs, texts):
        if got != rec["fasta"]:
            bad += 1
            print("seed %d FAILED: FASTA differs" % rec["seed"], flush=True)
    print("checked %d precomputed big scenarios with %d in flight (%.1f s on the device path), %d failures, %d region errors" % (
        len(recs), R, dt, bad, len(errors)))
    sys.exit(1 if bad else 0)
for path in sorted(glob.glob(os.path.join(ROOT, "tools", "_big_expect", "*.json.gz"))):
    rec = json.load(gzip.open(path, "rt"))
    seed = rec["seed"]
    if rec.get("crashed"):
        continue
    d = tempfile.mkdtemp(prefix="bc%d_" % seed)
    args, kw = T.big_case(seed, d)
    trf = os.path.join(d, "t.txt")
    n += 1
    try:
        got = T.run_product(args, trace_file=trf)
        assert got == rec["fasta"], "FASTA differs"
        sig = []
        for when, level, rows in T.parse_trace(open(trf).read()):
            sig.append([when[0], level, len(rows), sum(a for _, a in rows if a == a)])
        assert len(sig) == len(rec["sig"]), "block count"
        for a, b in zip(sig, rec["sig"]):
            assert a[:3] == b[:3], ("structure", a, b)
            assert abs(a[3] - b[3]) <= 1e-9 * max(abs(b[3]), 1e-300), ("abundance sum", a, b)
        print("seed %d ok %.0fs" % (seed, time.time() - t0), flush=True)
    except Exception as e:
        bad += 1
        print("seed %d FAILED: %s | %r" % (seed, str(e)[:200], kw), flush=True)
print("checked %d precomputed big scenarios, %d failures" % (n, bad))
