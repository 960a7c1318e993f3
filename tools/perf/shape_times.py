"""Aggregates a bench.py --dump-launches CSV: in-situ time per GEMM shape."""
import sys, collections
rows = [l.strip().split(",") for l in open(sys.argv[1]).read().splitlines()[1:]]
agg = collections.defaultdict(lambda: [0, 0.0])
for kid, m, n, k, lay, ms in rows:
    if kid not in ("0", "3"): continue
    a = agg[(kid, int(m), int(n), int(k), int(lay))]; a[0] += 1; a[1] += float(ms)
tot = sum(v[1] for v in agg.values())
print(f"total GEMM ms {tot:.2f}")
for key, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    kid, m, n, k, lay = key
    tf = 2.0 * m * n * k * c / (ms * 1e-3) / 1e12
    print(f"id={kid} M={m:6d} N={n:6d} K={k:6d} lay={lay} calls={c:5d} total={ms:8.3f}ms avg={1e3*ms/c:8.1f}us {tf:6.1f}TF  {100*ms/tot:5.1f}%")
