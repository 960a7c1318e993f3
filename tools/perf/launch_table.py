"""Aggregate a bench.py --dump-launches CSV by (kernel variant, layout, M, N, K): count, total ms, TFLOP/s.  Usage: launch_table.py CSV"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in rows:
    key = (r.get("variant", r["kernel_id"]), r["layout"], int(r["M"]), int(r["N"]), int(r["K"]))
    a = agg[key]; a[0] += 1; a[1] += float(r["ms"]); a[2] += 2.0 * int(r["M"]) * int(r["N"]) * int(r["K"])
tot = sum(a[1] for a in agg.values())
print(f"total {tot:.2f} ms over {len(rows)} launches")
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{key[0]:>6} {key[1]:>3} M={key[2]:6d} N={key[3]:6d} K={key[4]:6d}  n={a[0]:4d}  {a[1]:7.3f} ms  {a[2] / (a[1] * 1e-3) / 1e12:6.1f} TF  {a[1] / a[0] * 1e3:7.1f} us/launch")
