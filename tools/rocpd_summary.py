"""Summarise a rocprofv3 rocpd (.db) output: per-kernel stats (like --stats) and, if present, PMC sums per kernel.
usage: python tools/rocpd_summary.py <results.db> [--top N]"""
import sqlite3, sys, re, collections

def short(name):
    name = re.sub(r"\(.*$", "", name)
    return name if len(name) < 110 else name[:107] + "..."

def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 25
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, start, end from kernels").fetchall()
    agg = collections.defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    for name, s, e in rows:
        a = agg[short(name)]; d = (e - s) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print(f"# kernel-trace summary of {sys.argv[1]}: {len(rows)} dispatches, {tot/1e3:.3f} ms total GPU kernel time")
    print(f"{'kernel':110s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>10s} {'%':>6s}")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{k:110s} {a[0]:7d} {a[1]/1e3:10.3f} {a[1]/a[0]:10.2f} {a[2]:9.2f} {a[3]:10.2f} {100*a[1]/tot:6.2f}")
    try:
        pm = cur.execute("select kernel_name, counter_name, sum(value), count(*), sum(duration) from counters_collection group by kernel_name, counter_name").fetchall()
    except Exception as ex:
        pm = []
    if pm:
        print("\n# PMC sums per kernel (counter, sum over dispatches, dispatches)")
        by = collections.defaultdict(dict)
        for name, c, v, n, dur in pm:
            by[short(name)][c] = (v, n)
            by[short(name)]["_duration_ns"] = (dur, n)
        for k, d in sorted(by.items(), key=lambda kv: -max(v[0] for v in kv[1].values()))[:top]:
            print(k)
            for c, (v, n) in sorted(d.items()):
                print(f"    {c:28s} {v:20.1f}  over {n} dispatches  ({v/n:.1f} / dispatch)")

if __name__ == "__main__":
    main()
