"""Summarise a rocprofv3 rocpd sqlite DB: per-kernel count / total / avg / min / max (us)."""
import re
import sqlite3
import sys


def main(path, top=40):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({disp})")]
    scols = [r[1] for r in cur.execute(f"pragma table_info({sym})")]
    namecol = "display_name" if "display_name" in scols else "kernel_name"
    q = f"select s.{namecol}, d.start, d.end from {disp} d join {sym} s on d.kernel_id = s.id order by d.start"
    rows = list(cur.execute(q))
    agg = {}
    for name, st, en in rows:
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"\(.*", "", name)[:70]
        a = agg.setdefault(name, [0, 0.0, 1e30, 0.0])
        dur = (en - st) / 1e3
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    tot = sum(a[1] for a in agg.values())
    span = (rows[-1][2] - rows[0][1]) / 1e3 if rows else 0
    print(f"# {path}: {len(rows)} dispatches, sum of kernel time {tot:.1f} us, first-start..last-end span {span:.1f} us")
    print(f"{'kernel':70s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{name:70s} {a[0]:6d} {a[1]:12.1f} {a[1]/a[0]:10.2f} {a[2]:10.2f} {a[3]:10.2f} {100*a[1]/tot:6.1f}")
    return rows


if __name__ == "__main__":
    main(sys.argv[1])
