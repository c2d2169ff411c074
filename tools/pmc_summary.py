"""Summarise a rocprofv3 --pmc rocpd DB: per kernel, launches / avg duration / per-launch counter sums, and the
largest launches individually (counter values summed over all instances of a dispatch)."""
import re
import sqlite3
import sys
from collections import defaultdict


def main(path, top=12):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    T = lambda p: [t for t in tabs if t.startswith(p)][0]
    disp, sym, pe, pi = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"), T("rocpd_pmc_event"), T("rocpd_info_pmc")
    names = {r[0]: r[1] for r in cur.execute(f"select id, name from {pi}")}
    vals = defaultdict(lambda: defaultdict(float))
    for ev, pid, v in cur.execute(f"select event_id, pmc_id, value from {pe}"):
        vals[ev][names.get(pid, str(pid))] += v
    rows = list(cur.execute(f"select s.display_name, d.start, d.end, d.event_id, d.grid_size_x, d.workgroup_size_x "
                            f"from {disp} d join {sym} s on d.kernel_id = s.id order by d.start"))
    counters = sorted({c for v in vals.values() for c in v})
    agg = {}
    for name, st, en, ev, gx, wx in rows:
        name = re.sub(r"\(.*", "", re.sub(r"\(anonymous namespace\)::", "", name))[:48]
        a = agg.setdefault(name, [0, 0.0, defaultdict(float)])
        a[0] += 1; a[1] += (en - st) / 1e3
        for c in counters:
            a[2][c] += vals[ev].get(c, 0.0)
    print(f"# {path}: {len(rows)} dispatches; counters: {', '.join(counters)}  (values summed over instances, averaged per launch)")
    print(f"{'kernel':48s} {'calls':>6s} {'avg_us':>10s} " + " ".join(f"{c[:24]:>24s}" for c in counters))
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{name:48s} {a[0]:6d} {a[1]/a[0]:10.2f} " + " ".join(f"{a[2][c]/a[0]:24.4g}" for c in counters))
    print("\n# largest launches")
    big = sorted(rows, key=lambda r: -(r[2] - r[1]))[:top]
    for name, st, en, ev, gx, wx in big:
        name = re.sub(r"\(.*", "", re.sub(r"\(anonymous namespace\)::", "", name))[:40]
        print(f"{name:40s} grid {gx//max(wx,1):6d} dur_us {(en-st)/1e3:10.1f} " + " ".join(f"{c}={vals[ev].get(c,0):.4g}" for c in counters))


if __name__ == "__main__":
    main(sys.argv[1])
