"""Print the kernel timeline (start offset, duration, queue, grid) of one region of a rocprofv3 rocpd DB.
usage: timeline.py results.db <first-marker-kernel-substring> <occurrence> [max_rows]"""
import re
import sqlite3
import sys


def rows_of(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = (f"select s.display_name,d.start,d.end,d.queue_id,d.grid_size_x,d.grid_size_y,d.workgroup_size_x "
         f"from {disp} d join {sym} s on d.kernel_id=s.id order by d.start")
    out = []
    for r in cur.execute(q):
        name = re.sub(r"\(.*", "", re.sub(r"\(anonymous namespace\)::", "", r[0]))[:44]
        out.append((name,) + tuple(r[1:]))
    return out


if __name__ == "__main__":
    rows = rows_of(sys.argv[1])
    marker, occ = sys.argv[2], int(sys.argv[3])
    nmax = int(sys.argv[4]) if len(sys.argv) > 4 else 200
    idx = [i for i, r in enumerate(rows) if marker in r[0]]
    s = idx[occ]
    t0 = rows[s][1]
    for r in rows[s:s + nmax]:
        print(f"{(r[1]-t0)/1e3:9.1f} {(r[2]-r[1])/1e3:8.1f} q{r[3]} grid {r[4]//max(r[6],1):5d}x{r[5]} {r[0]}")
