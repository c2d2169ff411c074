"""Sum of every PMC counter over ALL kernel dispatches of a rocprofv3 rocpd DB:  pmc_total.py results.db [divide_by]
(divide_by = number of identical calls the profiled process made, tools/predict_pmc_probe.py: 3)."""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tabs if t.startswith(p)][0]  # noqa: E731
pe, pi, disp = T("rocpd_pmc_event"), T("rocpd_info_pmc"), T("rocpd_kernel_dispatch")
names = {r[0]: r[1] for r in cur.execute(f"select id, name from {pi}")}
tot = defaultdict(float)
for pid, v in cur.execute(f"select pmc_id, value from {pe}"):
    tot[names.get(pid, str(pid))] += v
ndisp = list(cur.execute(f"select count(*) from {disp}"))[0][0]
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
for name, v in sorted(tot.items()):
    print(f"{name}: total {v:.6e} over {ndisp} dispatches; / {div:g} calls = {v / div:.6e}")
