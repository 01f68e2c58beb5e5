"""Print the kernel timeline (start offset, duration, gap to previous) of the last N dispatches in a
rocprofv3 rocpd database.  python tools/prof_timeline.py db [N]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
cur = db.cursor()
kcols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "kernel_name" if "kernel_name" in kcols else kcols[1]
rows = cur.execute("select s.%s, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                   "on d.kernel_id = s.id order by d.start" % name_col).fetchall()
rows = rows[-n:]
t0 = rows[0][1]
prev_end = None
for name, st, en in rows:
    gap = (st - prev_end) / 1e3 if prev_end else 0.0
    print("%10.1f us  dur %8.1f us  gap %7.1f us  %s" % ((st - t0) / 1e3, (en - st) / 1e3, gap, name.split("(")[0][:60]))
    prev_end = en
