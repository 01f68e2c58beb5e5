"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table.
    python tools/prof_summary.py gpurun_out/prof/x_results.db [--skip N] > profiles/xxx.txt"""
import sqlite3
import sys


def main():
    path = sys.argv[1]
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    kcols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "kernel_name" if "kernel_name" in kcols else ("display_name" if "display_name" in kcols else kcols[1])
    q = ("select s.%s, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
         "order by d.start" % name_col)
    rows = cur.execute(q).fetchall()
    stats = {}
    for name, st, en in rows:
        name = name.split("(")[0]
        stats.setdefault(name, []).append((en - st) / 1e3)
    tot = sum(sum(v) for v in stats.values())
    print("%-70s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for name, v in sorted(stats.items(), key=lambda kv: -sum(kv[1])):
        print("%-70s %8d %12.1f %10.2f %10.2f %10.2f %6.1f" % (name[:70], len(v), sum(v), sum(v) / len(v), min(v), max(v),
                                                                 100 * sum(v) / tot))


if __name__ == "__main__":
    main()
