"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table: calls, total, average, MEDIAN,
minimum, maximum (microseconds) and the average over the last 60 % of a kernel's launches ("steady": a short bench run
spends its first launches with the clocks still coming up from idle, which the plain average includes).
    python tools/prof_summary.py gpurun_out/prof/x_results.db > profiles/xxx.txt"""
import sqlite3
import statistics
import sys


def main():
    path = sys.argv[1]
    db = sqlite3.connect(path)
    cur = db.cursor()
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
    print("%-62s %7s %11s %9s %9s %9s %9s %9s %6s" % ("kernel", "calls", "total_us", "avg_us", "median_us", "steady_us", "min_us", "max_us", "%"))
    for name, v in sorted(stats.items(), key=lambda kv: -sum(kv[1])):
        late = v[int(0.4 * len(v)):]
        print("%-62s %7d %11.1f %9.2f %9.2f %9.2f %9.2f %9.2f %6.1f" % (name[:62], len(v), sum(v), sum(v) / len(v), statistics.median(v),
                                                                        sum(late) / len(late), min(v), max(v), 100 * sum(v) / tot))


if __name__ == "__main__":
    main()
