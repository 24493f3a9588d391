"""Per-kernel summary (calls, total / average / min / max duration) of a rocprofv3 results database
(rocprofv3 --kernel-trace ... writes <name>_results.db): the table committed under profiles/."""
import sqlite3
import sys


def main(path, top=40, split=False):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    name_col = "display_name" if "display_name" in cols else "kernel_name"
    rows = cur.execute(f"select s.{name_col}, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                       f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.{name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("%-90s %8s %12s %12s %12s %12s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for name, calls, tot, avg, mn, mx in rows[:top]:
        print("%-90s %8d %12.1f %12.2f %12.2f %12.2f %6.1f" % (name[:90], calls, tot / 1e3, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
        if split and mx > 1.5 * mn and calls > 1:
            # one kernel launched at several text sizes (the literal scan at 5 GB and 50 GB, small calls of a sweep): a row per
            # group of launches whose durations lie within 25 % of each other, so that every size's average can be read off
            durs = sorted(r[0] for r in cur.execute(f"select d.end-d.start from {kd} d join {ks} s on d.kernel_id = s.id where s.{name_col} = ?", (name,)))
            groups = []
            for d in durs:
                if groups and d <= groups[-1][0] * 1.25:
                    groups[-1].append(d)
                else:
                    groups.append([d])
            for g in groups:
                print("%-90s %8d %12.1f %12.2f %12.2f %12.2f" % ("    launches of %.1f .. %.1f us" % (g[0] / 1e3, g[-1] / 1e3), len(g), sum(g) / 1e3, sum(g) / len(g) / 1e3, g[0] / 1e3, g[-1] / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40, len(sys.argv) > 3 and sys.argv[3] == "split")
