"""Summarise a rocprofv3 --kernel-trace --stats sqlite database (rocpd schema) per kernel:
calls, total/avg/min/max duration.  Usage: python tools/rocprof_summary.py results.db [out.txt]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name_col} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = [f"{'kernel':<58} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}"]
    for n, c, s, a, mn, mx in rows:
        short = n.split("(")[0][:58]
        lines.append(f"{short:<58} {c:>7} {s/1e6:>10.3f} {a/1e3:>10.2f} {mn/1e3:>10.2f} {mx/1e3:>10.2f} {100*s/tot:>6.2f}")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            f.write(out)
    print(out)


if __name__ == "__main__":
    main()
