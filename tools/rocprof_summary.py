#!/usr/bin/env python
"""Turn a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel stats table kept under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof/r01_results.db profiles/r01_kernel_stats.txt [forwards]
"""
import sqlite3
import sys


def main(db, out, forwards=None):
    con = sqlite3.connect(db)
    rows = con.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                       "max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = ["# rocprofv3 --kernel-trace --stats summary (durations in microseconds)",
             "# source db: %s ; total kernel time %.1f us over %d dispatches%s" %
             (db, tot, sum(r[1] for r in rows), "" if not forwards else " ; %d forwards -> %.3f ms kernel time / forward"
              % (forwards, tot / forwards / 1e3)),
             "%12s %6s %7s %10s %10s %10s  %s" % ("total_us", "pct", "calls", "avg_us", "min_us", "max_us", "kernel")]
    for name, n, t, avg, mn, mx in rows:
        lines.append("%12.1f %6.2f %7d %10.2f %10.2f %10.2f  %s" % (t, 100 * t / tot, n, avg, mn, mx, name[:160]))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:25]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else None)
