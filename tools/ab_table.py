#!/usr/bin/env python
"""Per-kernel A/B table from two rocprofv3 kernel traces (rocpd sqlite) taken inside ONE gpurun call.

    python tools/ab_table.py A_results.db B_results.db out.txt [labelA labelB]
Kernels are matched by name; avg / min duration in microseconds; only kernels whose total differs by > 0.5 % of the sum are
flagged.  The boxes of the pool differ by 12 %, so two traces from different calls must not be compared with this."""
import sqlite3
import sys


def stats(db):
    con = sqlite3.connect(db)
    return {r[0]: r[1:] for r in con.execute(
        "select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3 from kernels group by name")}


def main(a, b, out, la="A", lb="B"):
    sa, sb = stats(a), stats(b)
    ta, tb = sum(v[1] for v in sa.values()), sum(v[1] for v in sb.values())
    lines = ["# per-kernel A/B, one gpurun call, same box; durations in us; %s = %s ; %s = %s" % (la, a, lb, b),
             "# total kernel time: %s %.1f us, %s %.1f us (%+.2f %%)" % (la, ta, lb, tb, 100 * (tb - ta) / ta),
             "%7s %10s %10s %10s %10s %8s  %s" % ("calls", "avg_" + la, "avg_" + lb, "min_" + la, "min_" + lb, "d_total", "kernel")]
    names = sorted(set(sa) | set(sb), key=lambda n: -(sa.get(n, (0, 0))[1] + sb.get(n, (0, 0))[1]))
    for n in names:
        x, y = sa.get(n), sb.get(n)
        if x is None or y is None:
            v = x or y
            lines.append("%7d %10s %10s %10s %10s %8s  %s  [only in %s: avg %.2f]" % (v[0], "-", "-", "-", "-", "-", n[:120], la if x else lb, v[2]))
            continue
        d = y[1] - x[1]
        flag = " *" if abs(d) > 0.005 * ta else ""
        lines.append("%7d %10.2f %10.2f %10.2f %10.2f %+8.1f  %s%s" % (x[0], x[2], y[2], x[3], y[3], d, n[:120], flag))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:45]))


if __name__ == "__main__":
    main(*sys.argv[1:])
