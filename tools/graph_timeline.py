"""Timeline of one forward from a rocprofv3 kernel-trace database: per kernel start / duration / queue, the gap to the previous
kernel end on the critical path, the time the chip runs 0 / 1 / >= 2 kernels.   python tools/graph_timeline.py results.db [n]"""
import re
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = con.execute(f"select d.start, d.end, s.kernel_name, d.queue_id, d.stream_id from {kd} d join {ks} s on d.kernel_id = s.id "
                   "order by d.start").fetchall()
idx = [i for i, r in enumerate(rows) if "prep_images" in r[2]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
a, b = idx[which], idx[which + 1]
seg = rows[a:b]
t0 = seg[0][0]
print("forward = %d kernels, span %.1f us (prep_images to next prep_images)" % (len(seg), (rows[b][0] - t0) / 1e3))
events = []
busy_end = t0
gap_total = 0.0
ksum = 0.0
for s, e, n, q, st in seg:
    n = re.sub(r"\(.*", "", n)
    n = re.sub(r"^void ", "", n)[:64]
    gap = (s - busy_end) / 1e3            # > 0: the chip was idle before this kernel started
    if gap > 0:
        gap_total += gap
    print("%9.1f  dur %7.1f  idle-before %6.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, max(gap, 0.0), q, n))
    busy_end = max(busy_end, e)
    ksum += (e - s) / 1e3
    events += [(s, 1), (e, -1)]
events.sort()
lvl, last, hist = 0, t0, {}
for t, d in events:
    hist[min(lvl, 2)] = hist.get(min(lvl, 2), 0.0) + (t - last) / 1e3
    lvl += d
    last = t
print("kernel time sum %.1f us; chip idle inside the forward %.1f us; time with 1 kernel %.1f us, with >= 2 kernels %.1f us" % (
    ksum, gap_total, hist.get(1, 0.0), hist.get(2, 0.0)))
