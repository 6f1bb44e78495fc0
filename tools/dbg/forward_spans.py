import re, sqlite3, sys
con = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = con.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
idx = [i for i, r in enumerate(rows) if "prep_images" in r[2]]
for k in range(len(idx)-1):
    seg = rows[idx[k]:idx[k+1]]
    print(k, len(seg), "kernels span %.1f us, kernel sum %.1f" % ((rows[idx[k+1]][0]-seg[0][0])/1e3, sum(e-s for s,e,_ in seg)/1e3))
