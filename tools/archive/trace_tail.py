#!/usr/bin/env python3
"""Print the last N kernel records (start/end relative, us) of a rocprofv3 rocpd .db -- to see overlap across streams."""
import sqlite3, sys, glob, os, re
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
db = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0]
cur = sqlite3.connect(db).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
extra = [c for c in ("stream_id", "queue_id") if c in cols]
rows = cur.execute(f"select name, start, end{''.join(', ' + c for c in extra)} from kernels order by start").fetchall()[-n:]
t0 = rows[0][1]
for r in rows:
    nm = re.sub(r"\(.*", "", r[0]).replace("void ", "").replace("kagnn::", "")[:48]
    print(f"{nm:48s} {(r[1]-t0)/1e3:9.1f} {(r[2]-t0)/1e3:9.1f}  dur {(r[2]-r[1])/1e3:8.1f}", *r[3:])
