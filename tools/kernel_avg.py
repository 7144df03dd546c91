"""Average duration per kernel from a rocprofv3 results .db (kernel-trace): python tools/kernel_avg.py <db> [substring]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for name, calls, total, avg, pct in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    if pat in name:
        print(f"{avg:10.3f} us x {calls:6d}  {pct:5.1f} %  {name[:110]}")
