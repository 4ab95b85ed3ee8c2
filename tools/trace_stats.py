"""Per-kernel table (calls, average, total) of a rocprofv3 --kernel-trace --stats output directory (rocpd sqlite)."""
import glob
import os
import sqlite3
import sys

hits = glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)
if not hits:
    sys.exit("no .db under " + sys.argv[1])
con = sqlite3.connect(hits[0])
for name, calls, total, avg, pct in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    print("%-60s calls %5d  avg %10.1f us  total %10.3f ms  %5.1f%%" % (name[:60], calls, avg / 1e3, total / 1e6, pct))
