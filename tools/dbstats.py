#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 results database (rocpd format): tools/dbstats.py <results.db> [pattern]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
rows = db.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels group by name order by sum(end-start) desc")
print(f"{'kernel':100s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'total_ms':>9s}")
for n, c, a, lo, hi, tot in rows:
    if pat in n:
        print(f"{n[:100]:100s} {c:6d} {a / 1e3:9.1f} {lo / 1e3:9.1f} {hi / 1e3:9.1f} {tot / 1e6:9.2f}")
