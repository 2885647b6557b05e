#!/usr/bin/env python3
"""Timeline of the LAST RK4 step in a rocprofv3 kernel trace of tools/nl_rank_timing.py: one line per kernel with its queue, start and
end (us, relative to the step's first kernel), so that the exchange kernels (k_halo_map on the communication queue) can be seen
beside the interior launches of the compute queue.   tools/nl_timeline.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, queue_id, stream_id, start, end from kernels order by start"))
# the last step = from the 9th-last preparation launch of the interior patches on: 4 stages x (prep BH x2, prep I, stage B, stage I)
short = lambda n: n.split("(")[0].replace("void moka::", "").replace("moka::", "")
idx = [i for i, r in enumerate(rows) if "k_nl_prep" in r[0]]
first = idx[-12] if len(idx) >= 12 else 0          # 12 preparation launches per step: 4 x (interior, boundary, halo)
t0 = rows[first][3]
queues = sorted({r[1] for r in rows[first:]})
print(f"{'kernel':44s} {'queue':>5s} {'start_us':>10s} {'end_us':>10s} {'dur_us':>8s}")
for n, q, s_, a, e in rows[first:]:
    print(f"{short(n)[:44]:44s} {queues.index(q):5d} {(a - t0) / 1e3:10.1f} {(e - t0) / 1e3:10.1f} {(e - a) / 1e3:8.1f}")
