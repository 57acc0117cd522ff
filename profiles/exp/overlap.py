"""Do the merged launches of the front end's two lanes overlap on the GPU?  Reads a rocprofv3 --kernel-trace rocpd
database: the blind-rotation dispatches with their start / end stamps and queues.
    python profiles/exp/overlap.py <trace.db>"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print("columns:", cols)
q = "select name, start, end, queue_id, stream_id, grid_x, workgroup_x from kernels where name like '%blind_rotate%' order by start"
try:
    rows = list(cur.execute(q))
except sqlite3.OperationalError:
    q = q.replace("queue_id, stream_id", "queue_id, 0")
    rows = list(cur.execute(q))
print(len(rows), "blind-rotation dispatches")
busy = 0
overlap = 0
last_end = 0
queues = {}
for name, s, e, qid, sid, gx, wx in rows:
    queues[(qid, sid)] = queues.get((qid, sid), 0) + 1
    if s < last_end:
        overlap += min(e, last_end) - s
    busy += e - s
    last_end = max(last_end, e)
span = rows[-1][2] - rows[0][1] if rows else 0
print("queues (queue, stream) -> dispatches:", queues)
print(f"sum of durations {busy / 1e6:.1f} ms, span {span / 1e6:.1f} ms, overlapped {overlap / 1e6:.1f} ms")
for r in rows[200:216]:
    print(r[3], r[4], (r[1] - rows[0][1]) / 1e6, (r[2] - r[1]) / 1e6, r[5] // max(1, r[6]))
