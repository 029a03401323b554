"""dev helper: timeline of the LAST burst of kernels in a rocprofv3 rocpd database (one compute step): start offset,
duration, stream / queue, gap to the previous kernel's end on the busiest stream.
usage: python devtools/timeline.py results.db [gap_ms_between_steps=5]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
print("# columns:", cols)
pick = lambda *names: next((c for c in names if c in cols), None)
c_start, c_end, c_name = pick("start", "start_timestamp"), pick("end", "end_timestamp"), pick("name", "kernel_name")
c_stream = pick("stream_id", "stream", "queue_id", "queue")
rows = db.execute(f"select {c_name}, {c_start}, {c_end}, {c_stream or 0} from kernels order by {c_start}").fetchall()
gap = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 5e6
bursts, cur = [], [rows[0]]
for r in rows[1:]:
    if r[1] - max(x[2] for x in cur) > gap:
        bursts.append(cur); cur = []
    cur.append(r)
bursts.append(cur)
b = bursts[-1]
t0 = b[0][1]
short = lambda n: re.sub(r"^void |lsgpu::", "", re.sub(r"\(.*", "", n))[:46]
print("# last burst: %d kernels, %.3f ms from first start to last end; kernel time summed %.3f ms" % (len(b), (max(x[2] for x in b) - t0) / 1e6, sum(x[2] - x[1] for x in b) / 1e6))
last_end = {}
busy_until = t0
idle = 0.0
for n, s, e, st in b:
    g_all = (s - busy_until) / 1e3
    if g_all > 0: idle += g_all
    print("%9.1f us  +%7.1f  %-46s stream %-6s gap_any %6.1f  gap_own %6.1f" % ((s - t0) / 1e3, (e - s) / 1e3, short(n), st, g_all, (s - last_end.get(st, s)) / 1e3))
    last_end[st] = e
    busy_until = max(busy_until, e)
print("# device idle inside the burst (no kernel running on any stream): %.1f us" % idle)
