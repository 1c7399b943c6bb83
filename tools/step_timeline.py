"""One steady-state step of a rocprofv3 kernel trace as a per-stream timeline: which kernels ran on which stream, when, and
where a stream (or the whole GPU) sat idle.  The step boundary is the fused optimiser kernel (k_sgd_ema).
    python tools/step_timeline.py <rocpd .db> [step index from the end, default 3] [min gap us to print, default 15]"""
import sqlite3
import sys

db = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mingap = float(sys.argv[3]) if len(sys.argv) > 3 else 15.0
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
stream_col = next((x for x in ("stream_id", "stream", "queue_id", "queue") if x in cols), None)
rows = c.execute(f"select name, start, end, {stream_col or '0'} from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if "k_sgd_ema" in r[0]]
assert len(marks) > back + 1, (len(marks), cols)
lo, hi = marks[-back - 2] + 1, marks[-back - 1] + 1          # kernels after one update up to and including the next
step = rows[lo:hi]
t0 = step[0][1]
span = step[-1][2] - t0
streams = {}
for n, s, e, q in step:
    streams.setdefault(q, []).append((s - t0, e - t0, n))
print(f"columns: {cols}")
print(f"step of {len(step)} kernels, {span / 1e3:.1f} us, streams: " + ", ".join(f"{q}: {len(v)} kernels, busy {sum(e - s for s, e, _ in v) / 1e3:.0f} us" for q, v in streams.items()))
# union busy
ivs = sorted((s, e) for s, e, _, _ in [(a - t0, b - t0, n, q) for n, a, b, q in step])
busy, cs, ce, gaps = 0, ivs[0][0], ivs[0][1], []
for s, e in ivs[1:]:
    if s > ce:
        busy += ce - cs
        gaps.append((s - ce, ce))
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print(f"GPU busy {100 * busy / span:.1f} % of the step; idle {sum(g for g, _ in gaps) / 1e3:.0f} us in {len(gaps)} gaps")


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:60]


print("\ntimeline (start us | dur us | stream | kernel), '---' = the whole GPU idle for >= %.0f us:" % mingap)
allk = sorted((s - t0, e - t0, q, n) for n, s, e, q in step)
gapset = {round(at): g for g, at in gaps if g / 1e3 >= mingap}
last_end = 0
for s, e, q, n in allk:
    for at, g in list(gapset.items()):
        if at <= s and at >= last_end - 1:
            print(f"   --- GPU idle {g / 1e3:7.1f} us at {at / 1e3:9.1f}")
            del gapset[at]
    print(f"{s / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {str(q):>4}  {short(n)}")
    last_end = max(last_end, e)
