"""GPU busy fraction of the timed steps from a rocprofv3 rocpd database: union of kernel intervals / wall span.
  python tools/busy_fraction.py gpurun_out/prof/x_results.db [skip_fraction [upto_fraction]]
(the window [skip, upto] of the kernel span: e.g. 0.5 0.95 = steady timed steps of a run without measurement segments)"""
import sqlite3, sys
db = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
c = sqlite3.connect(db)
rows = c.execute("select start, end from kernels order by start").fetchall()
t0, t1 = rows[0][0], rows[-1][1]
upto = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
cut = t0 + skip * (t1 - t0)          # drop warm-up
end = t0 + upto * (t1 - t0)
rows = [(s, e) for s, e in rows if s >= cut and e <= end]
span = rows[-1][1] - rows[0][0]
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
gaps = []
for s, e in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
ksum = sum(e - s for s, e in rows)
print(f"span {span/1e6:.2f} ms, union busy {busy/1e6:.2f} ms ({100*busy/span:.1f} %), kernel-time sum {ksum/1e6:.2f} ms "
      f"(concurrency {ksum/busy:.2f}x), idle {100*(span-busy)/span:.1f} %")
gaps.sort(reverse=True)
print("largest gaps (us):", [round(g / 1e3, 1) for g, _ in gaps[:12]])
for lo, hi in ((0, 2e3), (2e3, 5e3), (5e3, 20e3), (20e3, 1e12)):
    sel = [g for g, _ in gaps if lo <= g < hi]
    print(f"gaps {lo/1e3:g}-{hi/1e3:g} us: {len(sel)}  idle in them {sum(sel)/1e6:.2f} ms ({100*sum(sel)/span:.2f} % of the span)")
print("gaps > 20 us:", sum(1 for g, _ in gaps if g > 20e3), " total idle in them (ms):", round(sum(g for g, _ in gaps if g > 20e3) / 1e6, 2))
