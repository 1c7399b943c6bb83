"""Per-kernel summary (calls, total, average, share) from a rocprofv3 rocpd sqlite database
(``rocprofv3 --kernel-trace --stats`` writes <name>_results.db).  Usage:
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [steps] > profiles/x_kernel_stats.csv
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else cols[0]
    rows = c.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     f"from kernels group by {name_col} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,MsPerStep")
    for n, calls, t, a, mn, mx in rows:
        print(f"\"{n}\",{calls},{t},{a:.1f},{100.0 * t / tot:.2f},{mn},{mx},{t / 1e6 / steps:.4f}")
    print(f"\"TOTAL\",,{tot},,100.0,,,{tot / 1e6 / steps:.4f}")


if __name__ == "__main__":
    main()
