"""Per-kernel SQ / GRBM counter summary from rocprofv3 --pmc passes (one directory per pass, each holding
*counter_collection.csv).  Usage: python tools/pmc_sq_summary.py out.json dir1 [dir2 ...]

Derived columns (gfx950, MI355X_MICROARCH.md "rocprofv3 PMC slots" / "Per-instruction cycle constants"):
  clock_GHz        GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel time
  mfma_busy        SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs): share of SIMD-cycles the matrix
                   pipe is busy, at the clock the kernel actually ran at
  wave_parked / issue_stalled / active   SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES
"""
import collections
import csv
import glob
import json
import sys


def main():
    out_path, dirs = sys.argv[1], sys.argv[2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0, 0.0]))   # kernel -> counter -> [n, sum, us]
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                a = agg[r["Kernel_Name"]][r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
                a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    rows = []
    for k, cs in agg.items():
        any_c = next(iter(cs.values()))
        n, us = any_c[0], any_c[2]
        row = {"kernel": k[:140], "launches": n, "avg_us_under_pmc": round(us / max(n, 1), 1), "total_us": us}
        per = {c: v[1] / max(v[0], 1) for c, v in cs.items()}
        row["counters_per_launch"] = {c: round(v, 1) for c, v in sorted(per.items())}
        g = per.get("GRBM_GUI_ACTIVE")
        avg_us = {c: v[2] / max(v[0], 1) for c, v in cs.items()}
        if g and avg_us.get("GRBM_GUI_ACTIVE"):
            row["clock_GHz"] = round(g / 8.0 / (avg_us["GRBM_GUI_ACTIVE"] * 1e3), 3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in per:
                row["mfma_busy"] = round(per["SQ_VALU_MFMA_BUSY_CYCLES"] / (g / 8.0 * 256 * 4), 4)
        wc = per.get("SQ_WAVE_CYCLES")
        if wc:
            for name, c in (("wave_parked", "SQ_WAIT_ANY"), ("issue_stalled", "SQ_WAIT_INST_ANY"),
                            ("active", "SQ_ACTIVE_INST_ANY"), ("lds_issue_stalled", "SQ_WAIT_INST_LDS")):
                if c in per:
                    row[name] = round(per[c] / wc, 4)
        if per.get("SQ_LDS_IDX_ACTIVE"):
            row["lds_bank_conflict_share"] = round(per.get("SQ_LDS_BANK_CONFLICT", 0.0) / per["SQ_LDS_IDX_ACTIVE"], 4)
        rows.append(row)
    rows.sort(key=lambda r: -r["total_us"])
    for r in rows:
        r.pop("total_us")
    json.dump(rows[:16], open(out_path, "w"), indent=1)
    for r in rows[:8]:
        print({k: v for k, v in r.items() if k != "counters_per_launch"})


if __name__ == "__main__":
    main()
