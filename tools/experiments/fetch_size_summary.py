"""FETCH_SIZE (KB, as rocprofv3 reports it on gfx950) / true bytes per launch of tools/experiments/fetch_size_calibration.hip."""
import collections, csv, glob, sys
BYTES = 768 << 20
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            agg[r["Kernel_Name"]].append((float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
print(f"{'kernel':44s} launches  FETCH_SIZE*1024/bytes   avg us   true GB/s")
for k in sorted(agg):
    v = agg[k]
    ratio = sum(a for a, _ in v) / len(v) * 1024 / BYTES
    us = sum(b for _, b in v) / len(v)
    print(f"{k[:44]:44s} {len(v):5d}     {ratio:8.4f}           {us:8.1f}  {BYTES / us / 1e3:8.1f}")
