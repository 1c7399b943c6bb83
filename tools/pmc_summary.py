"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per kernel (gfx950 corrections of
MI355X_MICROARCH.md section HBM: values are KB; FETCH_SIZE of wide coalesced reads is doubled)."""
import csv, glob, sys, collections, json

def load(pattern, counter):
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    files = set(glob.glob(pattern) + glob.glob(pattern.replace("/*/*", "/*")) + glob.glob(pattern.split("/*")[0] + "/**/*counter_collection.csv", recursive=True))
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"][:120]
            a = agg[name]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return agg

fetch = load(sys.argv[1] + "/*/*counter_collection.csv", "FETCH_SIZE")
write = load(sys.argv[2] + "/*/*counter_collection.csv", "WRITE_SIZE")
rows = []
for k in fetch:
    n, kb, us = fetch[k]
    wn, wkb, wus = write.get(k, [0, 0.0, 0.0])
    rd = 2 * kb * 1024 / max(n, 1)            # gfx950: FETCH_SIZE reads 1/2 of wide coalesced streams
    wr = wkb * 1024 / max(wn, 1)
    rows.append((us, k, n, rd, wr, us / max(n, 1)))
rows.sort(reverse=True)
out = []
for us, k, n, rd, wr, avg in rows[:14]:
    gbs = (rd + wr) / (avg * 1e-6) / 1e9 if avg > 0 else 0
    print(f"{k[:70]:70s} x{n:4d} avg {avg:9.1f} us  read {rd/1e6:9.2f} MB  write {wr/1e6:9.2f} MB  -> {gbs:8.1f} GB/s")
    out.append({"kernel": k, "launches": n, "avg_us": round(avg, 1), "hbm_read_MB_per_launch": round(rd / 1e6, 3),
                "hbm_write_MB_per_launch": round(wr / 1e6, 3), "GBps": round(gbs, 1)})
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    # stamp: the source tree these counters were taken on (bench.py drops roofline.traffic when it differs)
    import importlib.util, os
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("sfod_csrc_build", os.path.join(here, "simple-sfod_amd", "csrc", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    json.dump({"csrc_fingerprint": mod.source_fingerprint()}, open(sys.argv[3] + ".meta", "w"))
