# What runs outside libsfod_hip.so in a one-frame-per-GPU step?  rocprofv3 kernel trace of bench.py --batch 1 (single stream), then
# the kernels that are NOT the library's, per step.
#   bash tools/experiments/b1_small_ops.sh <out.txt>
export TMPDIR=/tmp
OUT=$1; shift; EXTRA="$@"; D=gpurun_out/b1_trace; rm -rf $D; mkdir -p $D
B=$(python3 bench.py $EXTRA --batch 1 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-timer 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['config']['planted_labels']['background_bias'])")
rocprofv3 --kernel-trace --stats -d $D -o kt -- python3 bench.py $EXTRA --batch 1 --plant-bias $B --no-overlap --no-cpu-baseline --no-secondary --no-kernel-timer --no-smi --steps 40 --warmup 10 > /dev/null 2> $D/err.txt
DB=$(find $D -name "*.db" | head -1)
python3 tools/rocpd_stats.py $DB 50 > $D/stats.csv
python3 - $D/stats.csv > $OUT <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Name"] != "TOTAL"]
tot = sum(float(r["MsPerStep"]) for r in rows)
lib = lambda n: n.startswith(("k_", "void k_", "(anonymous namespace)::k_", "void (anonymous namespace)::k_", "_Z")) 
oth = [r for r in rows if not lib(r["Name"])]
print(f"step (sum of kernels, single stream, profiled): {tot:.3f} ms; outside the library: {sum(float(r['MsPerStep']) for r in oth):.3f} ms in {sum(int(r['Calls']) for r in oth) / 50:.1f} launches per step")
for r in sorted(oth, key=lambda r: -float(r["MsPerStep"]))[:40]:
    print(f"{float(r['MsPerStep']):7.4f} ms  {int(r['Calls']) / 50:6.1f}/step  {float(r['AverageNs']) / 1e3:7.1f} us  {r['Name'][:150]}")
print("--- library kernels under 10 us average (launch-bound) ---")
small = [r for r in rows if lib(r["Name"]) and float(r["AverageNs"]) < 10e3]
print(f"{sum(float(r['MsPerStep']) for r in small):.3f} ms in {sum(int(r['Calls']) for r in small) / 50:.1f} launches per step")
for r in sorted(small, key=lambda r: -float(r["MsPerStep"]))[:25]:
    print(f"{float(r['MsPerStep']):7.4f} ms  {int(r['Calls']) / 50:6.1f}/step  {float(r['AverageNs']) / 1e3:7.1f} us  {r['Name'][:150]}")
PY
rm -rf $D
