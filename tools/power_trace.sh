#!/bin/bash
# Package power and clocks beside a bench run (review item 5 of round 4: "measure package power rather than only time").
#   bash tools/power_trace.sh <out.txt> <bench.py args...>
# rocm-smi is sampled every ~0.2 s in a child process while bench.py runs in the foreground; the samples taken during the
# timed region (after the first "timed" stamp in bench's stderr would need parsing -- instead: the top quartile of the
# samples by power = the steady state of the step loop) are summarised.
export TMPDIR=/tmp
OUT=$1; shift
TMP=$(mktemp -d)
if [ ! -s "$OUT" ]; then
  echo "# rocm-smi --showmaxpower: $(rocm-smi --showmaxpower --json 2>/dev/null | tr -d '\n')" >> $OUT
  echo "# first raw sample: $(rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d '\n' | cut -c1-600)" >> $OUT
fi
( while true; do
    rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d '\n'; echo
    sleep 0.2
  done ) > $TMP/smi.jsonl &
SMI=$!
python3 bench.py "$@" --no-cpu-baseline --no-secondary > $TMP/bench.json 2> $TMP/bench.err
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
python3 - $TMP/smi.jsonl $TMP/bench.json "$*" >> $OUT <<'PY'
import json, sys, re
rows = []
for line in open(sys.argv[1]):
    line = line.strip()
    if not line.startswith("{"):
        continue
    try:
        d = json.loads(line)
    except Exception:
        continue
    card = next(iter(d.values()))
    p = next((float(v) for k, v in card.items() if "ower" in k and re.match(r"^[0-9.]+$", str(v))), None)
    sclk = next((v for k, v in card.items() if k.lower().startswith("sclk")), None)
    m = re.search(r"([0-9]+)\s*Mhz", str(sclk), re.I) if sclk else None
    if p is not None:
        rows.append((p, int(m.group(1)) if m else None))
b = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
rows.sort()
top = rows[len(rows) * 3 // 4:] if rows else []
avg = lambda xs: sum(xs) / max(len(xs), 1)
print(f"bench.py {sys.argv[3]}: {b['value']} images/s, {b['ms_per_step']} ms/step, dtype {b['dtype']}")
print(f"   rocm-smi samples {len(rows)}; idle-side quartile {avg([r[0] for r in rows[:max(1, len(rows)//4)]]):.0f} W; "
      f"busy quartile: power {avg([r[0] for r in top]):.0f} W (max {max([r[0] for r in rows], default=0):.0f} W), "
      f"sclk {avg([r[1] for r in top if r[1]]):.0f} MHz (min in it {min([r[1] for r in top if r[1]], default=0)} MHz)")
PY
rm -rf $TMP
