# One-launch BatchNorm finalize (SFOD_BN_FINALIZE_FUSED, default 1) against the two launches (0): the configurations whose layers have
# <= 64 statistics blocks -- one frame per GPU on both networks -- and the headline (where only a few layers qualify), alternating.
#   bash tools/experiments/bn_finalize_ab.sh <out.txt>
export TMPDIR=/tmp
OUT=$1; D=gpurun_out/bnf_ab; mkdir -p $D; : > $OUT
for cfg in "--model r101 --batch 1 --steps 150" "--batch 1 --steps 200" "--model r101 --steps 30" "--steps 40"; do
  for rep in 1 2; do
    for f in 1 0; do
      SFOD_BN_FINALIZE_FUSED=$f python3 bench.py --no-cpu-baseline --no-secondary --no-kernel-timer --no-smi $cfg > $D/x.json 2> $D/x.err
      python3 - "$cfg" $f $D/x.json >> $OUT <<'PY'
import json, sys
d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print(f"{sys.argv[1]:34s} fused={sys.argv[2]} {d['value']:8.2f} images/s  {d['ms_per_step']:8.3f} ms/step")
PY
      sleep 3
    done
  done
done
