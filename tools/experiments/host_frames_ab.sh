# PCIe-inclusive rate: frames in pinned host memory, uploaded every step on the loader's stream
# (bench.py --host-frames = SFOD.SYNTHETIC.HOST_FRAMES) against the headline's HBM-resident frames, alternating on one box.
#   bash tools/experiments/host_frames_ab.sh <out.txt>
export TMPDIR=/tmp
OUT=$1; D=gpurun_out/host_frames_ab; mkdir -p $D
: > $OUT
for cfg in "--steps 40" "--batch 1 --steps 150" "--model r101 --steps 30"; do
  for rep in 1 2; do
    for hf in "" "--host-frames"; do
      python3 bench.py --no-cpu-baseline --no-secondary --no-kernel-timer --no-smi $cfg $hf > $D/x.json 2> $D/x.err
      python3 - "$cfg" "${hf:-hbm-resident}" $D/x.json >> $OUT <<'PY'
import json, sys
d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print(f"{sys.argv[1]:28s} {sys.argv[2]:14s} {d['value']:8.2f} images/s  {d['ms_per_step']:8.3f} ms/step")
PY
      sleep 3
    done
  done
done
