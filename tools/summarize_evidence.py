"""One table of an evidence set (tools/collect_evidence.sh <tag> all): the rows of DESIGN.md section 6.
   python tools/summarize_evidence.py <tag> [dir=gpurun_out]"""
import json, os, sys

tag = sys.argv[1]
d = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"


def load(name):
    p = os.path.join(d, f"{tag}_{name}.json")
    if not os.path.exists(p):
        return None
    t = open(p).read().strip()
    return json.loads(t.splitlines()[-1]) if t else None


def row(label, j, extra=""):
    if j is None:
        return
    r = j.get("roofline") or {}
    print(f"| {label} | `{j['dtype']}` | **{j['value']:.1f}** | {j['ms_per_step']:.1f} | {r.get('achieved', '')} | {r.get('frac', '')}{extra} |")


b = load("bench_default")
if b:
    tr = (b["roofline"].get("traffic") or {})
    tel = ((b.get("gpu_telemetry") or {}).get("cards") or {}).get("card0", {})
    row("default (hot yaml, teacher+student, B = 8, 600×1200)", b,
        f" ({b['roofline'].get('frac_of_bf16_peak_algorithmic')} of 2500 per algorithmic FLOP); HBM per launch {tr.get('hbm_read_MB_per_launch')} MB read / "
        f"{tr.get('hbm_write_MB_per_launch')} MB written; {tel.get('power_W_mean')} W at {tel.get('sclk_MHz_mean')} MHz")
    for m in b.get("other_parity_modes", []):
        print(f"| … `other_parity_modes` | `{m['dtype']}` | {m.get('value')} | {m.get('ms_per_step')} | | |")
    rp = b.get("reduced_precision_mode")
    if isinstance(rp, dict):
        print(f"| … `reduced_precision_mode` (not a parity mode) | `{rp['dtype']}` | {rp.get('value')} | {rp.get('ms_per_step')} | | |")
    for o in b.get("other_shapes", []):
        t = (o.get("gpu_telemetry") or {}).get("card0", {})
        print(f"| … `other_shapes`: {o['shape']} | `{o.get('dtype')}` | {o.get('value')} | {o.get('ms_per_step')} | | reserved {o.get('peak_hbm_reserved_GB')} GB, "
              f"{o.get('pseudo_labels_per_image')} labels / image, {t.get('power_W_mean')} W at {t.get('sclk_MHz_mean')} MHz |")
    g, w = b.get("roofline_gemm") or {}, b.get("roofline_wgrad") or {}
    print(f"| … generic GEMM / weight-gradient families of the same line | | | | {g.get('achieved')} / {w.get('achieved')} | {g.get('frac')} / {w.get('frac')} |")
    c = b.get("cpu_baseline") or {}
    print(f"| `cpu_baseline` (oracle port, {c.get('cores')} threads, the GPU run's planted head) | fp32 | {c.get('value')} | | | |")
r = load("bench_modelr101steps40")
if r:
    w = r.get("roofline_wgrad") or {}
    row("`--model r101 --steps 40` (config #5, its parity mode)", r, f"; weight gradients {w.get('frac')}")
    for m in r.get("other_parity_modes", []):
        print(f"| … `other_parity_modes` | `{m['dtype']}` | {m.get('value')} | {m.get('ms_per_step')} | | |")
    for o in r.get("other_shapes", []):
        print(f"| … `other_shapes`: {o['shape']} | `{o.get('dtype')}` | {o.get('value')} | {o.get('ms_per_step')} | | |")
for name, label in (("bench_batch1steps200", "`--batch 1 --steps 200`"), ("bench_trainerbase", "`--trainer base` (config #2)"),
                    ("bench_trainerbaseresfullsteps40", "`--trainer base --res full`"), ("bench_resfullsteps40", "`--res full --steps 40`"),
                    ("bench_optsSFOD_ELIDE_DEAD_BRANCHESFalse", "`--opts SFOD.ELIDE_DEAD_BRANCHES False`"), ("bench_dtypef16x3", "`--dtype f16x3`")):
    row(label, load(name))
