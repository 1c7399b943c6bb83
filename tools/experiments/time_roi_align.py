"""ROIAlign forward at the teacher's sizes (2000 boxes per frame, B = 8): VGG16 (512 ch, 19 x 38 map) and ResNet-101-C4
(1024 ch, 38 x 75 map), half-pair features.   python tools/experiments/time_roi_align.py"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sfod = importlib.import_module("simple-sfod_amd"); native = sfod.native; native.load()
g = torch.Generator().manual_seed(0)
for name, H, W, C, scale in (("vgg16", 19, 38, 512, 1 / 32), ("r101", 38, 75, 1024, 1 / 16)):
    B, R = 8, 16000
    feat = native.cast(torch.randn(B, H, W, C, generator=g).cuda(), native.SPLITH_DTYPE)
    xy = torch.rand(R, 2, generator=g) * torch.tensor([1200.0, 600.0])
    wh = torch.rand(R, 2, generator=g) ** 2 * 500 + 16
    rois = torch.cat([torch.arange(R).div(R // B, rounding_mode="floor").float().view(-1, 1), xy - wh / 2, xy + wh / 2], 1).cuda()
    ts = []
    for r in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = native.roi_align_fwd(feat, rois, 7, scale); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = sorted(ts)[len(ts) // 2]
    print(f"{name}: {R} boxes x 49 x {C} ch  {t:.3f} ms  ({4e-9 * R * 49 * C / t * 1e3:.2f} GB/s written)"
          f"  [SFOD_ROI_CBLK={os.environ.get('SFOD_ROI_CBLK', 'default')} SFOD_ROI_NT={os.environ.get('SFOD_ROI_NT', 'default')}]")
