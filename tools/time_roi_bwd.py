"""Time sfod_roi_align_bwd on the student's shapes (SFOD_ROI_BWD_ATOMIC=1 selects the scatter form)."""
import importlib
import sys
import torch

sys.path.insert(0, ".")
nat = importlib.import_module("simple-sfod_amd.native")


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    for (B, H, W, C, per, size) in [(8, 37, 75, 512, 512, 160.0), (8, 37, 75, 512, 512, 400.0), (8, 64, 128, 512, 512, 200.0),
                                    (8, 37, 75, 1024, 512, 160.0)]:
        R = B * per
        cx = torch.rand(R, generator=g) * W * 16
        cy = torch.rand(R, generator=g) * H * 16
        w = size * (0.3 + 1.4 * torch.rand(R, generator=g))
        h = size * (0.3 + 1.4 * torch.rand(R, generator=g))
        rois = torch.stack([torch.arange(R).div(per, rounding_mode="floor").float(),
                            (cx - w / 2).clamp(0, W * 16), (cy - h / 2).clamp(0, H * 16),
                            (cx + w / 2).clamp(0, W * 16), (cy + h / 2).clamp(0, H * 16)], 1).to(dev)
        dout = torch.randn(R, 49, C, generator=g).to(dev).to(torch.bfloat16)
        df = torch.zeros(B, H, W, C, device=dev)
        for _ in range(3):
            nat.roi_align_bwd(dout, rois, (B, H, W, C), 7, 1 / 16, dfeat=df)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            nat.roi_align_bwd(dout, rois, (B, H, W, C), 7, 1 / 16, dfeat=df)
        e1.record()
        torch.cuda.synchronize()
        print("B%d %dx%dx%d R%d size~%g: %.1f us" % (B, H, W, C, R, size, e0.elapsed_time(e1) * 50), flush=True)


if __name__ == "__main__":
    main()
