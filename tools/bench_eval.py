"""Evaluation-path throughput: eval-mode inference images/s on the synthetic evaluation set
(1024x2048 frames -> ResizeShortestEdge(600) on the device -> backbone -> RPN (TEST top-k 6000 / 1000) -> box head
-> per-class NMS -> detector_postprocess), plus the host-side AP table time.  One JSON line."""
import argparse
import importlib
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--images", type=int, default=64)
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    args = ap.parse_args()
    sfod = importlib.import_module("simple-sfod_amd")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = sfod.config.setup_cfg(
        os.path.join(root, "configs", "faster_rcnn_VGG_cityscapes_foggy_adaptive_teacher_source_free.yaml"),
        ["OUTPUT_DIR", "", "SFOD.COMPUTE_DTYPE", args.dtype, "TEST.IMS_PER_BATCH", str(args.batch), "SFOD.SYNTHETIC.NUM_TEST_IMAGES", str(args.images),
         "DATASETS.TEST", "('synthetic_cityscapes_foggy_val',)"])
    torch.manual_seed(0)
    T = sfod.engine.BaseTrainer
    model = T.build_model(cfg)
    with torch.no_grad():        # planted scores so that the per-class NMS and the evaluator see detections
        model.roi_heads.box_predictor.cls_score.weight.mul_(60.0)
    loader = T.build_test_loader(cfg, "synthetic_cityscapes_foggy_val")
    evaluator = T.build_evaluator(cfg, "synthetic_cityscapes_foggy_val", data_loader=loader)
    model.eval()
    with torch.no_grad():
        for batch in loader:     # warm-up pass
            model(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        for _ in range(args.passes):
            evaluator.reset()
            for batch in loader:
                outs = model(batch)
                evaluator.process(batch, outs)
                n += len(batch)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
    res = evaluator.evaluate()
    t2 = time.perf_counter()
    print(json.dumps({"metric": "eval-mode inference images/s (incl. postprocess + evaluator.process)",
                      "value": round(n / (t1 - t0), 2), "unit": "images/s", "n_gpus": 1, "batch": args.batch,
                      "images": n, "ap_table_seconds": round(t2 - t1, 3),
                      "detections": len([d for p in evaluator._predictions for d in p["instances"]]),
                      "AP50": res["bbox"]["AP50"], "dtype": cfg.SFOD.COMPUTE_DTYPE, "data": "synthetic"}))


if __name__ == "__main__":
    main()
