#!/usr/bin/env python3
"""Launch-time / bandwidth table of the streaming kernels around the network (K12 variants,
K15, K16, K17) at the C5 volume size, HIP events over repeated launches on torch's current
stream.  Prints one JSON line per kernel: algorithmic bytes, mean launch time, GB/s and the
fraction of the 8 TB/s HBM peak.   python tools/measure_aux_kernels.py [--reps 30]"""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "gnn-tumor-seg_amd")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from gts import ops, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    pairs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        pairs.append((a, b))
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in pairs])) * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    shape = (240, 240, 240)
    n_vox = int(np.prod(shape))
    rng = np.random.default_rng(0)
    svs = torch.from_numpy(synth.supervoxel_volume(shape, cube=10, shell=20)).to(dev)
    node_logits = rng.standard_normal((15000, 4)).astype(np.float32)
    node_logits[:, 0] += np.where(rng.random(15000) < 0.05, -4.0, 4.0)        # ~5 % tumour nodes
    logits = torch.from_numpy(node_logits).to(dev)
    bg = torch.tensor([1.0, -1.0, -1.0, -1.0], device=dev)
    truth = torch.from_numpy(rng.choice(4, size=shape, p=[0.85, 0.07, 0.05, 0.03]).astype(np.int16)).to(dev)
    img = torch.from_numpy(rng.standard_normal(shape + (4,)).astype(np.float32)).to(dev)
    pred = ops.project_argmax(svs, logits)
    box = ops.CropBox(np.arange(40, 200), np.arange(50, 210), np.arange(30, 190), shape, dev)   # 160^3 crop
    n_crop = int(np.prod(box.shape))
    scores = torch.from_numpy(rng.standard_normal((4,) + box.shape).astype(np.float32)).to(dev)
    cases = [
        ("K12 project_rows (fp32x4 rows)", 18 * n_vox, lambda: ops.project_rows(svs, logits, bg)),
        ("K12 project_argmax (int16 labels)", 4 * n_vox, lambda: ops.project_argmax(svs, logits)),
        ("K12 project_argmax_occupancy", 4 * n_vox, lambda: ops.project_argmax_occupancy(svs, logits)),
        ("K15 label_confusion", 4 * n_vox, lambda: ops.label_confusion(pred, truth)),
        ("K16 crop_concat (4+4 channels, 160^3 box)", 50 * n_crop, lambda: ops.crop_concat(img, svs, logits, bg, box)),
        ("K17 argmax_scatter (4 classes, 160^3 box)", 18 * n_crop, lambda: ops.argmax_scatter(scores, box)),
    ]
    for name, nbytes, fn in cases:
        sec = timed(fn, args.reps)
        gbs = nbytes / sec / 1e9
        print(json.dumps({"kernel": name, "algorithmic_bytes": nbytes, "avg_launch_us": round(sec * 1e6, 2),
                          "achieved_GBps": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4),
                          "note": "event-bracketed op incl. its output allocation/zeroing"}), flush=True)


if __name__ == "__main__":
    main()
