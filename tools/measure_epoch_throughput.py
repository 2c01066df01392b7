#!/usr/bin/env python3
"""End-to-end epoch throughput of GNN.run_epoch on in-memory samples (collate + upload + step),
with and without the batch-prefetch thread, next to bench.py's resident-batch figure.
  python tools/measure_epoch_throughput.py [--config c2|real] [--samples 48] [--batch N] [--graph-kind lattice]
--config real: the reference's own workload (in_feats 20, layer_sizes [256]*4, 6 graphs of ~6k nodes per step)."""
import argparse
import contextlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "gnn-tumor-seg_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402

import bench  # noqa: E402
from gts import synth  # noqa: E402
from model.gnn_model import GNN  # noqa: E402


class MemDataset(torch.utils.data.Dataset):
    def __init__(self, n, kind, cfg):
        if "in_feats" in cfg:      # the reference's real shape
            self.items = [bench.real_sample(i, cfg["in_feats"]) for i in range(n)]
            return
        self.items = [synth.make_sample(i, kind=kind, in_feats=bench.IN_FEATS) for i in range(n)]
        self.items = [(f"s{i}", g, f.astype("float64"), y) for i, (_, g, f, y) in enumerate(self.items)]

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=None, help="default: 48 steps per epoch")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--graph-kind", default="lattice")
    ap.add_argument("--config", default="c2", choices=["c2", "real"])
    ap.add_argument("--epochs", type=int, default=3, help="timed epochs per mode (after one warm-up epoch)")
    args = ap.parse_args()
    cfg = bench.CONFIGS[args.config]
    args.batch = args.batch or cfg.get("graphs_per_gpu", 4)
    args.samples = args.samples or 48 * args.batch
    if os.environ.get("GTS_SWITCH_INTERVAL"):       # experiment: how long a thread may keep the interpreter lock
        sys.setswitchinterval(float(os.environ["GTS_SWITCH_INTERVAL"]))
    data = MemDataset(args.samples, args.graph_kind, cfg)
    modes = [(False, False), (False, True), (True, False), (True, True)] * 2     # (prefetch thread, C host collate)
    for prefetch, host_collate in modes:
        torch.manual_seed(0)
        with contextlib.redirect_stdout(sys.stderr):
            model = GNN(cfg["model"], bench.hyperparams(cfg), data, batch_size=args.batch, prefetch=prefetch,
                        host_collate=host_collate)
        model.run_epoch()                                   # warm-up epoch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.epochs):
            loss = model.run_epoch()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        steps = len(model.train_loader) * args.epochs
        print(json.dumps({"config": args.config, "batch": args.batch, "prefetch": prefetch, "host_collate": host_collate,
                          "graphs_per_s": round(steps * args.batch / dt, 1),
                          "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps, "epoch_loss": float(loss)}),
              flush=True)


if __name__ == "__main__":
    main()
