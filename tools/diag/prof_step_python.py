"""cProfile of the host side of GNN.train_step on a resident batch (--config real shape): where the enqueue time goes.
Usage: python tools/diag/prof_step_python.py"""
import contextlib
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gnn-tumor-seg_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from model.gnn_model import GNN  # noqa: E402

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "real"]
with contextlib.redirect_stdout(io.StringIO()):
    model = GNN(cfg["model"], bench.hyperparams(cfg), None)
model.net.train()
b = bench.build_batches(0, cfg.get("graphs_per_gpu", 4), "lattice", 1, torch.device("cuda", 0), cfg=cfg)[0]
for _ in range(20):
    model.train_step(*b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    model.train_step(*b)
enq = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"enqueue {enq / 200 * 1e3:.3f} ms/step, wall {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    model.train_step(*b)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
