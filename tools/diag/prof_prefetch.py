"""Where a fresh batch spends its host time on the way to the GPU (--config real shape): collate, CSR upload, cluster
schedule concatenation + upload, feature upload; plain copies against the page-locked ring; in line against the
prefetch thread.  Usage: python tools/diag/prof_prefetch.py"""
import contextlib
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gnn-tumor-seg_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from data_processing.data_loader import minibatch_graphs  # noqa: E402
from gts.graph import PinnedRing, uploads_through  # noqa: E402
from model.gnn_model import GNN  # noqa: E402

cfg = bench.CONFIGS["real"]
ds = bench.RealDataset(48, 20)
with contextlib.redirect_stdout(io.StringIO()):
    model = GNN("GSpool", bench.hyperparams(cfg), ds, batch_size=6)
model.net.train()
samples = [ds[i] for i in range(6)]


def t(label, fn, n=20, sync=True):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    if sync:
        torch.cuda.synchronize()
    print(f"{label:44s} {(time.perf_counter() - t0) / n * 1e3:8.3f} ms", flush=True)


for ring in (None, PinnedRing()):
    uploads_through(ring)
    tag = "ring " if ring is not None else "plain"
    t(f"[{tag}] minibatch_graphs", lambda: minibatch_graphs(samples))
    t(f"[{tag}] + graph.to().dev()", lambda: minibatch_graphs(samples)[1].to("cuda").dev())
    t(f"[{tag}] + dev() + host schedule concat", lambda: (lambda g: (g.dev(), g.cluster_schedule("out")))(minibatch_graphs(samples)[1].to("cuda")))
    t(f"[{tag}] + dev() + dev_schedule(out)", lambda: (lambda g: (g.dev(), g.dev_schedule("out")))(minibatch_graphs(samples)[1].to("cuda")))
    item = minibatch_graphs(samples)
    t(f"[{tag}] features + labels upload", lambda: model._to_device(item[1], item[2], item[3]))
    if ring is not None:
        t(f"[{tag}] next_batch()", ring.next_batch)
uploads_through(None)
b = model._to_device(*minibatch_graphs(samples)[1:])
t("train_step resident", lambda: model.train_step(*b))


def epoch():
    for batch in model._device_batches():
        model.train_step(*batch)


t("epoch of 8 steps (prefetch thread)", epoch, n=5)
model.prefetch = False
t("epoch of 8 steps (in line)", epoch, n=5)
