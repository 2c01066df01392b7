"""Would two half batches on two streams beat one batch on one stream?  Two independent models train on 2 graphs each,
concurrently on two HIP streams (one's memory-bound K1 / K2 and output drains beside the other's GEMMs), against one
model on 4 graphs.  Throughput only (the two-model run is not the same optimisation problem).
Usage: python tools/diag/two_streams.py"""
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
from model.gnn_model import GNN  # noqa: E402

cfg = bench.CONFIGS["c2"]
dev = torch.device("cuda", 0)


def model():
    with contextlib.redirect_stdout(io.StringIO()):
        m = GNN(cfg["model"], bench.hyperparams(cfg), None)
    m.net.train()
    return m


def run(label, models, batches, streams, steps=30):
    def block():
        for i in range(steps):
            for m, b, s in zip(models, batches, streams):
                with torch.cuda.stream(s):
                    m.train_step(*b[i % len(b)])
    block()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    block()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    graphs = sum(b[0][0].batch_size for b in batches)
    print(f"{label:40s} {1e3 * dt:7.3f} ms/step  {graphs / dt:8.1f} graphs/s", flush=True)


one = model()
run("one stream, 4 graphs", [one], [bench.build_batches(0, 4, "lattice", 2, dev)], [torch.cuda.current_stream()])
a, b = model(), model()
ba, bb = bench.build_batches(0, 2, "lattice", 2, dev), bench.build_batches(1, 2, "lattice", 2, dev)
run("one stream, 2 + 2 graphs in turn", [a, b], [ba, bb], [torch.cuda.current_stream()] * 2)
run("two streams, 2 + 2 graphs", [a, b], [ba, bb], [torch.cuda.Stream(), torch.cuda.Stream()])
c, d = model(), model()
bc, bd = bench.build_batches(2, 4, "lattice", 2, dev), bench.build_batches(3, 4, "lattice", 2, dev)
run("two streams, 4 + 4 graphs", [c, d], [bc, bd], [torch.cuda.Stream(), torch.cuda.Stream()])
