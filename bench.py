#!/usr/bin/env python3
"""Headline benchmark: training throughput (supervoxel-graphs/sec) of 7xGraphSAGE-pool-256
on synthetic 15k-node supervoxel graphs, N MI355X GPUs (BASELINE.json metric, config C2).

  python bench.py --gpus N --steps 20 --warmup 5          (N > 1: starts its own torchrun child)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One "step" = forward + class-weighted cross-entropy + backward + (gradient all-reduce when
N > 1) + AdamW update on one batch of `--graphs-per-gpu` (default 4) 15 000-node lattice
graphs per GPU (N_b = 60 000 nodes, E_b = 345 400 edges), in_feats 4, fp32, layer_sizes
[256]*7 => 8 SAGEConv layers.  Inputs are resident in HBM before the timed region.  Weak
scaling: per-GPU work is fixed, global batch = graphs-per-gpu x N.

Timing: after W warm-up steps the block of EXACTLY K steps (barrier + synchronize on both
sides, MAX over ranks) is timed `--blocks` times (default 30: ~3 s of GPU time at the headline shape); `value` is the MEDIAN block's
rate and `blocks` carries min / median / max.  One further block of the same K steps runs
with HIP events around every launch of the kernels the roofline objects describe.

Rank 0 prints ONE JSON line with the throughput, plus
  "roofline":     K11, the MFMA fp32 GEMM kernel most of the step is spent in: flops of every
                  forward / input-gradient / weight-gradient launch / their summed HIP-event
                  durations, against the 157.3 TFLOP/s dense fp32 MFMA peak;
  "roofline_hbm": the HBM-bound kernels of the path (K1 spmm_max_fwd, K2 spmm_max_bwd at
                  F=256; gat_fwd; project_rows): COMPULSORY bytes (each input/output array
                  once) / mean launch duration against the 8 TB/s HBM peak, with the per-edge
                  `algorithmic_*` figure of SURVEY §8d and the PMC traffic of the committed
                  profile (tagged with its source file) beside it;
  "cpu_baseline": the CPU oracle (pure-PyTorch restatement of the DGL CPU path) timed on this
                  box's host cores on a bounded sample of the same workload (N=1 only);
  "ranks", "backend", "all_reduce": what the collective saw when N > 1.

Other BASELINE.json configurations (parity-test cases, not the headline) can be timed with
  --config c3   GAT 4 layers x 4 heads x 256 training on the same graphs   (graphs/s)
  --config c4   the C2 model at 8 graphs per rank: with --gpus 8 this is BASELINE.json's data-parallel
                configuration (global batch 64, RCCL gradient all-reduce); with --gpus 1 its per-GPU workload
  --config c5   batched no-grad forward + node->voxel logits projection to 240^3 (volumes/s)
  --config real the reference's own training workload (utils/hyperparam_helpers.py:36-39, model/gnn_model.py:12):
                in_feats 20, layer_sizes [256]*4, batches of 6 graphs of ~6k nodes (18^3 lattices), every step a
                FRESH batch collated on the host and uploaded through GNN's prefetch path (graphs/s, with the
                resident-batch rate and the host enqueue time beside it)
"""
import argparse
import contextlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, "gnn-tumor-seg_amd")
for _p in (REPO, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

# dmabuf IPC for RCCL / device-tensor sharing: must be in the environment before the HIP runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak BW (spec)
MFMA_F32_PEAK_TFLOPS = 157.3   # same guide: dense fp32 MFMA (v_mfma_f32_32x32x2_f32) peak
IN_FEATS = 4
N_CLASSES = 4
CLASS_WEIGHTS = [0.1, 1.0, 2.0, 2.0]
CONFIGS = {
    "c2": dict(model="GSpool", layer_sizes=[256] * 7, heads=None, residuals=None,
               metric="supervoxel-graphs/sec training, 7xSAGE-pool-256, 15k-node graphs",
               hbm_kernels=("spmm_max_fwd_f256", "spmm_max_bwd_f256")),
    "c4": dict(model="GSpool", layer_sizes=[256] * 7, heads=None, residuals=None, graphs_per_gpu=8,
               metric="supervoxel-graphs/sec training, 7xSAGE-pool-256, 15k-node graphs",
               hbm_kernels=("spmm_max_fwd_f256", "spmm_max_bwd_f256")),
    "real": dict(model="GSpool", layer_sizes=[256] * 4, heads=None, residuals=None, graphs_per_gpu=6, in_feats=20,
                 metric="supervoxel-graphs/sec training, the reference's own workload: 4xSAGE-pool-256, in_feats 20, "
                        "6 graphs of ~6k nodes per step, fresh batches",
                 hbm_kernels=("spmm_max_fwd_f256", "spmm_max_bwd_f256")),
    "c3": dict(model="GAT", layer_sizes=[256] * 4, heads=[4] * 4, residuals=[False] * 4,
               metric="supervoxel-graphs/sec training, GAT 4 layers x 4 heads x 256, 15k-node graphs",
               hbm_kernels=("gat_fwd", "gat_bwd_edge", "gat_bwd_src")),
    "c5": dict(model="GSpool", layer_sizes=[256] * 7, heads=None, residuals=None,
               metric="volumes/sec inference: 7xSAGE-pool-256 forward + node-logit->voxel projection to 240^3",
               hbm_kernels=("project_rows",)),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--blocks", type=int, default=30,
                    help="the --steps block is timed this many times; value = the median block")
    ap.add_argument("--graphs-per-gpu", type=int, default=None,
                    help="graphs in one rank's batch (default 4; 8 for --config c4, 6 for --config real)")
    ap.add_argument("--graph-kind", default="lattice", choices=["lattice", "random", "selfloop"])
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=5, help="timed CPU-oracle steps after one warm-up (SURVEY.md §8d: >= 5)")
    args = ap.parse_args()
    if args.graphs_per_gpu is None:
        args.graphs_per_gpu = CONFIGS[args.config].get("graphs_per_gpu", 4)
    return args


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


class KernelTimers:
    """HIP-event pairs around kernel launches on torch's current stream (the stream gts launches
    on), grouped by name; `work` = flops (or anything additive) the caller attaches to a launch."""

    def __init__(self):
        self.pairs = {}
        self.enabled = False

    def bracket(self, name, work=0.0):
        def run(launch):
            if not self.enabled:
                return launch()
            a = torch.cuda.Event(enable_timing=True)
            b = torch.cuda.Event(enable_timing=True)
            a.record()
            out = launch()
            b.record()
            self.pairs.setdefault(name, []).append((a, b, work))
            return out
        return run

    def summary(self, name):
        pairs = self.pairs.get(name)
        if not pairs:
            return None
        return {"launches": len(pairs), "total_ms": float(sum(a.elapsed_time(b) for a, b, _ in pairs)),
                "work": float(sum(w for _, _, w in pairs))}


REAL_DIMS = (18, 18, 18)     # 5 832 nodes: "less than half" of the 15k requested supervoxels survive (mri2graph/graphgen.py:210-211)


def real_sample(g_index, in_feats):
    """One sample of the reference's real shape: ~6k-node supervoxel graph (lattice), float64 features as its
    loader yields them (data_processing/data_loader.py:67-83)."""
    from gts import synth

    g = synth.lattice_graph(REAL_DIMS)
    seed = 1000 + g_index
    return (f"synth_real_{g_index:04d}", g, synth.node_features(g.n, in_feats, seed).astype(np.float64),
            synth.node_labels(g.n, seed))


class RealDataset(torch.utils.data.Dataset):
    """In-memory stand-in for ImageGraphDataset at the reference's real shape (same item layout)."""

    def __init__(self, n_samples, in_feats):
        self.items = [real_sample(i, in_feats) for i in range(n_samples)]
        self.read_label = True

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def build_batches(rank, graphs_per_gpu, kind, n_batches, device, cfg=None):
    """Distinct synthetic batches for this rank, uploaded once (graph g uses seed 1000+g)."""
    import gts
    from gts import synth

    in_feats = (cfg or {}).get("in_feats", IN_FEATS)
    batches = []
    for b in range(n_batches):
        first = (rank * n_batches + b) * graphs_per_gpu
        if in_feats != IN_FEATS:      # --config real
            samples = [real_sample(first + i, in_feats) for i in range(graphs_per_gpu)]
            samples = [(s[0], s[1], s[2].astype(np.float32), s[3]) for s in samples]
        else:
            samples = [synth.make_sample(first + i, kind=kind, in_feats=IN_FEATS) for i in range(graphs_per_gpu)]
        g = gts.batch([s[1] for s in samples]).to(device)
        g.dev()  # upload CSR now, not inside the timed region
        feats = torch.from_numpy(np.concatenate([s[2] for s in samples])).to(device)
        labels = torch.from_numpy(np.concatenate([s[3] for s in samples])).to(device)
        batches.append((g, feats, labels))
    return batches


def host_cores():
    """Cores this process may actually use: CPU affinity, capped by the cgroup CPU quota."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return cores


def hyperparams(cfg):
    from utils.hyperparam_helpers import FullParamSet

    return FullParamSet(1, cfg.get("in_feats", IN_FEATS), N_CLASSES, 1e-4, 0.98, 1e-4, CLASS_WEIGHTS,
                        cfg["layer_sizes"], 0, cfg["heads"], cfg["residuals"])


def cpu_baseline(cfg, graphs_per_gpu, kind, steps):
    """Oracle training step on the host cores, same workload, bounded sample."""
    from gts import synth
    from oracle import graph_ref, torch_ref

    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu baseline on {cores} host cores (os.cpu_count()={os.cpu_count()})")
    torch.manual_seed(0)
    net = torch_ref.ref_init_graph_net(cfg["model"], hyperparams(cfg))
    opt = torch_ref.make_optimizer(net)
    if "in_feats" in cfg:
        samples = [real_sample(i, cfg["in_feats"]) for i in range(graphs_per_gpu)]
        samples = [(s[0], s[1], s[2].astype(np.float32), s[3]) for s in samples]
    else:
        samples = [synth.make_sample(i, kind=kind, in_feats=IN_FEATS) for i in range(graphs_per_gpu)]
    ref = graph_ref.batch_ref([graph_ref.RefGraph(s[1].src, s[1].dst, s[1].n) for s in samples])
    tg = torch_ref.TGraph(ref)
    feats = torch.from_numpy(np.concatenate([s[2] for s in samples]))
    labels = torch.from_numpy(np.concatenate([s[3] for s in samples]))
    w = torch.tensor(CLASS_WEIGHTS)
    t0 = time.perf_counter()
    torch_ref.train_step(net, tg, feats, labels, w, opt)      # warm-up
    log(f"cpu warm-up step {time.perf_counter() - t0:.1f} s")
    t0 = time.perf_counter()
    for _ in range(steps):
        torch_ref.train_step(net, tg, feats, labels, w, opt)
        log(f"cpu step done at +{time.perf_counter() - t0:.1f} s")
    dt = time.perf_counter() - t0
    return {"value": graphs_per_gpu * steps / dt, "unit": "graphs/s", "cores": cores, "kind": "port",
            "sample": f"{steps} training steps (+1 warm-up) of the same batch of {graphs_per_gpu} "
                      f"{samples[0][1].n}-node {kind} graphs, {cfg['model']} {cfg['layer_sizes']}, fp32, torch CPU oracle"}


def algorithmic_bytes(kernel, n_b, e_b, arg_bytes, heads=4, dim=256, n_vox=240 ** 3):
    """Per-edge traffic model of one launch: every edge fetches its source row.
    Most of those re-reads are served by the XCD's L2, so this is NOT what HBM sees; it is kept
    as `algorithmic_*` beside the compulsory figure the roofline fraction is computed from."""
    f = 256
    if kernel == "spmm_max_fwd_f256":   # source row per edge + out row + argmax slots + int32 CSR
        return 4 * f * e_b + 4 * f * n_b + arg_bytes * f * n_b + 4 * (e_b + n_b + 1)
    if kernel == "spmm_max_bwd_f256":   # gout row + winner slots per out-edge, gx row, out-CSR + t_slot
        return (4 * f + arg_bytes * f) * e_b + 4 * f * n_b + 4 * (2 * e_b + n_b + 1)
    if kernel == "gat_fwd":             # ft slice per (edge, head) + out + attn + el/er + CSR
        return 4 * heads * dim * e_b + 4 * heads * dim * n_b + 4 * heads * (3 * e_b + 2 * n_b) + 4 * (e_b + n_b + 1)
    if kernel == "gat_bwd_edge":        # ft slice + gradient slice per (edge, head), attn in, ge out, scores, CSR
        return 2 * 4 * heads * dim * e_b + 4 * heads * (2 * e_b + 3 * n_b) + 4 * (e_b + n_b + 1)
    if kernel == "gat_bwd_src":         # gradient slice per (out-edge, head) + gft out + attn / ge per edge + out-CSR with positions
        return 4 * heads * dim * e_b + 4 * heads * dim * n_b + 4 * heads * (2 * e_b + 2 * n_b) + 4 * (2 * e_b + n_b + 1)
    if kernel == "project_rows":        # int16 id in, 16-byte row out, per voxel
        return (2 + 16) * n_vox
    raise ValueError(kernel)


def compulsory_bytes(kernel, n_b, e_b, arg_bytes, heads=4, dim=256, n_vox=240 ** 3):
    """Bytes one launch cannot avoid moving across HBM: every input and output array once."""
    f = 256
    if kernel == "spmm_max_fwd_f256":   # x in, out, argmax slots, in-CSR
        return 4 * f * n_b + 4 * f * n_b + arg_bytes * f * n_b + 4 * (e_b + n_b + 1)
    if kernel == "spmm_max_bwd_f256":   # gout in, winner slots in, gx out, out-CSR + t_slot
        return 4 * f * n_b + arg_bytes * f * n_b + 4 * f * n_b + 4 * (2 * e_b + n_b + 1)
    if kernel == "gat_fwd":             # ft in, out, attn out, el/er in, bias, in-CSR
        return (2 * 4 * heads * dim * n_b + 4 * heads * e_b + 2 * 4 * heads * n_b + 4 * heads * dim
                + 4 * (e_b + n_b + 1))
    if kernel == "gat_bwd_edge":        # ft in, gradient in, attn in, ge out, el / er in, ger out, in-CSR
        return 2 * 4 * heads * dim * n_b + 2 * 4 * heads * e_b + 3 * 4 * heads * n_b + 4 * (e_b + n_b + 1)
    if kernel == "gat_bwd_src":         # gradient in, gft out, attn + ge in, gel out (+ ger, attn_l / attn_r), out-CSR + t_pos
        return (2 * 4 * heads * dim * n_b + 2 * 4 * heads * e_b + 2 * 4 * heads * n_b + 2 * 4 * heads * dim
                + 4 * (2 * e_b + n_b + 1))
    if kernel == "project_rows":        # ids in, rows out, the node table once
        return (2 + 16) * n_vox + 16 * n_b
    raise ValueError(kernel)


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside torchrun: start the ranks as a CHILD process
    (one per GPU, RCCL) before this process has made any GPU call, and leave with its code."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    log("not under torchrun: " + " ".join(cmd))
    return subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))


def main():
    args = parse_args()
    cfg = CONFIGS[args.config]
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))          # nothing above touched the GPU
    from gts import dist as gdist

    rank, world, local_rank = gdist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs MI355X GPUs (the HIP path has no CPU fallback)")
    device = torch.device("cuda", torch.cuda.current_device())

    from gts import _lib, dense, ops, synth
    from model.gnn_model import GNN

    _lib.load()
    torch.manual_seed(0)
    # 48 steps per epoch: the order of the reference's own epochs (a few hundred BraTS volumes in batches of 6)
    dataset = RealDataset(48 * args.graphs_per_gpu, cfg["in_feats"]) if args.config == "real" else None
    with contextlib.redirect_stdout(sys.stderr):   # stdout carries the JSON line only
        model = GNN(cfg["model"], hyperparams(cfg), dataset, batch_size=args.graphs_per_gpu)
    if world > 1:
        gdist.broadcast_parameters(model.net.parameters(), src=0)
        model.grad_sync = gdist.FlatGradSync(model.net.parameters())
    if dataset is not None:
        # ingest: the cluster row schedules are built once per graph (9 ms per 15k-node graph on one host core), here instead
        # of inside the first epoch's timed blocks
        for which in model._schedules_wanted(sum(s[1].n for s in dataset.items[:args.graphs_per_gpu])):
            for sample in dataset.items:
                sample[1].cluster_schedule(which)
    batches = build_batches(rank, args.graphs_per_gpu, args.graph_kind, n_batches=2, device=device, cfg=cfg)
    n_b, e_b = batches[0][0].n, batches[0][0].number_of_edges()

    timers = KernelTimers()
    for name in cfg["hbm_kernels"]:
        ops.KERNEL_TIMERS[name] = timers.bracket(name)
    dense.GEMM_TIMER = lambda kind, flops, launch: timers.bracket("gemm_" + kind, flops)(launch)
    gdist.COLLECTIVE_TIMER = timers.bracket("all_reduce")

    if args.config == "c5":
        model.net.eval()
        svs = torch.from_numpy(synth.supervoxel_volume((240, 240, 240), cube=10, shell=20)).to(device)
        bg = torch.tensor([1.0, -1.0, -1.0, -1.0], device=device)
        per_graph = n_b // args.graphs_per_gpu

        def step(i):
            g, feats, _ = batches[i % len(batches)]
            with torch.no_grad():
                logits = model.net(g, feats)
                vols = [ops.project_rows(svs, logits[k * per_graph:(k + 1) * per_graph], bg)
                        for k in range(args.graphs_per_gpu)]
            return vols[-1][0, 0, 0, 0]
    elif args.config == "real":
        model.net.train()

        def fresh_batches():      # what GNN.run_epoch iterates: collate on the host, upload on the copy stream, one step ahead
            while True:
                for batch in model._device_batches():
                    yield batch
        stream_of_batches = fresh_batches()

        def step(i):
            batch = next(stream_of_batches)
            return model.empty_step() if batch is None else model.train_step(*batch)

        def resident_step(i):
            g, feats, labels = batches[i % len(batches)]
            return model.train_step(g, feats, labels)
    else:
        model.net.train()

        def step(i):
            g, feats, labels = batches[i % len(batches)]
            return model.train_step(g, feats, labels)

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed_block(step=step):
        """EXACTLY --steps steps between two fences; (seconds [max over ranks], host enqueue s, last)."""
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            last = step(i)
        enqueue = time.perf_counter() - t0      # CPU time to enqueue K steps (no sync inside)
        fence()
        elapsed = time.perf_counter() - t0
        own = elapsed
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=device)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, enqueue, last, own

    log(f"rank {rank}/{world}: {args.config} batches resident (N_b={n_b}, E_b={e_b}); warm-up")
    for i in range(args.warmup):
        step(i)
    log(f"timed region: {args.blocks} blocks of {args.steps} steps")
    blocks = [timed_block() for _ in range(args.blocks)]
    times = sorted(b[0] for b in blocks)
    elapsed = float(np.median(times))            # the headline: median block
    host_enqueue = float(np.median([b[1] for b in blocks]))
    last = blocks[-1][2]
    # one more block of the same K steps with HIP events around every launch of the kernels the
    # roofline objects describe (kept out of the blocks above: events cost a few us per launch)
    timers.enabled = ops.INSTRUMENTED = True
    instrumented = timed_block()[0]
    timers.enabled = ops.INSTRUMENTED = False
    # per-rank view (N > 1): every rank's own wall time for the last timed block and the time its all-reduce launches took
    # (a rank that finishes its graphs early waits there: the spread between ranks IS the load imbalance)
    per_rank = None
    if world > 1:
        ar = timers.summary("all_reduce")
        mine = torch.tensor([blocks[-1][3], (ar["total_ms"] / ar["launches"]) if ar else 0.0], dtype=torch.float64, device=device)
        everyone = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(everyone, mine)
        per_rank = [[float(v) for v in t.tolist()] for t in everyone]
    # --config real: the same steps on two RESIDENT batches (every rank; the blocks fence collectively): what the host path costs
    resident = sorted(timed_block(resident_step)[:2] for _ in range(3))[1] if args.config == "real" else None
    log("block seconds in order: " + " ".join(f"{b[0]:.4f}" for b in blocks))
    log(f"blocks: median {elapsed:.4f} s, min {times[0]:.4f}, max {times[-1]:.4f} for {args.steps} steps; "
        f"instrumented block {instrumented:.4f} s (host enqueue {1e3 * host_enqueue / args.steps:.2f} ms/step)")
    global_batch = args.graphs_per_gpu * world
    value = global_batch * args.steps / elapsed

    if rank == 0:
        arg_b = batches[0][0].arg_bytes
        # the committed profiles ran the lattice workload at 4, 8 and 32 graphs per GPU (C2 / C3 models)
        tag = "real_" if args.config == "real" else {4: "", 8: "b8_", 32: "b32_"}.get(args.graphs_per_gpu)
        profiled = tag is not None and args.graph_kind == "lattice" and args.config in ("c2", "c3", "c4", "c5", "real")

        def from_profile(fname, key):
            path = os.path.join(REPO, "profiles", fname)
            if not (profiled and os.path.exists(path)):
                return None, None
            with open(path) as fh:
                value = json.load(fh).get(key)
            return (value, "profiles/" + fname) if value is not None else (None, None)

        workloads = {
            "c2": "C2: 7xGraphSAGE-pool-256 (8 SAGEConv), fwd+weighted-CE+bwd+AdamW",
            "c4": "C4: 7xGraphSAGE-pool-256 (8 SAGEConv) data-parallel, fwd+weighted-CE+bwd+gradient all-reduce+AdamW, "
                  f"global batch {args.graphs_per_gpu * world} graphs over {world} GPU(s) (BASELINE.json: 64 over 8)",
            "real": "the reference's training workload: 4xGraphSAGE-pool-256 (5 SAGEConv), in_feats 20, "
                    "fwd+weighted-CE+bwd+AdamW through GNN's loader path (fresh batch every step: host collate + upload "
                    "one step ahead)",
            "c3": "C3: GAT 4 layers x 4 heads x 256 (5 GATConv), fwd+weighted-CE+bwd+AdamW",
            "c5": "C5: no-grad forward of 7xGraphSAGE-pool-256 + logits projection of every graph to an "
                  "int16 240^3 partition (fp32 x4 rows)",
        }
        rate = lambda t: global_batch * args.steps / t          # noqa: E731
        result = {
            "metric": cfg["metric"], "value": round(value, 3),
            "unit": "volumes/s" if args.config == "c5" else "graphs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{workloads[args.config]}, {args.graphs_per_gpu} x {n_b // args.graphs_per_gpu}-node/"
                                   f"{e_b // args.graphs_per_gpu}-edge {args.graph_kind} graphs per GPU, "
                                   f"{cfg.get('in_feats', IN_FEATS)}-chan feat, fp32",
                       "global_batch": global_batch, "nodes_per_batch": n_b, "edges_per_batch": e_b,
                       "parallelism": f"dp{world}", "last_value": round(float(last), 6),
                       "host_enqueue_ms_per_step": round(1e3 * host_enqueue / args.steps, 3)},
            "blocks": {"n": args.blocks, "steps_per_block": args.steps, "value": "median block",
                       "median": round(rate(elapsed), 3), "min": round(rate(times[-1]), 3),
                       "max": round(rate(times[0]), 3)},
            "ranks": world,
            "backend": torch.distributed.get_backend() if world > 1 else None,
        }
        if resident is not None:
            # (with resident batches nothing in the loop waits for the loader: `host_enqueue` is the pure cost of issuing a step)
            result["config"]["resident_batches"] = {"value": round(rate(resident[0]), 3),
                                                    "ms_per_step": round(1e3 * resident[0] / args.steps, 4),
                                                    "host_enqueue_ms_per_step": round(1e3 * resident[1] / args.steps, 3)}
        # --- K11, where most of the step goes (all three GEMM forms are one kernel template)
        kinds = {k: timers.summary("gemm_" + k) for k in ("fwd", "igrad", "wgrad")}
        kinds = {k: v for k, v in kinds.items() if v}
        sample = f"HIP events around every launch in one extra block of {args.steps} steps after the timed blocks"
        # the timed blocks issue the SAGE-pool layer stack through the one-call entry points (gts_sage_pool_stack_*_f32);
        # events need single launches, so the instrumented block issues the same kernels, in the same order, one library
        # call each (gts/nn.py::_SagePoolStack) — same device work, more host calls
        path = ("launch-by-launch: one C-ABI call per kernel (the timed blocks enqueue the same launches through "
                "gts_sage_pool_stack_fwd_f32 / _bwd_f32)") if cfg["model"] == "GSpool" else "as in the timed blocks"
        if kinds:
            flops = sum(v["work"] for v in kinds.values())
            secs = sum(v["total_ms"] for v in kinds.values()) * 1e-3
            tf = flops / secs / 1e12
            result["roofline"] = {
                "bound": "mfma",
                "kernel": "K11, every forward / input-gradient / weight-gradient launch of the step: "
                          "gts::gemm_panel_direct_kernel (16x16x4 MFMA row panels: the chained layer GEMMs, ~58 % of a C2 step) "
                          "and gts::wgrad_stream_kernel (32x32x2 MFMA split-reduction weight gradients, ~25 %)",
                "achieved": round(tf, 1), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None,
                "flops_per_step": flops / args.steps, "gemm_ms_per_step": round(1e3 * secs / args.steps, 4),
                "share_of_step": round(secs / instrumented, 3), "path": path,
                "by_kind": {k: {"launches_per_step": v["launches"] / args.steps,
                                "avg_launch_us": round(1e3 * v["total_ms"] / v["launches"], 2),
                                "tflops": round(v["work"] / (v["total_ms"] * 1e-3) / 1e12, 1)}
                            for k, v in kinds.items()},
                "sample": sample}
        # --- the HBM-bound kernels of the path: fraction from COMPULSORY bytes (each array once)
        # what the events of an entry bracket: one library call = these kernels (clustered form when the graph has a schedule)
        calls = {
            "spmm_max_fwd_f256": "gts_spmm_max_fwd_cluster_f32: spmm_cluster_stream_kernel<fwd> (from 100 000 rows), else gts_spmm_max_fwd_f32",
            "spmm_max_bwd_f256": "gts_spmm_max_bwd_cluster_f32: spmm_cluster_stream_kernel<bwd>, else gts_spmm_max_bwd_f32",
            "gat_fwd": "gts_gat_fwd_cluster_f32: gat_weights_one_chunk_kernel (edge softmax) + gat_cluster_stream_kernel<fwd>, "
                       "else gts_gat_fwd_f32",
            "gat_bwd_edge": "gts_gat_bwd_edge_cluster_f32: gat_cluster_stream_kernel<edge> + gat_edge_finish_kernel (graphs with a "
                            "'gat_edge_in' schedule: in-degree <= 8, >= 20 000 rows), else gts_gat_bwd_edge_f32: gat_bwd_edge_kernel",
            "gat_bwd_src": "gts_gat_bwd_src_cluster_f32: gat_weights_one_chunk_kernel + gat_cluster_stream_kernel<bwd>, "
                           "else gts_gat_bwd_src_f32",
            "project_rows": "gts_project_rows_i16: project_rows_kernel",
        }
        hbm = []
        for name in cfg["hbm_kernels"]:
            s = timers.summary(name)
            if not s:
                continue
            us = 1e3 * s["total_ms"] / s["launches"]
            need = compulsory_bytes(name, n_b, e_b, arg_b)
            alg = algorithmic_bytes(name, n_b, e_b, arg_b)
            pmc_file = "pmc_traffic_c3.json" if args.config == "c3" else f"pmc_traffic{'_' + tag[:-1] if tag else ''}.json"
            traffic, source = from_profile(pmc_file, name + "_bytes_per_launch")
            rocprof_us, rsource = from_profile("rocprof_kernel_avg.json", (tag or "") + name + "_avg_us")
            gbs = need / (us * 1e-6) / 1e9
            hbm.append({"bound": "hbm", "kernel": name, "call": calls.get(name), "model": "compulsory bytes: every input/output array once",
                        "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(gbs / HBM_PEAK_GBS, 4), "compulsory_bytes_per_launch": need,
                        "traffic": traffic, "traffic_source": source,
                        "algorithmic_bytes_per_launch": alg,
                        "algorithmic_gbs": round(alg / (us * 1e-6) / 1e9, 1),
                        "avg_launch_us": round(us, 2), "launches_timed": s["launches"],
                        "rocprof_avg_launch_us": rocprof_us, "rocprof_source": rsource, "sample": sample, "path": path})
        result["roofline_hbm"] = hbm
        if "roofline" not in result and hbm:
            result["roofline"] = hbm[0]
        ar = timers.summary("all_reduce")
        if ar:
            result["all_reduce"] = {"payload_bytes": int(model.grad_sync.flat.numel() * 4),
                                    "avg_us": round(1e3 * ar["total_ms"] / ar["launches"], 1),
                                    "launches_timed": ar["launches"]}
        if per_rank is not None:
            step_ms = [1e3 * r[0] / args.steps for r in per_rank]
            result["per_rank"] = {"step_ms": [round(v, 4) for v in step_ms], "step_ms_max": round(max(step_ms), 4),
                                  "step_ms_min": round(min(step_ms), 4),
                                  "all_reduce_us": [round(1e3 * r[1], 1) for r in per_rank],
                                  "note": "own wall time per step of the last timed block (fences on both sides), and the mean "
                                          "duration of the rank's all-reduce launches in the instrumented block: a rank with "
                                          "less work waits longer inside the collective"}
        if world == 1 and not args.no_cpu_baseline and args.config != "c5":
            cpu_steps = 1 if args.config == "c3" else args.cpu_steps   # a GAT step takes ~1 min on the CPU
            result["cpu_baseline"] = cpu_baseline(cfg, args.graphs_per_gpu, args.graph_kind, cpu_steps)
        print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
