"""Data-parallel gradient exchange (gts.dist) on CPU: world_size 2 over gloo, with the oracle
network standing in for the HIP layers (the exchange logic is device-agnostic).

Claim under test: after FlatGradSync every rank holds the gradient of the GLOBAL-batch
class-weighted cross-entropy — not the average of per-rank mean losses."""
import os
import socket
from collections import namedtuple

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from gts import dist as gdist
from oracle import graph_ref, torch_ref
from tests.helpers import random_coo

HP = namedtuple("HP", "in_feats out_classes layer_sizes gat_heads gat_residuals")
W = [0.1, 1.0, 2.0, 2.0]


def _samples(n_graphs):
    out = []
    for i in range(n_graphs):
        n = 40 + 7 * i
        src, dst = random_coo(n, 5 * n, seed=i)
        rng = np.random.default_rng(100 + i)
        # very different class mixes per graph so that per-rank denominators differ a lot
        p = [0.97, 0.01, 0.01, 0.01] if i % 2 == 0 else [0.1, 0.3, 0.3, 0.3]
        out.append((graph_ref.RefGraph(src, dst, n), rng.standard_normal((n, 4)).astype(np.float32),
                    rng.choice(4, size=n, p=p)))
    return out


def _net():
    torch.manual_seed(0)
    return torch_ref.ref_init_graph_net("GSpool", HP(4, 4, [16, 16], None, None))


def _batch(samples):
    g = torch_ref.TGraph(graph_ref.batch_ref([s[0] for s in samples]))
    return g, torch.from_numpy(np.concatenate([s[1] for s in samples])), \
        torch.from_numpy(np.concatenate([s[2] for s in samples]))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_graphs, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    r, w, _ = gdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and gdist.world() == (rank, world)
    net = _net()
    sync = gdist.FlatGradSync(net.parameters())
    samples = _samples(n_graphs)
    mine = gdist.shard_indices(list(range(n_graphs)), 0, n_graphs // world, rank, world)
    g, x, y = _batch([samples[i] for i in mine])
    sync.zero_grad()
    sync.weighted_ce_backward(net(g, x), y, torch.tensor(W))
    loss = sync.all_reduce_and_normalise()
    torch.save({"loss": float(loss), "grads": [p.grad.clone() for p in net.parameters()], "mine": mine},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_equals_global_batch_gradient(tmp_path):
    n_graphs, world = 4, 2
    mp.spawn(_worker, args=(world, _free_port(), n_graphs, str(tmp_path)), nprocs=world, join=True)
    net = _net()
    g, x, y = _batch(_samples(n_graphs))
    loss = F.cross_entropy(net(g, x), y, weight=torch.tensor(W))
    loss.backward()
    ranks = [torch.load(tmp_path / f"rank{r}.pt", weights_only=False) for r in range(world)]
    assert sorted(ranks[0]["mine"] + ranks[1]["mine"]) == list(range(n_graphs))
    for r in ranks:
        assert abs(r["loss"] - float(loss)) < 1e-6 * abs(float(loss))
        for got, p in zip(r["grads"], net.parameters()):
            assert torch.allclose(got, p.grad, rtol=1e-5, atol=1e-7)
    for a, b in zip(ranks[0]["grads"], ranks[1]["grads"]):
        assert torch.equal(a, b)                       # replicas stay bit-identical
    # and the naive scheme (average of per-rank mean-loss gradients) is measurably different
    naive = [torch.zeros_like(p) for p in net.parameters()]
    for r in range(world):
        net.zero_grad()
        gg, xx, yy = _batch([_samples(n_graphs)[i] for i in ranks[r]["mine"]])
        F.cross_entropy(net(gg, xx), yy, weight=torch.tensor(W)).backward()
        for acc, p in zip(naive, net.parameters()):
            acc += p.grad / world
    rel = max(float((a - b).abs().max() / (b.abs().max() + 1e-12)) for a, b in zip(naive, ranks[0]["grads"]))
    assert rel > 1e-2


def test_single_process_sync_is_identity():
    net = _net()
    sync = gdist.FlatGradSync(net.parameters())
    g, x, y = _batch(_samples(3))
    sync.zero_grad()
    sync.weighted_ce_backward(net(g, x), y, torch.tensor(W))
    loss = sync.all_reduce_and_normalise()
    flat_ptr = sync.flat.data_ptr()
    ref = _net()
    ref_loss = F.cross_entropy(ref(g, x), y, weight=torch.tensor(W))
    ref_loss.backward()
    assert abs(float(loss) - float(ref_loss)) < 1e-6 * abs(float(ref_loss))
    off = 0
    for p, q in zip(net.parameters(), ref.parameters()):
        assert p.grad.data_ptr() == flat_ptr + 4 * off      # grads are views of the flat buffer
        assert torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-7)
        off += p.numel()
    sync.zero_grad()                                         # gradients are dropped, nothing launched
    assert all(p.grad is None for p in net.parameters())


def test_shard_indices_partition():
    perm = list(np.random.default_rng(0).permutation(64))
    for world in (1, 2, 4, 8):
        per_rank = 8 // world if world <= 8 else 1
        for step in range(64 // (per_rank * world)):
            got = sorted(i for r in range(world) for i in gdist.shard_indices(perm, step, per_rank, r, world))
            assert got == sorted(perm[step * per_rank * world:(step + 1) * per_rank * world])
