"""Shared test helpers (CPU side): small graphs, oracle/product pairing."""
import numpy as np

from oracle import graph_ref, torch_ref

# hand-made 6-node graph: tie (node 0, col 0), zero in-degree (node 3), self-loop (2->2),
# and a row whose COO order is NOT ascending (node 4: sources 5 then 1)
HAND_EDGES = [(1, 0), (2, 0), (0, 1), (0, 2), (2, 2), (3, 2), (5, 4), (1, 4), (4, 5)]
HAND_X = np.array([[1, 5], [3, 5], [3, -1], [7, 7], [2, 2], [2, 9]], dtype=np.float32)


def hand_graph():
    src, dst = zip(*HAND_EDGES)
    return graph_ref.RefGraph(np.array(src), np.array(dst), 6)


def random_coo(n, e, seed, min_in_degree=0):
    rng = np.random.default_rng(seed)
    src = rng.integers(0, n, size=e)
    dst = rng.integers(0, n, size=e)
    if min_in_degree:
        extra_dst = np.repeat(np.arange(n), min_in_degree)
        extra_src = rng.integers(0, n, size=extra_dst.size)
        src, dst = np.concatenate([src, extra_src]), np.concatenate([dst, extra_dst])
        perm = rng.permutation(src.size)
        src, dst = src[perm], dst[perm]
    return src.astype(np.int64), dst.astype(np.int64)


def ref_and_gts(src, dst, n):
    """(oracle TGraph, product gts.Graph) for the same COO."""
    import gts

    ref = graph_ref.RefGraph(src, dst, n)
    return torch_ref.TGraph(ref), gts.Graph(src, dst, n)


def copy_state(dst_module, src_module):
    dst_module.load_state_dict({k: v.detach().clone() for k, v in src_module.state_dict().items()})


def slots_to_sources(g, arg):
    """Product argmax (slot in the in-CSR row, 255/-1 = none) -> source node ids (-1 = none)."""
    arg = arg.cpu().numpy().astype(np.int64)
    none = 255 if arg.dtype == np.uint8 or g.arg_bytes == 1 else -1
    pos = g.indptr[:-1].astype(np.int64)[:, None] + arg
    out = np.full(arg.shape, -1, dtype=np.int64)
    ok = arg != none
    out[ok] = g.indices[pos[ok]]
    return out
