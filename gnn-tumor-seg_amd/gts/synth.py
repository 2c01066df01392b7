"""Synthetic supervoxel graphs / features / partitions of the BASELINE.json configs.

There is no BraTS data (and no network) on the build or GPU boxes, so bench.py, smoke()
and the GPU tests use these generators:
  A "lattice"  25x25x24 = 15 000 supervoxels in raster order, 6-neighbourhood
               => 43 175 undirected pairs => 86 350 directed edges (SLIC-like adjacency);
  B "random"   ring (i, i+1 mod N) + distinct uniform random pairs, symmetrised
               => ~90 000 directed edges at N=15 000, min in-degree >= 2.
Seeds: graph g uses seed 1000+g.
"""
import numpy as np

from .graph import Graph

CLASS_PRIOR = (0.85, 0.07, 0.05, 0.03)


def lattice_graph(dims=(25, 25, 24), self_loops=False):
    """Generator A.  Edge order = what from_networkx would give for the same nx.Graph:
    sources ascending, each source's neighbours ascending."""
    nx_, ny, nz = dims
    n = nx_ * ny * nz
    idx = np.arange(n, dtype=np.int64).reshape(nx_, ny, nz)
    pairs = []
    for axis in range(3):
        a = np.take(idx, np.arange(dims[axis] - 1), axis=axis).ravel()
        b = np.take(idx, np.arange(1, dims[axis]), axis=axis).ravel()
        pairs.append(np.stack([a, b], 1))
    und = np.concatenate(pairs)
    src = np.concatenate([und[:, 0], und[:, 1]])
    dst = np.concatenate([und[:, 1], und[:, 0]])
    if self_loops:
        src = np.concatenate([src, np.arange(n)])
        dst = np.concatenate([dst, np.arange(n)])
    order = np.lexsort((dst, src))
    return Graph(src[order], dst[order], n)


def random_graph(n=15000, n_pairs=30000, seed=1000):
    """Generator B."""
    rng = np.random.default_rng(seed)
    ring = np.stack([np.arange(n), (np.arange(n) + 1) % n], 1)
    have = set(map(tuple, np.sort(ring, 1)))
    extra = []
    while len(extra) < n_pairs:
        cand = rng.integers(0, n, size=(2 * (n_pairs - len(extra)) + 16, 2))
        for a, b in cand:
            if a == b:
                continue
            key = (min(a, b), max(a, b))
            if key in have:
                continue
            have.add(key)
            extra.append(key)
            if len(extra) == n_pairs:
                break
    und = np.concatenate([ring, np.array(extra, dtype=np.int64).reshape(-1, 2)])
    src = np.concatenate([und[:, 0], und[:, 1]])
    dst = np.concatenate([und[:, 1], und[:, 0]])
    order = np.lexsort((dst, src))
    return Graph(src[order], dst[order], n)


def geometric_graph(n=6000, k=8, seed=1000, self_loops=False):
    """Generator C ("SLIC-like"): n uniform random points in the unit cube, each joined to its k nearest
    neighbours, symmetrised; node ids in raster order of a 16^3 grid of cells (roughly the order SLIC labels
    come in).  Irregular degrees (k .. ~2k), spatial locality without a lattice.  Optional self-loops, as the
    reference's touching-adjacency graphs have (mri2graph/graphgen.py:189)."""
    from scipy.spatial import cKDTree

    rng = np.random.default_rng(seed)
    pts = rng.random((n, 3))
    cell = np.floor(pts * 16).astype(np.int64)
    order = np.lexsort((pts[:, 2], cell[:, 2], cell[:, 1], cell[:, 0]))
    pts = pts[order]
    _, nbr = cKDTree(pts).query(pts, k=k + 1)
    a = np.repeat(np.arange(n), k)
    b = nbr[:, 1:].reshape(-1)
    und = np.unique(np.sort(np.stack([a, b], 1), 1), axis=0)
    src = np.concatenate([und[:, 0], und[:, 1]])
    dst = np.concatenate([und[:, 1], und[:, 0]])
    if self_loops:
        src = np.concatenate([src, np.arange(n)])
        dst = np.concatenate([dst, np.arange(n)])
    order = np.lexsort((dst, src))
    return Graph(src[order], dst[order], n)


def node_features(n, in_feats=4, seed=1000):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((n, in_feats)).astype(np.float32)


def node_labels(n, seed=1000):
    rng = np.random.default_rng(seed + 7919)
    return rng.choice(4, size=n, p=CLASS_PRIOR).astype(np.int64)


def make_sample(g_index, kind="lattice", n=15000, in_feats=4):
    """(id, graph, feats, labels) like ImageGraphDataset.__getitem__ (data_loader.py:53-55,83)."""
    seed = 1000 + g_index
    if kind == "lattice":
        if n != 15000:
            raise ValueError("lattice generator is fixed at 25x25x24 = 15000 nodes")
        g = lattice_graph()
    elif kind == "random":
        g = random_graph(n=n, n_pairs=2 * n, seed=seed)
    elif kind == "selfloop":      # every node its own only neighbour: the reducers become row copies (what-if runs)
        ids = np.arange(n)
        g = Graph(ids, ids, n)
    else:
        raise ValueError(kind)
    return f"synth_{kind}_{g_index:04d}", g, node_features(g.n, in_feats, seed), node_labels(g.n, seed)


def supervoxel_volume(shape=(240, 240, 240), cube=10, shell=20, permute_tile=None, seed=0):
    """C5 partition: block-constant ids (raster index of the `cube`^3 block), an outer shell
    of `shell` voxels set to -1 (background); optional voxel permutation inside
    `permute_tile`^3 tiles to model ragged SLIC boundaries.  int16."""
    gx, gy, gz = (s // cube for s in shape)
    ids = np.arange(gx * gy * gz, dtype=np.int16).reshape(gx, gy, gz)
    vol = np.repeat(np.repeat(np.repeat(ids, cube, 0), cube, 1), cube, 2)
    vol = np.ascontiguousarray(vol[:shape[0], :shape[1], :shape[2]])
    if permute_tile:
        rng = np.random.default_rng(seed)
        t = permute_tile
        for x in range(0, shape[0], t):
            for y in range(0, shape[1], t):
                for z in range(0, shape[2], t):
                    blk = vol[x:x + t, y:y + t, z:z + t]
                    flat = blk.reshape(-1).copy()
                    vol[x:x + t, y:y + t, z:z + t] = flat[rng.permutation(flat.size)].reshape(blk.shape)
    if shell:
        vol[:shell], vol[-shell:] = -1, -1
        vol[:, :shell], vol[:, -shell:] = -1, -1
        vol[:, :, :shell], vol[:, :, -shell:] = -1, -1
    return vol
