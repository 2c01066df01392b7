"""Product host code: gts.Graph construction, from_networkx, batch — bit-exact integers
against the oracle restatement and against networkx itself.  CPU only."""
import networkx as nx
import numpy as np
import pytest
import torch

import gts
from oracle import graph_ref
from tests.helpers import HAND_EDGES, random_coo


def _same_csr(g, ref):
    assert g.n == ref.n and g.number_of_edges() == ref.number_of_edges()
    for name in ("indptr", "indices", "t_indptr", "t_indices"):
        a = getattr(g, name)
        assert a.dtype == np.int32
        assert np.array_equal(a, getattr(ref, name)), name
    assert np.array_equal(g.src, ref.src) and np.array_equal(g.dst, ref.dst)


def _check_edge_maps(g):
    """t_pos / t_slot must point every out-CSR entry at its own in-CSR entry."""
    for u in range(g.n):
        for k in range(g.t_indptr[u], g.t_indptr[u + 1]):
            v = g.t_indices[k]
            assert g.indptr[v] <= g.t_pos[k] < g.indptr[v + 1]
            assert g.indices[g.t_pos[k]] == u
            assert g.t_slot[k] == g.t_pos[k] - g.indptr[v]
    assert sorted(g.t_pos.tolist()) == list(range(g.number_of_edges()))   # a permutation


def test_hand_graph_csr():
    src, dst = zip(*HAND_EDGES)
    g = gts.Graph(np.array(src), np.array(dst), 6)
    _same_csr(g, graph_ref.RefGraph(np.array(src), np.array(dst), 6))
    _check_edge_maps(g)
    assert g.max_in_degree == 3 and g.min_in_degree == 0 and g.arg_bytes == 1
    assert g.in_degrees().tolist() == [2, 1, 3, 0, 2, 1]


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_multigraph_csr(seed):
    src, dst = random_coo(50, 400, seed)       # duplicates and self-loops included
    g = gts.Graph(src, dst, 50)
    _same_csr(g, graph_ref.RefGraph(src, dst, 50))
    _check_edge_maps(g)


def test_empty_and_isolated():
    g = gts.Graph(np.array([], dtype=np.int64), np.array([], dtype=np.int64), 4)
    assert g.indptr.tolist() == [0] * 5 and g.number_of_edges() == 0 and g.max_in_degree == 0
    b = gts.batch([g, g])
    assert b.n == 8 and b.indptr.tolist() == [0] * 9
    with pytest.raises(ValueError):
        gts.Graph(np.array([0]), np.array([5]), 4)
    with pytest.raises(ValueError):
        gts.batch([])


def _nx_sample(seed, n=40):
    rng = np.random.default_rng(seed)
    G = nx.Graph()
    labels = rng.permutation(n * 3)[:n]                 # non-contiguous, unsorted node ids
    for u in labels:
        G.add_node(int(u), features=[float(u)], label=int(u) % 4)
    for _ in range(n * 3):
        a, b = rng.choice(labels, 2)
        G.add_edge(int(a), int(b), weight=1.0)           # may add self-loops
    return G


@pytest.mark.parametrize("seed", [0, 1])
def test_from_networkx_rule(seed):
    G = _nx_sample(seed)
    g = gts.from_networkx(G)
    ref = graph_ref.from_networkx_ref(G)
    _same_csr(g, ref)
    # the documented rule, checked against networkx directly
    H = nx.convert_node_labels_to_integers(G, ordering="sorted").to_directed()
    assert [(int(a), int(b)) for a, b in zip(g.src, g.dst)] == list(H.edges())
    n_self = nx.number_of_selfloops(G)
    assert g.number_of_edges() == 2 * (G.number_of_edges() - n_self) + n_self
    # When nodes were inserted in sorted order (what mri2graph/graphgen.py produces: ids
    # 0..N-1 in order), edges come out source-ascending, so every in-CSR row lists its
    # sources in ascending order.  For an arbitrary insertion order networkx iterates
    # sources in insertion order instead, and the rows follow THAT order.
    S = nx.Graph()
    S.add_nodes_from(sorted(G.nodes))
    S.add_edges_from(G.edges)
    s = gts.from_networkx(S)
    for v in range(s.n):
        row = s.indices[s.indptr[v]:s.indptr[v + 1]]
        assert np.all(np.diff(row) > 0)


def test_json_roundtrip_keeps_graph(tmp_path):
    from data_processing import graph_io

    G = _nx_sample(3)
    graph_io.save_networkx_graph(G, tmp_path / "g.json")
    G2 = graph_io.load_networkx_graph(tmp_path / "g.json")
    a, b = gts.from_networkx(G), gts.from_networkx(G2)
    assert np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices)


def test_batch_matches_oracle_and_resort():
    parts, refs = [], []
    for i, (n, e) in enumerate([(10, 30), (1, 0), (25, 90), (7, 7)]):
        src, dst = random_coo(n, e, seed=10 + i)
        parts.append(gts.Graph(src, dst, n))
        refs.append(graph_ref.RefGraph(src, dst, n))
        parts[-1].ndata["norm"] = torch.arange(n, dtype=torch.float32).unsqueeze(1)
    b = gts.batch(parts)
    ref = graph_ref.batch_ref(refs)
    _same_csr(b, ref)
    _check_edge_maps(b)
    # concatenating prebuilt CSRs must equal building the union from scratch
    scratch = gts.Graph(b.src, b.dst, b.n)
    for name in ("indptr", "indices", "t_indptr", "t_indices", "t_slot", "t_pos"):
        assert np.array_equal(getattr(b, name), getattr(scratch, name)), name
    assert b.batch_num_nodes().tolist() == [10, 1, 25, 7] and b.batch_size == 4
    assert b.ndata["norm"].shape == (43, 1)
    assert torch.equal(b.ndata["norm"][10:11], torch.zeros(1, 1))
    bb = gts.batch([b, parts[0]])                         # batching a batch keeps member sizes
    assert bb.batch_num_nodes().tolist() == [10, 1, 25, 7, 10]


def test_norm_ndata_rule():
    """data_loader.py:75-78: in_deg^-0.5 with inf -> 0, shape [N,1]."""
    src, dst = zip(*HAND_EDGES)
    g = gts.Graph(np.array(src), np.array(dst), 6)
    norm = torch.pow(g.in_degrees().float(), -0.5)
    norm[torch.isinf(norm)] = 0
    assert np.array_equal(norm.unsqueeze(1).numpy(), graph_ref.norm_ref(graph_ref.RefGraph(np.array(src), np.array(dst), 6)))


def test_cpu_graph_refuses_device_ops():
    src, dst = zip(*HAND_EDGES)
    g = gts.Graph(np.array(src), np.array(dst), 6)
    with pytest.raises(RuntimeError, match="no CPU path"):
        g.dev()


def test_synthetic_generators():
    from gts import synth

    g = synth.lattice_graph()
    assert g.n == 15000 and g.number_of_edges() == 86350          # SURVEY.md §8d generator A
    assert g.min_in_degree == 3 and g.max_in_degree == 6
    assert np.array_equal(g.indptr, g.t_indptr) and np.array_equal(g.indices, g.t_indices)  # symmetric
    r = synth.random_graph(n=1000, n_pairs=3000, seed=1000)         # C1 plumbing graph
    assert r.n == 1000 and r.number_of_edges() == 2 * (1000 + 3000) and r.min_in_degree >= 2
    r2 = synth.random_graph(n=1000, n_pairs=3000, seed=1000)
    assert np.array_equal(r.indices, r2.indices)
    vol = synth.supervoxel_volume((40, 40, 40), cube=10, shell=5)
    assert vol.dtype == np.int16 and vol.min() == -1 and vol.max() == 63
