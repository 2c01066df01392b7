"""Property tests (hypothesis) of the host graph code: any multigraph, any batching."""
import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

import gts
from oracle import graph_ref


@st.composite
def coo_graphs(draw, max_nodes=30, max_edges=120):
    n = draw(st.integers(min_value=1, max_value=max_nodes))
    e = draw(st.integers(min_value=0, max_value=max_edges))
    src = draw(st.lists(st.integers(0, n - 1), min_size=e, max_size=e))
    dst = draw(st.lists(st.integers(0, n - 1), min_size=e, max_size=e))
    return np.array(src, dtype=np.int64), np.array(dst, dtype=np.int64), n


@settings(max_examples=60, deadline=None)
@given(coo_graphs())
def test_csr_invariants(graph):
    src, dst, n = graph
    g = gts.Graph(src, dst, n)
    ref = graph_ref.RefGraph(src, dst, n)
    for name in ("indptr", "indices", "t_indptr", "t_indices"):
        assert np.array_equal(getattr(g, name), getattr(ref, name))
    e = len(src)
    assert g.indptr[0] == 0 and g.indptr[-1] == e and np.all(np.diff(g.indptr) >= 0)
    # every COO edge appears exactly once in each CSR, rows keep COO order
    for v in range(n):
        assert list(g.indices[g.indptr[v]:g.indptr[v + 1]]) == [s for s, d in zip(src, dst) if d == v]
        assert list(g.t_indices[g.t_indptr[v]:g.t_indptr[v + 1]]) == [d for s, d in zip(src, dst) if s == v]
    # t_pos is a permutation that maps out-edges onto their in-CSR entries
    assert sorted(g.t_pos.tolist()) == list(range(e))
    assert np.array_equal(g.indices[g.t_pos], np.repeat(np.arange(n), np.diff(g.t_indptr)))
    assert np.array_equal(g.t_slot, g.t_pos - g.indptr[g.t_indices])
    assert g.max_in_degree == (np.bincount(dst, minlength=n).max() if e else 0)


@settings(max_examples=30, deadline=None)
@given(st.lists(coo_graphs(max_nodes=12, max_edges=40), min_size=1, max_size=5))
def test_batch_is_block_diagonal_union(graphs):
    parts = [gts.Graph(s, d, n) for s, d, n in graphs]
    b = gts.batch(parts)
    off = np.cumsum([0] + [p.n for p in parts])
    assert b.n == off[-1] and b.number_of_edges() == sum(p.number_of_edges() for p in parts)
    scratch = gts.Graph(b.src, b.dst, b.n)
    for name in ("indptr", "indices", "t_indptr", "t_indices", "t_slot", "t_pos"):
        assert np.array_equal(getattr(b, name), getattr(scratch, name)), name
    for i, p in enumerate(parts):      # no edge crosses a member boundary
        lo, hi = off[i], off[i + 1]
        inside = (b.src >= lo) & (b.src < hi)
        assert np.all((b.dst[inside] >= lo) & (b.dst[inside] < hi))
