"""Host side of the cluster row schedules (gts/schedule.py, gts_cluster_schedule in libgts_hip.so): the schedule
is an exact cover of the rows, keeps every row's edges in CSR slot order (so the clustered K1 / K2 reduce in the
order DGL's copy_u / max does, /root/reference/model/networks.py:25,28,30), respects its limits, and the schedule
of a batch is the shifted concatenation of its members'.  Runs without a GPU."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import gts
from gts import schedule, synth
from tests.helpers import random_coo


@pytest.fixture(scope="module", autouse=True)
def _lib(hip_lib):
    old = schedule.MAX_MEAN_DEGREE_FORWARD, schedule.MAX_DEGREE_BACKWARD
    schedule.MAX_MEAN_DEGREE_FORWARD, schedule.MAX_DEGREE_BACKWARD = 1e9, 10 ** 9     # density rules: tested on their own below
    yield hip_lib
    schedule.MAX_MEAN_DEGREE_FORWARD, schedule.MAX_DEGREE_BACKWARD = old


def test_density_rules_pick_the_kernel_per_graph():
    """Speed policy, not semantics: dense or irregular graphs keep the plain kernels (profiles/r03_cluster_other_graphs.log)."""
    old = schedule.MAX_MEAN_DEGREE_FORWARD, schedule.MAX_DEGREE_BACKWARD
    schedule.MAX_MEAN_DEGREE_FORWARD, schedule.MAX_DEGREE_BACKWARD = 11.0, 8
    try:
        lattice = synth.lattice_graph((10, 10, 10), self_loops=True)            # degree <= 7: both clustered
        assert lattice.cluster_schedule("in") is not None and lattice.cluster_schedule("out") is not None
        sparse = synth.geometric_graph(n=3000, k=6, seed=1)                     # mean degree ~7, rows of up to ~15 edges
        assert sparse.cluster_schedule("in") is not None and sparse.cluster_schedule("out") is None
        dense = synth.geometric_graph(n=3000, k=16, seed=1)                     # mean degree ~19
        assert dense.cluster_schedule("in") is None and dense.cluster_schedule("out") is None
    finally:
        schedule.MAX_MEAN_DEGREE_FORWARD, schedule.MAX_DEGREE_BACKWARD = old


def _check_cover(g, which, sched):
    ip, ix = (g.indptr, g.indices) if which == "in" else (g.t_indptr, g.t_indices)
    seen = np.zeros(g.n, dtype=np.int64)
    for rows, srcs, per_row in sched.decode():
        assert 1 <= len(rows) <= sched.limits[0] and len(srcs) <= sched.limits[1]
        assert len(set(srcs.tolist())) == len(srcs), "a neighbour row is staged twice in one cluster"
        assert sum((len(nb) + 7) // 8 * 8 for _, nb, _ in per_row) <= sched.limits[2]
        assert list(rows) == sorted(rows)
        for r, nb, tag in per_row:
            seen[r] += 1
            assert nb == ix[ip[r]:ip[r + 1]].tolist(), "edges of a row must stay in CSR slot order"
            if which == "out":
                assert tag == g.t_slot[ip[r]:ip[r + 1]].tolist()
    assert (seen == 1).all(), "every row belongs to exactly one cluster"


@pytest.mark.parametrize("which", ["in", "out"])
def test_lattice_schedule_is_an_exact_cover_and_saves_two_thirds_of_the_row_fetches(which):
    g = synth.lattice_graph()
    s = g.cluster_schedule(which)
    assert s is not None and s.n_rows == g.n and s.n_edges == g.number_of_edges()
    _check_cover(g, which, s)
    assert s.staged_rows < 0.4 * s.n_edges          # 1.9 rows per destination instead of 5.76
    assert 2 * s.lds_bytes(0 if which == "in" else 1) <= 160 * 1024   # a ring needs two slots at least; four fit


def test_geometric_graph_with_self_loops_and_isolated_rows():
    base = synth.geometric_graph(n=1500, k=8, seed=3, self_loops=True)
    keep = base.dst % 97 != 5                        # rows 5, 102, ... lose every in-edge
    g = gts.Graph(base.src[keep], base.dst[keep], base.n)
    assert g.min_in_degree == 0
    for which in ("in", "out"):
        s = g.cluster_schedule(which)
        assert s is not None
        _check_cover(g, which, s)


def test_random_graph_is_not_worth_a_schedule_and_hubs_do_not_fit():
    assert synth.random_graph(n=3000, n_pairs=6000, seed=1).cluster_schedule("in") is None
    n = 400
    src = np.concatenate([np.arange(n), np.arange(n - 1)])
    dst = np.concatenate([np.zeros(n, dtype=np.int64), np.arange(1, n)])     # row 0 has 400 in-edges
    g = gts.Graph(src, dst, n)
    assert g.cluster_schedule("in") is None          # beyond max_srcs: the plain kernel runs


def test_records_repeat_the_last_neighbour_and_repack_losslessly():
    g = synth.lattice_graph((7, 6, 5))
    s = g.cluster_schedule("out")
    lay = s.layout
    for r in s.rec:
        n_srcs = int(r[1])
        tail = r[lay.srcs + n_srcs:lay.eoff]
        assert n_srcs > 0 and (tail == r[lay.srcs + n_srcs - 1]).all()   # gathers run in whole pairs / octets
    wide = s.with_loc_words(s.loc_words + 8)
    assert wide.rec.shape[1] == s.rec.shape[1] + 16
    _check_cover(g, "out", wide)
    assert np.array_equal(wide.with_loc_words(s.loc_words).rec, s.rec)


def test_batch_schedule_is_the_shifted_concatenation():
    parts = [synth.lattice_graph((6, 5, 7)), synth.geometric_graph(n=500, k=6, seed=2), synth.lattice_graph((4, 9, 5))]
    b = gts.batch(parts)
    for which in ("in", "out"):
        sb = b.cluster_schedule(which)
        assert sb is not None and sb.n_clusters == sum(p.cluster_schedule(which).n_clusters for p in parts)
        _check_cover(b, which, sb)
    mixed = gts.batch([parts[0], synth.random_graph(n=3000, n_pairs=6000, seed=1)])
    assert mixed.cluster_schedule("in") is None      # one member without a schedule: the batch has none


def test_limits_are_respected_and_bad_limits_rejected():
    g = synth.lattice_graph((8, 8, 8))
    s = schedule.ClusterSchedule.build(g.indptr, g.indices, g.t_indptr, g.t_indices, None, (16, 40, 128))
    _check_cover(g, "in", s)
    assert schedule.ClusterSchedule.build(g.indptr, g.indices, g.t_indptr, g.t_indices, None, (16, 4, 128)) is None
    with pytest.raises(ValueError):
        schedule.ClusterSchedule.build(g.indptr, g.indices, g.t_indptr, g.t_indices, None, (16, 300, 128))
    with pytest.raises(ValueError):                  # records above 512 words
        schedule.ClusterSchedule.build(g.indptr, g.indices, g.t_indptr, g.t_indices, None, (200, 256, 2000))
    tiny = schedule.ClusterSchedule.build(g.indptr, g.indices, g.t_indptr, g.t_indices, None, (1, 6, 8))
    assert tiny.n_clusters == g.n                    # more clusters than the first guess of the record buffer
    _check_cover(g, "in", tiny)


@settings(max_examples=40, deadline=None)
@given(n=st.integers(1, 120), e=st.integers(0, 500), seed=st.integers(0, 10 ** 6),
       rows=st.integers(1, 20), srcs=st.integers(30, 60))
def test_any_graph_any_limits_exact_cover(n, e, seed, rows, srcs):
    src, dst = random_coo(n, e, seed)
    g = gts.Graph(src, dst, n)
    for which in ("in", "out"):
        ip, ix, tp, tx = (g.indptr, g.indices, g.t_indptr, g.t_indices) if which == "in" else \
            (g.t_indptr, g.t_indices, g.indptr, g.indices)
        s = schedule.ClusterSchedule.build(ip, ix, tp, tx, g.t_slot if which == "out" else None, (rows, srcs, 400))
        if s is None:
            deg = np.diff(ip)
            assert deg.max() > srcs or (which == "out" and g.t_slot.max() > 255)
        else:
            _check_cover(g, which, s)


def test_gat_schedules_use_their_own_limits_and_carry_no_tags():
    g = synth.lattice_graph((9, 8, 7))
    for which in ("gat_in", "gat_out"):
        s = g.cluster_schedule(which)
        assert s is not None and not s.tagged and s.limits == schedule.limits(which) == (32, 64, 256)
        rows = np.concatenate([c[0] for c in s.decode()])
        assert np.array_equal(np.sort(rows), np.arange(g.n))            # an exact cover of the rows
    assert g.cluster_schedule("gat_out") is not g.cluster_schedule("out")
    dense = synth.geometric_graph(n=600, k=12, seed=1)                   # rows of two 8-edge chunks: the plain kernels keep them
    assert dense.cluster_schedule("gat_in") is None
