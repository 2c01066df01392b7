"""gts_collate_batch (host C ABI, csrc/gts_collate.hip) against the Python collate it stands in for on the training
loader's hot path: `data_loader.minibatch_graphs` = gts.batch + np.concatenate + ClusterSchedule.concat — the restated
/root/reference/data_processing/data_loader.py:165-169.  Bytes must be equal: features, labels, both CSRs, the degree
vectors, the schedule records.  No GPU needed (host function writing into host memory)."""
import ctypes

import numpy as np
import pytest

import gts
from data_processing.data_loader import minibatch_graphs
from gts import _lib, collate, schedule, synth


def _samples(sizes, in_feats=20, f64=True, seed=7, labelled=True):
    out = []
    for i, dims in enumerate(sizes):
        g = synth.lattice_graph(dims) if isinstance(dims, tuple) else synth.random_graph(n=dims, n_pairs=3 * dims, seed=seed + i)
        feats = synth.node_features(g.n, in_feats, seed + i)
        feats = feats.astype(np.float64) * 1.000000123 if f64 else feats
        out.append((f"s{i}", g, feats, synth.node_labels(g.n, seed + i) if labelled else None))
    return out


def _segments(plan, block, width, labelled):
    n, e = int(plan.n_nodes), int(plan.n_edges)

    def seg(off, count, dtype):
        return block[off:off + count * np.dtype(dtype).itemsize].view(dtype)
    sizes = [n + 1, e, n + 1, e, e, e, n, n]
    csr = [seg(plan.csr[q], sizes[q], np.int32) for q in range(8)]
    return (seg(plan.features, n * width, np.float32).reshape(n, width),
            seg(plan.labels, n, np.int64) if labelled else None, csr)


@pytest.mark.parametrize("threads", [1, 4])
@pytest.mark.parametrize("f64", [True, False])
def test_collated_block_equals_the_python_collate(hip_lib, threads, f64):
    samples = _samples([(6, 5, 4), (7, 7, 3), 300, (4, 4, 4)], f64=f64)
    kinds = ("out", "in")
    plan, width, labelled, block = collate.collate_host(samples, kinds, lambda nbytes: np.full(nbytes, 0xAB, np.uint8), threads)
    ids, union, feats, labels = minibatch_graphs(samples)
    got_f, got_l, csr = _segments(plan, block, width, labelled)
    assert width == 20 and labelled
    assert got_f.tobytes() == feats.numpy().tobytes()
    assert got_l.tobytes() == labels.numpy().tobytes()
    for name, got in zip(("indptr", "indices", "t_indptr", "t_indices", "t_slot", "t_pos"), csr):
        assert np.array_equal(got, getattr(union, name)), name
    deg = np.diff(union.indptr).astype(np.float32)
    assert np.array_equal(csr[6].view(np.float32), np.maximum(deg, np.float32(1)))
    assert np.array_equal(csr[7].view(np.float32), deg + np.float32(1))
    for k, which in enumerate(kinds):
        want = union.cluster_schedule(which)
        if want is None:
            assert plan.sched[k] < 0 and plan.sched_clusters[k] == -1
            continue
        words = int(plan.sched_record_words[k])
        assert (plan.sched_clusters[k], plan.sched_loc_words[k], words) == (want.n_clusters, want.loc_words, want.layout.words)
        got = block[plan.sched[k]:plan.sched[k] + 4 * want.n_clusters * words].view(np.int32).reshape(-1, words)
        assert np.array_equal(got, want.rec), which
    # every segment starts 256-byte aligned and nothing is written past the planned total
    offs = [plan.features, plan.labels, *plan.csr, *[plan.sched[k] for k in range(len(kinds)) if plan.sched[k] >= 0]]
    assert all(o % 256 == 0 for o in offs) and plan.total_bytes == block.size


def test_members_with_different_record_widths_are_repacked(hip_lib):
    """A 20-edge hub row makes one member's clusters need more per-edge words than the lattice members'."""
    hub = np.arange(1, 21)
    src = np.concatenate([hub, np.zeros(20, np.int64), np.arange(21, 60), np.arange(22, 61)])
    dst = np.concatenate([np.zeros(20, np.int64), hub, np.arange(22, 61), np.arange(21, 60)])
    g_hub = gts.Graph(src, dst, 61)
    lat = synth.lattice_graph((5, 5, 5))
    samples = [("a", lat, synth.node_features(lat.n, 4, 1), synth.node_labels(lat.n, 1)),
               ("b", g_hub, synth.node_features(61, 4, 2), synth.node_labels(61, 2))]
    for g in (lat, g_hub):      # force schedules regardless of the worthwhile rule: the repacking is what is tested
        g._sched["in"] = schedule.ClusterSchedule.build(g.indptr, g.indices, g.t_indptr, g.t_indices, None, schedule.limits("in"))
    assert lat._sched["in"].loc_words != g_hub._sched["in"].loc_words
    plan, width, _, block = collate.collate_host(samples, ("in",), lambda nbytes: np.zeros(nbytes, np.uint8), 2)
    want = schedule.ClusterSchedule.concat([lat._sched["in"], g_hub._sched["in"]], np.array([0, lat.n, lat.n + 61], np.int32))
    words = int(plan.sched_record_words[0])
    got = block[plan.sched[0]:plan.sched[0] + 4 * want.n_clusters * words].view(np.int32).reshape(-1, words)
    assert plan.sched_loc_words[0] == want.loc_words and np.array_equal(got, want.rec)


def test_a_member_without_a_schedule_drops_the_kind_and_unlabelled_batches_work(hip_lib):
    samples = _samples([(5, 5, 5), 64], in_feats=4, f64=False, labelled=False)
    samples[1][1]._sched["out"] = None
    plan, width, labelled, block = collate.collate_host(samples, ("out",), lambda nbytes: np.zeros(nbytes, np.uint8), 1)
    assert not labelled and plan.labels == -1 and plan.sched[0] == -1
    got_f, _, csr = _segments(plan, block, width, False)
    union = gts.batch([s[1] for s in samples])
    assert np.array_equal(csr[0], union.indptr) and np.array_equal(csr[5], union.t_pos)
    assert np.array_equal(got_f, np.concatenate([s[2] for s in samples]))


def test_collate_rejects_bad_arguments_without_touching_memory(hip_lib):
    m = (_lib.CollateMember * 1)()
    plan = _lib.CollatePlan()
    assert hip_lib.gts_collate_plan(None, 1, 4, None, 0, ctypes.byref(plan)) == -1
    assert hip_lib.gts_collate_plan(m, 0, 4, None, 0, ctypes.byref(plan)) == -2
    m[0].n_nodes, m[0].n_edges = 4, 0
    assert hip_lib.gts_collate_plan(m, 1, 4, None, 0, ctypes.byref(plan)) == -1          # no indptr
    one = np.zeros(8, np.int32)
    m[0].indptr = m[0].t_indptr = one.ctypes.data
    feats = np.zeros((4, 4), np.float32)
    m[0].features, m[0].feat_bytes = feats.ctypes.data, 2
    assert hip_lib.gts_collate_plan(m, 1, 4, None, 0, ctypes.byref(plan)) == -3          # fp16 features
    m[0].feat_bytes = 4
    assert hip_lib.gts_collate_plan(m, 1, 4, None, 0, ctypes.byref(plan)) == 0
    dst = np.zeros(int(plan.total_bytes), np.uint8)
    assert hip_lib.gts_collate_batch(m, 1, 4, None, 0, dst.ctypes.data, plan.total_bytes - 1, 1, ctypes.byref(plan)) == -2
    assert hip_lib.gts_collate_batch(m, 1, 4, None, 0, None, plan.total_bytes, 1, ctypes.byref(plan)) == -1
    assert hip_lib.gts_collate_batch(m, 1, 4, None, 0, dst.ctypes.data, plan.total_bytes, 1, ctypes.byref(plan)) == 0


def test_collated_batch_builds_its_host_arrays_lazily(hip_lib):
    """The graph object the collator hands out: sizes and degree bounds without touching the host arrays, the arrays
    themselves (and the union's schedule records) through the Python path on demand."""
    import torch

    samples = _samples([(5, 5, 5), (6, 5, 4)], in_feats=4, f64=False)
    graphs = [s[1] for s in samples]
    off = np.array([0, graphs[0].n, graphs[0].n + graphs[1].n], np.int32)
    g = collate.CollatedBatch(graphs, off, sum(x.number_of_edges() for x in graphs), torch.device("cpu"))
    want = gts.batch(graphs)
    assert g._host is None
    assert (g.n, g.number_of_edges(), g.max_in_degree, g.min_in_degree, g.arg_bytes) == \
        (want.n, want.number_of_edges(), want.max_in_degree, want.min_in_degree, want.arg_bytes)
    assert g.batch_size == 2 and g.batch_num_nodes().tolist() == [125, 120]
    assert g.max_out_degree == want.max_out_degree and g._host is None
    assert "norm" not in g.ndata
    for x in graphs:
        x.ndata["norm"] = torch.ones(x.n, 1)
    g2 = collate.CollatedBatch(graphs, off, g.number_of_edges(), torch.device("cpu"))
    assert "norm" in g2.ndata and g2.ndata["norm"].shape == (want.n, 1)
    assert np.array_equal(g.indices, want.indices) and np.array_equal(g.src, want.src) and g._host is not None
    us = collate._UnionSchedule([x.cluster_schedule("out") for x in graphs], off, schedule.limits("out"), True, 0, 0)
    assert np.array_equal(us.rec, want.cluster_schedule("out").rec)
