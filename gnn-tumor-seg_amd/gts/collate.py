"""Batch collate off the interpreter lock: samples -> one page-locked block -> ONE asynchronous upload.

The training loader's hot path.  Stands in for what the reference does per step in
/root/reference/data_processing/data_loader.py:165-169 (`minibatch_graphs`: dgl.batch, np.concatenate, FloatTensor /
LongTensor) plus /root/reference/model/gnn_model.py:37-40 (the three `.to(device)` calls): the members' cached host
arrays (CSR, features, labels, cluster row schedules) are assembled by ONE host-side C-ABI call
(`gts_collate_batch`, csrc/gts_collate.hip; ctypes releases the interpreter lock for its whole duration) straight into
a slab of the loader's page-locked ring, in the layout the kernels read, and go to the device in one copy.  The loader
thread holds the lock for tens of microseconds per batch instead of milliseconds of numpy.

The Python path (`data_loader.minibatch_graphs` = `gts.batch` + `ClusterSchedule.concat`) stays the tested reference of
the bytes (tests/test_collate_host.py) and is what everything outside the training loop uses.  A `CollatedBatch` is a
`gts.Graph` whose device arrays exist already; its HOST arrays are assembled lazily through that Python path if anyone
asks for them.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from . import schedule as _schedule
from .graph import Graph, _DeviceCSR, _DeviceSchedule

N_THREADS = 4       # helper threads of one gts_collate_batch call (a C2 batch is 9 MB of shifted copies)


class _UnionSchedule:
    """What the kernels' callers need of a ClusterSchedule (n_clusters, limits, loc_words, tagged) for a union whose
    records exist on the device only; the host records are concatenated on demand (`rec`, `materialize()`)."""

    def __init__(self, parts, node_off, limits, tagged, loc_words, n_clusters):
        self._parts, self._node_off, self._full = parts, node_off, None
        self.limits, self.tagged, self.loc_words, self.n_clusters = tuple(limits), bool(tagged), int(loc_words), int(n_clusters)
        self.owner = None

    def materialize(self):
        if self._full is None:
            self._full = _schedule.ClusterSchedule.concat(self._parts, self._node_off)
        return self._full

    def __getattr__(self, name):          # rec, layout, n_rows, n_edges, staged_rows, decode, lds_bytes, ...
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.materialize(), name)


class _LazyNdata(dict):
    """`ndata` of a collated union: node data of the members, concatenated (and moved to the graph's device) when a key
    is first read — the training step never reads any (`norm` is written by the dataset, reference
    data_loader.py:73-78, and read by nobody)."""

    def __init__(self, graphs, device):
        super().__init__()
        self._graphs, self._device = graphs, device
        common = set(graphs[0].ndata)
        for g in graphs[1:]:
            common &= set(g.ndata)
        self._pending = common

    def _fill(self, key=None):
        for k in ([key] if key is not None else list(self._pending)):
            if k in self._pending:
                self._pending.discard(k)
                super().__setitem__(k, torch.cat([g.ndata[k] for g in self._graphs], dim=0).to(self._device))

    def __missing__(self, key):
        if key in self._pending:
            self._fill(key)
            return super().__getitem__(key)
        raise KeyError(key)

    def __contains__(self, key):
        return key in self._pending or super().__contains__(key)

    def __setitem__(self, key, value):
        self._pending.discard(key)
        super().__setitem__(key, value)

    def keys(self):
        self._fill()
        return super().keys()

    def items(self):
        self._fill()
        return super().items()

    def values(self):
        self._fill()
        return super().values()

    def __iter__(self):
        self._fill()
        return super().__iter__()

    def __len__(self):
        return len(self._pending) + super().__len__()


class CollatedBatch(Graph):
    """Block-diagonal union (dgl.batch semantics, reference data_loader.py:168) whose device arrays were assembled by
    gts_collate_batch.  Host-side CSR arrays, COO and schedule records are built through the Python path on first use."""

    _HOST_FIELDS = ("indptr", "indices", "t_indptr", "t_indices", "t_slot", "t_pos")

    def __init__(self, graphs, node_off, n_edges, device):      # noqa: super().__init__ not called: nothing is built on the host
        self.n = int(node_off[-1])
        self._n_edges = int(n_edges)
        self._members = (graphs, node_off)
        self._sched_members = (graphs, node_off)
        self._sched = {}
        self._src = self._dst = None
        self._batch_num_nodes = [s for g in graphs for s in g._batch_num_nodes]
        self.device = device
        self.ndata = _LazyNdata(graphs, device)
        self._dev_cache = {}
        self._dev = None
        self.max_in_degree = max(g.max_in_degree for g in graphs)
        self.min_in_degree = min(g.min_in_degree for g in graphs)
        self._max_out_degree = None
        self._host = None

    def _union(self):
        if self._host is None:
            from .graph import batch

            self._host = batch(self._sched_members[0])
        return self._host

    def number_of_edges(self):
        return self._n_edges

    num_edges = number_of_edges

    @property
    def max_out_degree(self):
        if self._max_out_degree is None:
            self._max_out_degree = max(g.max_out_degree for g in self._sched_members[0])
        return self._max_out_degree

    def _coo(self):
        return self._union()._coo()

    def to(self, device, **_kwargs):
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if device == self.device:
            return self
        return self._union().to(device)        # another device (or the host): the plain route


for _name in CollatedBatch._HOST_FIELDS:
    setattr(CollatedBatch, _name, property(lambda self, _n=_name: getattr(self._union(), _n)))


def _member(graph, features, labels, kinds):
    """The gts_collate_member_t of one sample, cached on its host graph (arrays it points at are kept alive there)."""
    hit = getattr(graph, "_collate_member", None)
    if hit is not None and hit[1] is features and hit[2] is labels and hit[3] == kinds:
        return hit[0]
    feats = np.asarray(features)
    if feats.dtype not in (np.float32, np.float64):
        feats = feats.astype(np.float32)           # what torch.FloatTensor(...) makes of any other dtype
    feats = np.ascontiguousarray(feats)
    if feats.ndim != 2 or feats.shape[0] != graph.n:
        raise ValueError(f"features must be [graph nodes = {graph.n}, F], got {feats.shape}")
    m = _lib.CollateMember()
    m.n_nodes, m.n_edges = graph.n, graph.number_of_edges()
    keep = [feats]
    for name in CollatedBatch._HOST_FIELDS:
        a = np.ascontiguousarray(getattr(graph, name), dtype=np.int32)
        keep.append(a)
        setattr(m, name, a.ctypes.data)
    m.features, m.feat_bytes = feats.ctypes.data, feats.dtype.itemsize
    if labels is not None:
        lab = np.asarray(labels)
        if lab.dtype not in (np.int64, np.int32):
            lab = lab.astype(np.int64)
        lab = np.ascontiguousarray(lab).reshape(-1)
        if lab.shape[0] != graph.n:
            raise ValueError(f"labels must be [graph nodes = {graph.n}], got {lab.shape}")
        keep.append(lab)
        m.labels, m.label_bytes = lab.ctypes.data, lab.dtype.itemsize
    scheds = []
    for k, which in enumerate(kinds):
        s = graph.cluster_schedule(which)          # built once per host graph (gts_cluster_schedule), cached there
        scheds.append(s)
        if s is not None:
            m.sched_rec[k], m.sched_clusters[k], m.sched_loc_words[k] = s.rec.ctypes.data, s.n_clusters, s.loc_words
    graph._collate_member = (m, features, labels, kinds, keep, scheds, feats.shape[1])
    return m


def collate_host(samples, kinds, reserve, n_threads=N_THREADS):
    """Assemble `samples` into the block `reserve(total_bytes)` returns (anything with `.data_ptr()` or a numpy uint8
    array): the host half of HostCollator, usable without a GPU.  Returns (plan, feature width, labelled, block)."""
    lib = _lib.load()
    graphs = [s[1] for s in samples]
    if any(not isinstance(g, Graph) or isinstance(g, CollatedBatch) or g._sched_members is not None for g in graphs):
        raise TypeError("the host collate takes single host graphs (gts.Graph built from one sample)")
    if len(kinds) > _lib.COLLATE_MAX_SCHEDULES:
        raise ValueError(f"at most {_lib.COLLATE_MAX_SCHEDULES} schedule kinds per batch")
    labelled = len(samples[0]) > 3 and samples[0][3] is not None
    members = (_lib.CollateMember * len(samples))()
    width = None
    for i, s in enumerate(samples):
        members[i] = _member(s[1], s[2], s[3] if labelled else None, kinds)
        w = s[1]._collate_member[6]
        if width is None:
            width = w
        elif w != width:
            raise ValueError(f"samples disagree on the feature width ({width} vs {w})")
    ckinds = (_lib.CollateKind * max(1, len(kinds)))()
    for k, which in enumerate(kinds):
        lim = _schedule.limits(which)
        ckinds[k].max_rows, ckinds[k].max_srcs, ckinds[k].tagged = lim[0], lim[1], 1 if which == "out" else 0
    plan = _lib.CollatePlan()
    _lib.check(lib.gts_collate_plan(members, len(samples), width, ckinds, len(kinds), ctypes.byref(plan)), "gts_collate_plan")
    block = reserve(int(plan.total_bytes))
    address = block.ctypes.data if isinstance(block, np.ndarray) else block.data_ptr()
    _lib.check(lib.gts_collate_batch(members, len(samples), width, ckinds, len(kinds), address, int(plan.total_bytes),
                                     n_threads, ctypes.byref(plan)), "gts_collate_batch")
    return plan, width, labelled, block


class HostCollator:
    """samples [(id, graph, features, labels), ...] -> (ids, CollatedBatch on `device`, features fp32 [N, F] on
    `device`, labels int64 [N] on `device`), the copy enqueued on the CURRENT stream out of `ring`'s current slab.
    `kinds(n_rows)` names the cluster schedules the step will ask the graph for (GNN._schedules_wanted)."""

    def __init__(self, device, ring, kinds, n_threads=N_THREADS):
        self.device, self.ring, self.kinds, self.n_threads = device, ring, kinds, n_threads

    def __call__(self, samples):
        import weakref

        ids = [s[0] for s in samples]
        graphs = [s[1] for s in samples]
        kinds = tuple(self.kinds(sum(g.n for g in graphs)))
        plan, width, labelled, slab = collate_host(samples, kinds, self.ring.reserve, self.n_threads)
        block = slab.to(self.device, non_blocking=True)          # the ONE upload of this batch

        def view(offset, count, dtype, itemsize):
            return block[offset:offset + count * itemsize].view(dtype)

        n, e = int(plan.n_nodes), int(plan.n_edges)
        feats = view(plan.features, n * width, torch.float32, 4).view(n, width)
        labels = view(plan.labels, n, torch.int64, 8) if labelled else None
        node_off = np.zeros(len(graphs) + 1, dtype=np.int32)
        np.cumsum([g.n for g in graphs], out=node_off[1:])
        g = CollatedBatch(graphs, node_off, e, self.device)
        sizes = [n + 1, e, n + 1, e, e, e, n, n]
        d = _DeviceCSR.from_views(block, [view(plan.csr[q], sizes[q], torch.int32, 4) for q in range(8)], self.device)
        for k, which in enumerate(kinds):
            if plan.sched[k] < 0:
                g._sched[which] = None
                continue
            parts = [gr._collate_member[5][k] for gr in graphs]
            host = _UnionSchedule(parts, node_off, _schedule.limits(which), which == "out", plan.sched_loc_words[k],
                                  plan.sched_clusters[k])
            g._sched[which] = host
            words = int(plan.sched_record_words[k])
            rec = view(plan.sched[k], int(plan.sched_clusters[k]) * words, torch.int32, 4).view(-1, words)
            d.schedules[which] = _DeviceSchedule.from_view(host, rec)
        g._dev = d
        g._dev_cache[self.device] = weakref.ref(d)
        return ids, g, feats, labels
