"""Graph container that crosses the host/device boundary of the GNN path.

Stands in for the DGLGraph objects produced by data_processing/data_loader.py:72
(`dgl.from_networkx`) and :168 (`dgl.batch`) of the reference and consumed by
model/networks.py:32-36,60-66 (`layer(graph, h)`).  Duck-typed surface kept for the
callers: `.to(device)`, `.in_degrees()`, `.number_of_edges()`, `.number_of_nodes()`,
`.ndata[...]`, `.batch_num_nodes()`, plus the constructors `from_networkx` / `batch`.

Layout (all int32, built once on the host, uploaded once per device):
  COO           src[e], dst[e]           edges in insertion order (DGL edge ids)
  in-CSR        indptr[N+1], indices[E]  rows = destinations; a row lists its sources in COO
                                         order (stable sort on dst == DGL's COO->CSC)
  out-CSR       t_indptr, t_indices      rows = sources, entries = destinations (stable on src)
  t_slot[e']    position of out-CSR edge e' inside its destination's in-CSR row
  t_pos[e']     absolute in-CSR position of out-CSR edge e'  (= indptr[dst] + t_slot)
The max-pool argmax is stored as a slot (uint8 when max in-degree <= 254), which is why
the backward needs t_slot; GAT's backward reads per-edge data through t_pos.
"""
import weakref

import numpy as np
import torch

_I32_MAX = 2 ** 31 - 1
import threading

_uploads = threading.local()


def host_staging_int32(n_words):
    """(`n_words` int32 of host memory as a numpy array, the torch tensor that shares it): what batch() and the
    schedule concatenation assemble in upload layout, so that it goes to the device in ONE copy without re-packing."""
    a = np.empty(max(int(n_words), 4), dtype=np.int32)
    return a, torch.from_numpy(a)


class PinnedRing:
    """A few page-locked slabs allocated ONCE (hipHostMalloc / hipHostFree cost milliseconds and synchronise the
    device: never per batch) through which a loader thread stages its uploads: `upload(t, device)` copies the host
    tensor into the current slab and enqueues an asynchronous host-to-device copy from there on the current stream.
    `next_batch()` moves on to the next slab, first waiting for the copies that last read it."""

    def __init__(self, slabs=3, nbytes=8 << 20):
        self.slabs = [torch.empty(nbytes, dtype=torch.uint8, pin_memory=True) for _ in range(slabs)]
        self.events = [None] * slabs
        self.cur, self.at = 0, 0
        self.wanted = 0       # bytes the batches so far would have needed in one slab (uploads that did not fit included)

    def next_batch(self):
        self.events[self.cur] = torch.cuda.Event()
        self.events[self.cur].record()
        self.cur, self.at = (self.cur + 1) % len(self.slabs), 0
        if self.events[self.cur] is not None:
            self.events[self.cur].synchronize()
        if self.wanted > self.slabs[self.cur].numel():
            # a batch outgrew the slabs (a C2 batch: 7.4 MB of features, labels and CSR, then 1.9 MB of schedule records):
            # its tail went up as plain pageable copies, which block the loader thread for milliseconds.  Grow this slab
            # once, with room to spare (the slab is idle here: its last copies have completed).
            self.slabs[self.cur] = torch.empty(self.wanted + self.wanted // 2, dtype=torch.uint8, pin_memory=True)

    def reserve(self, nbytes):
        """`nbytes` of the current slab as a uint8 tensor for the caller to fill in place (gts.collate: the C collate
        writes a whole batch there and uploads it with one copy).  A slab that is too small is replaced once, with
        room to spare, when it is still untouched; otherwise the block gets page-locked memory of its own."""
        start = (self.at + 255) & ~255
        if start + nbytes > self.slabs[self.cur].numel():
            if self.at != 0:
                return torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
            self.slabs[self.cur] = torch.empty(nbytes + nbytes // 2, dtype=torch.uint8, pin_memory=True)
            start = 0
        self.at = start + nbytes
        self.wanted = max(self.wanted, self.at)
        return self.slabs[self.cur][start:start + nbytes]

    def upload(self, host, device):
        nbytes = host.numel() * host.element_size()
        start = (self.at + 255) & ~255
        if start + nbytes > self.slabs[self.cur].numel():
            if self.at == 0:      # larger than a whole slab: grow this one (rare; the old block is freed by torch when idle)
                self.slabs[self.cur] = torch.empty(max(2 * nbytes, 1 << 20), dtype=torch.uint8, pin_memory=True)
                start = 0
            else:                 # slab full: plain copy for the rest of this batch (next_batch() grows the slabs)
                self.at = start + nbytes
                self.wanted = max(self.wanted, self.at)
                return host.to(device)
        stage = self.slabs[self.cur][start:start + nbytes].view(host.dtype).view(host.shape)
        # a plain memcpy: torch's copy_ would start an OpenMP team in the loader thread (a new team per thread, spinning
        # beside the main thread: 47 ms per batch measured), numpy copies in the calling thread
        np.copyto(stage.numpy(), host.numpy())
        self.at = start + nbytes
        self.wanted = max(self.wanted, self.at)
        return stage.to(device, non_blocking=True)


def uploads_through(ring):
    """Route the uploads this THREAD makes from now on (graph CSRs, cluster schedules) through `ring` (None: plain copies)."""
    _uploads.ring = ring


def _upload(host_tensor, device):
    ring = getattr(_uploads, "ring", None)
    return ring.upload(host_tensor, device) if ring is not None else host_tensor.to(device)


def _pad4(n):
    return (int(n) + 3) & ~3


class _DeviceCSR:
    """Device-resident int32 arrays + degree vectors of one Graph, uploaded as ONE packed buffer
    (a single host->device copy instead of eight) and handed out as views of it."""

    __slots__ = ("indptr", "indices", "t_indptr", "t_indices", "t_slot", "t_pos",
                 "deg_clamped", "deg_plus1", "device", "packed", "schedules", "__weakref__")
    _INT_FIELDS = ("indptr", "indices", "t_indptr", "t_indices", "t_slot", "t_pos")

    def __init__(self, g, device):
        staged = getattr(g, "_staged", None)
        if staged is not None:            # batch() assembled the arrays in upload layout already (page-locked memory)
            host_tensor, offsets, sizes = staged
        else:
            deg = np.diff(g.indptr).astype(np.float32)
            parts = [np.ascontiguousarray(getattr(g, name), dtype=np.int32) for name in self._INT_FIELDS]
            parts += [np.maximum(deg, np.float32(1)).view(np.int32), (deg + np.float32(1)).view(np.int32)]
            offsets, at = [], 0
            for a in parts:                                  # every segment starts 16-byte aligned
                offsets.append(at)
                at += _pad4(a.size)
            host, host_tensor = host_staging_int32(at)
            for a, off in zip(parts, offsets):
                host[off:off + a.size] = a
            sizes = [a.size for a in parts]
        self.device = device
        self.schedules = {}               # 'in' / 'out' -> _DeviceSchedule (cluster row schedules, uploaded on first use)
        self.packed = _upload(host_tensor, device)
        views = [self.packed[off:off + size] for size, off in zip(sizes, offsets)]
        for name, v in zip(self._INT_FIELDS, views):
            setattr(self, name, v)
        self.deg_clamped = views[6].view(torch.float32)
        self.deg_plus1 = views[7].view(torch.float32)

    @classmethod
    def from_views(cls, packed, views, device):
        """Device CSR over arrays that are on the device already (gts.collate: views of one uploaded block, in
        _INT_FIELDS order, then max(deg, 1) and deg + 1 as fp32 bits)."""
        self = cls.__new__(cls)
        self.device, self.schedules, self.packed = device, {}, packed
        for name, v in zip(cls._INT_FIELDS, views):
            setattr(self, name, v)
        self.deg_clamped = views[6].view(torch.float32)
        self.deg_plus1 = views[7].view(torch.float32)
        return self

    def record_stream(self, stream):
        """The buffer was uploaded on another stream than the one that will read it."""
        self.packed.record_stream(stream)
        for s in self.schedules.values():
            s.packed.record_stream(stream)


class _DeviceSchedule:
    """Device copy of one ClusterSchedule (gts/schedule.py): its records, one upload."""

    __slots__ = ("host", "packed")

    def __init__(self, sched, device):
        self.host = sched
        owner = sched.owner if sched.owner is not None else torch.from_numpy(sched.rec)
        self.packed = _upload(owner, device).view(sched.rec.shape)

    @classmethod
    def from_view(cls, sched, records):
        """Records that are on the device already (gts.collate); `sched` describes them (n_clusters, limits, ...)."""
        self = cls.__new__(cls)
        self.host, self.packed = sched, records
        return self


class Graph:
    def __init__(self, src, dst, num_nodes, batch_num_nodes=None, _prebuilt=None, _members=None):
        self.n = int(num_nodes)
        self._members = _members          # batch(): COO is assembled from the members on first use
        self._sched_members = _members    # batch(): cluster schedules are concatenated from the members' on first use
        self._sched = {}                  # 'in' / 'out' -> ClusterSchedule or None (shared by every view)
        if _members is None:
            self._src = np.ascontiguousarray(src, dtype=np.int32)
            self._dst = np.ascontiguousarray(dst, dtype=np.int32)
            if self._src.shape != self._dst.shape or self._src.ndim != 1:
                raise ValueError("src/dst must be 1-D arrays of equal length")
            if self.n > _I32_MAX or self._src.size > _I32_MAX:
                raise ValueError("graph too large for int32 CSR")
            if self._src.size and (min(self._src.min(), self._dst.min()) < 0 or
                                   max(self._src.max(), self._dst.max()) >= self.n):
                raise ValueError("edge endpoint outside [0, num_nodes)")
        else:
            self._src = self._dst = None  # members were validated when they were built
        self._batch_num_nodes = list(batch_num_nodes) if batch_num_nodes is not None else [self.n]
        self.ndata = {}
        self.device = torch.device("cpu")
        self._dev_cache = {}              # device -> weakref(_DeviceCSR), shared by every view of this graph
        self._dev = None                  # the strong reference lives on the device view that uses it
        if _prebuilt is not None:
            (self.indptr, self.indices, self.t_indptr, self.t_indices, self.t_slot,
             self.t_pos) = _prebuilt
        else:
            self._build_csr()
        deg = np.diff(self.indptr)
        self.max_in_degree = int(deg.max()) if deg.size else 0
        self.min_in_degree = int(deg.min()) if deg.size else 0
        self._max_out_degree = None

    def _coo(self):
        if self._src is None:
            graphs, node_off = self._members
            self._src = np.concatenate([g.src + node_off[i] for i, g in enumerate(graphs)]).astype(np.int32, copy=False)
            self._dst = np.concatenate([g.dst + node_off[i] for i, g in enumerate(graphs)]).astype(np.int32, copy=False)
            self._members = None
        return self._src, self._dst

    @property
    def src(self):
        """COO sources in edge order (int32)."""
        return self._coo()[0]

    @property
    def dst(self):
        return self._coo()[1]

    # ------------------------------------------------------------------ construction
    def _build_csr(self):
        n, e = self.n, self.src.size
        order = np.argsort(self.dst, kind="stable")
        self.indices = np.ascontiguousarray(self.src[order])
        counts = np.bincount(self.dst, minlength=n).astype(np.int64)
        self.indptr = np.zeros(n + 1, dtype=np.int32)
        np.cumsum(counts, out=self.indptr[1:])
        # absolute in-CSR position of every COO edge
        pos_of_edge = np.empty(e, dtype=np.int32)
        pos_of_edge[order] = np.arange(e, dtype=np.int32)
        torder = np.argsort(self.src, kind="stable")
        self.t_indices = np.ascontiguousarray(self.dst[torder])
        tcounts = np.bincount(self.src, minlength=n).astype(np.int64)
        self.t_indptr = np.zeros(n + 1, dtype=np.int32)
        np.cumsum(tcounts, out=self.t_indptr[1:])
        self.t_pos = np.ascontiguousarray(pos_of_edge[torder])
        self.t_slot = (self.t_pos - self.indptr[self.t_indices]).astype(np.int32)

    # ------------------------------------------------------------------ DGL-like surface
    def number_of_nodes(self):
        return self.n

    num_nodes = number_of_nodes

    def number_of_edges(self):
        return int(self.indices.size)

    num_edges = number_of_edges

    def batch_num_nodes(self):
        return torch.tensor(self._batch_num_nodes, dtype=torch.int64)

    @property
    def batch_size(self):
        return len(self._batch_num_nodes)

    def in_degrees(self):
        return torch.from_numpy(np.diff(self.indptr).astype(np.int64)).to(self.device)

    def out_degrees(self):
        return torch.from_numpy(np.diff(self.t_indptr).astype(np.int64)).to(self.device)

    def edges(self):
        return (torch.from_numpy(self.src.astype(np.int64)).to(self.device),
                torch.from_numpy(self.dst.astype(np.int64)).to(self.device))

    def to(self, device, **_kwargs):
        """Return a view of this graph on `device`.  Host arrays are shared.  The device copy of the
        CSR is uploaded on first use and shared by every live view on that device, but it is owned
        by those views (the host graph only keeps a weak reference): a dataset that caches host
        graphs does not pin ~3 MB of device memory per sample it has ever evaluated."""
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        out = type(self).__new__(type(self))
        out.__dict__.update(self.__dict__)
        out.device = device
        out._dev = None
        out.ndata = {k: v.to(device) for k, v in self.ndata.items()}
        return out

    def dev(self):
        """Device CSR for self.device (uploaded lazily; kept alive by the views that use it)."""
        if self.device.type != "cuda":
            raise RuntimeError("Graph is on the CPU: call graph.to('cuda') first "
                               "(the gts operators have no CPU path)")
        d = self._dev
        if d is None or d.device != self.device:
            ref = self._dev_cache.get(self.device)
            d = ref() if ref is not None else None
            if d is None:
                d = _DeviceCSR(self, self.device)
                self._dev_cache[self.device] = weakref.ref(d)
            self._dev = d
        return d

    @property
    def max_out_degree(self):
        if self._max_out_degree is None:
            deg = np.diff(self.t_indptr)
            self._max_out_degree = int(deg.max()) if deg.size else 0
        return self._max_out_degree

    @property
    def arg_bytes(self):
        return 1 if self.max_in_degree <= 254 else 4

    def cluster_schedule(self, which):
        """Cluster row schedule of the in-CSR ('in': K1, rows = destinations; 'gat_in': GATConv forward) or of the out-CSR
        ('out': K2, rows = sources, tagged with t_slot; 'gat_out': GATConv backward); None when clustering does not save enough row fetches on this graph (or a
        row's degree is beyond a cluster).  Built once per host graph; a batch concatenates its members'."""
        from . import schedule as _schedule

        if which not in self._sched:
            sched = None
            if self._sched_members is not None:
                graphs, node_off = self._sched_members
                parts = [g.cluster_schedule(which) for g in graphs]
                if all(p is not None for p in parts):
                    sched = _schedule.ClusterSchedule.concat(parts, node_off)
            else:
                lim = _schedule.limits(which)
                if which.endswith("in"):
                    sched = _schedule.ClusterSchedule.build(self.indptr, self.indices, self.t_indptr, self.t_indices,
                                                            None, lim)
                else:       # 'out' carries K2's tags (t_slot); 'gat_out' has no use for them
                    sched = _schedule.ClusterSchedule.build(self.t_indptr, self.t_indices, self.indptr, self.indices,
                                                            self.t_slot if which == "out" else None, lim)
                if sched is not None:
                    degrees = np.diff(self.indptr if which.endswith("in") else self.t_indptr)
                    if not sched.worthwhile(int(degrees.max()) if degrees.size else 0, which):
                        sched = None
            self._sched[which] = sched
        return self._sched[which]

    def dev_schedule(self, which):
        """Device copy of cluster_schedule(which) on self.device (None when there is no schedule)."""
        sched = self.cluster_schedule(which)
        if sched is None:
            return None
        d = self.dev()
        held = d.schedules.get(which)
        if held is None or held.host is not sched:
            held = _DeviceSchedule(sched, self.device)
            d.schedules[which] = held
        return held

    def __repr__(self):
        return (f"Graph(num_nodes={self.n}, num_edges={self.number_of_edges()}, "
                f"batch_size={self.batch_size}, device={self.device})")


def graph(data, num_nodes=None):
    """graph((src, dst), num_nodes) — COO constructor (dgl.graph counterpart)."""
    src, dst = data
    src = np.asarray(src)
    dst = np.asarray(dst)
    if num_nodes is None:
        num_nodes = int(max(src.max(initial=-1), dst.max(initial=-1)) + 1)
    return Graph(src, dst, num_nodes)


def from_networkx(nx_graph):
    """Counterpart of dgl.from_networkx as used at data_processing/data_loader.py:72.

    Nodes are relabelled to 0..N-1 in sorted order, an undirected graph contributes both
    directions of every pair (a self-loop once), and the COO edge order is that of
    `to_directed().edges()` of the relabelled graph: sources in node-iteration order (=
    ascending when the nodes were inserted in sorted order, as mri2graph/graphgen.py does),
    targets in adjacency order.  Edge and node attributes are not carried over (the
    reference passes none)."""
    import networkx as nx

    g = nx.convert_node_labels_to_integers(nx_graph, ordering="sorted")
    if not g.is_directed():
        g = g.to_directed()
    e = g.number_of_edges()
    coo = np.fromiter((x for uv in g.edges() for x in uv), dtype=np.int64, count=2 * e)
    coo = coo.reshape(e, 2)
    return Graph(coo[:, 0], coo[:, 1], g.number_of_nodes())


def batch(graphs):
    """Counterpart of dgl.batch (data_processing/data_loader.py:168): block-diagonal union,
    node ids of graph j shifted by sum_{i<j} N_i, edges concatenated in graph order.
    Both CSRs of the union are the shifted concatenations of the members' CSRs (a stable
    sort of the concatenated COO gives exactly that), so nothing is re-sorted."""
    graphs = list(graphs)
    if not graphs:
        raise ValueError("batch() needs at least one graph")
    node_off = np.cumsum([0] + [g.n for g in graphs])
    edge_off = np.cumsum([0] + [g.number_of_edges() for g in graphs])
    if node_off[-1] > _I32_MAX or edge_off[-1] > _I32_MAX:
        raise ValueError("batched graph too large for int32 CSR")
    no32 = node_off.astype(np.int32)
    eo32 = edge_off.astype(np.int32)
    n_total, e_total = int(node_off[-1]), int(edge_off[-1])

    # ONE host buffer in the layout _DeviceCSR uploads (indptr | indices | t_indptr | t_indices | t_slot | t_pos |
    # max(deg, 1) | deg + 1 as fp32 bits; every segment 16-byte aligned); every member slice is written (shifted)
    # straight into its place and the Graph's arrays are views of it: no second packing pass before the upload
    seg_sizes = [n_total + 1, e_total, n_total + 1, e_total, e_total, e_total, n_total, n_total]
    seg_at, at = [], 0
    for size in seg_sizes:
        seg_at.append(at)
        at += _pad4(size)
    host, owner = host_staging_int32(at)
    seg = [host[o:o + size] for o, size in zip(seg_at, seg_sizes)]

    def cat_ptr(name, out):
        for i, g in enumerate(graphs):
            np.add(getattr(g, name)[:-1], eo32[i], out=out[node_off[i]:node_off[i + 1]])
        out[n_total] = eo32[-1]
        return out

    def cat_edges(name, shift, out):
        for i, g in enumerate(graphs):
            part = out[edge_off[i]:edge_off[i + 1]]
            if shift is None:
                part[:] = getattr(g, name)
            else:
                np.add(getattr(g, name), shift[i], out=part)
        return out

    prebuilt = (cat_ptr("indptr", seg[0]), cat_edges("indices", no32, seg[1]), cat_ptr("t_indptr", seg[2]),
                cat_edges("t_indices", no32, seg[3]), cat_edges("t_slot", None, seg[4]), cat_edges("t_pos", eo32, seg[5]))
    deg = np.subtract(seg[0][1:], seg[0][:-1]).astype(np.float32)
    np.maximum(deg, np.float32(1), out=seg[6].view(np.float32))
    np.add(deg, np.float32(1), out=seg[7].view(np.float32))
    sizes = [s for g in graphs for s in g._batch_num_nodes]
    out = Graph(None, None, n_total, batch_num_nodes=sizes, _prebuilt=prebuilt, _members=(graphs, no32))
    out._staged = (owner, seg_at, seg_sizes)
    common = set(graphs[0].ndata)
    for g in graphs[1:]:
        common &= set(g.ndata)
    for k in common:
        out.ndata[k] = torch.cat([g.ndata[k] for g in graphs], dim=0)
    return out
