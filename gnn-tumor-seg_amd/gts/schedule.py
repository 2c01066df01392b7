"""Cluster row schedules for the max-pool reducers (K1 / K2 at F = 256, csrc/gts_spmm_cluster.hip).

A schedule deals the rows of one CSR to workgroups as clusters of rows that share their neighbours, so that a
workgroup stages each distinct neighbour row in LDS once instead of fetching one row per edge.  It is computed
once per graph at ingest (`gts_cluster_schedule`, host C++ inside libgts_hip.so), cached on the host `Graph`,
and `gts.batch` concatenates the members' schedules with node offsets — nothing is recomputed per batch.

The schedule never changes WHAT a row computes: edges stay in CSR slot order inside every row and every output
lands in its own row, so DGL's `copy_u` / `max` semantics as called at /root/reference/model/networks.py:25,28,30
(first maximum wins, node order of `dgl.from_networkx` / `dgl.batch`) are untouched and the results are
bit-identical to the unscheduled kernels (tests/test_gpu_cluster.py).

One fixed-size int32 record per cluster, `rec[C, W]` (layout: include/gts_hip.h, gts_cluster_schedule):
  words 0..3   n_rows, n_srcs, n_edges, 0
  row ids      [max_rows]   rows of the cluster, ascending
  neighbours   [max_srcs]   distinct neighbour rows in order of first use (the tail repeats the last one)
  row info     [max_rows]   first 8-edge chunk of the row (low 16 bits) | its degree (high 16 bits)
  loc          uint8 per edge, CSR slot order, every row padded to whole chunks of 8 (pads repeat the row's last
               edge): position of the edge's neighbour in the cluster's neighbour list
  tag          uint8 per edge, same shape (K2 only): t_slot, the edge's slot in its destination's in-row
"""
import ctypes
import os

import numpy as np

from . import _lib

# (max rows, max distinct neighbour rows, max padded edges: rows take whole chunks of 8) per cluster.  One LDS ring slot (gts_cluster_lds_bytes) is
# then about 39 KiB: four slots per persistent workgroup, one workgroup per CU.
# 'gat_in' / 'gat_out': the GATConv aggregation over the in- / out-CSR (csrc/gts_gat_cluster.hip): its workgroups also hold three
# weight blocks and three epilogue-vector slots, so the neighbour list is shorter (two workgroups of 78 KB per CU).
# 'gat_edge_in': the edge pass of GATConv's backward stages the rows' own gradient slices behind the neighbours' (74 slices of
# 512 B per image).
_DEFAULT_LIMITS = {"in": (32, 76, 512), "out": (32, 60, 512), "gat_in": (32, 64, 256), "gat_out": (32, 64, 256),
                   "gat_edge_in": (24, 50, 192)}
# A schedule is used when it stages at most this share of the rows the plain kernel would fetch (one per edge).
WORTHWHILE = float(os.environ.get("GTS_CLUSTER_WORTHWHILE", "0.6"))
ENABLED = os.environ.get("GTS_CLUSTER_SPMM", "1") != "0"
# K1 takes the clustered kernel from this many rows on.  Below it the whole working set of the launch (2.25 KiB per
# row: input, output, winners) still sits in the 256 MiB Infinity Cache when the kernel starts — its input was just
# written by the GEMM in front of it — and the plain kernel's per-edge fetches are served from there faster than
# the clustered kernel stages them (in the training step at 60 000 rows: 30 us against 34 us; at 120 000 rows: 86 us
# against 72 us).  K2 is faster clustered at every size measured (60 000 rows: 46 -> 38 us).
MIN_ROWS_FORWARD = int(os.environ.get("GTS_CLUSTER_MIN_ROWS_FWD", "100000"))
# Where the clustered kernels stop paying (profiles/r03_cluster_other_graphs.log: k-nearest-neighbour graphs of random points,
# 8 x 15 000 nodes, against the plain kernels).  Their reduction costs per edge what the plain kernel's does, but runs in 24 - 32
# waves per CU that move in step with their unit's barrier, so denser graphs become compute-bound earlier:
#   K1: faster up to a mean in-degree of about 11 (degree 7.2: 65 vs 88 us; 10.4: 96 vs 107; 13.8: 135 vs 121);
#   K2: faster only while every row fits ONE 8-edge chunk (lattice, degree <= 8: 68 vs 86 us; degree 7.2 with rows of up to 15
#       edges: 101 vs 90).
MAX_MEAN_DEGREE_FORWARD = float(os.environ.get("GTS_CLUSTER_MAX_MEAN_DEGREE_FWD", "11"))
MAX_DEGREE_BACKWARD = int(os.environ.get("GTS_CLUSTER_MAX_DEGREE_BWD", "8"))
# GATConv over a schedule: rows of one 8-edge chunk, as K2 (the weighted sum costs per edge what K2's masked sum does)
ENABLED_GAT = os.environ.get("GTS_CLUSTER_GAT", "1") != "0"
MAX_DEGREE_GAT = int(os.environ.get("GTS_CLUSTER_MAX_DEGREE_GAT", "8"))
MIN_ROWS_GAT = int(os.environ.get("GTS_CLUSTER_MIN_ROWS_GAT", "20000"))


def limits(which):
    """Cluster limits for 'in' (K1) / 'out' (K2); GTS_CLUSTER_LIMITS="r,s,e;r,s,e" overrides (tuning runs)."""
    env = os.environ.get("GTS_GAT_CLUSTER_LIMITS" if which.startswith("gat_") else "GTS_CLUSTER_LIMITS")
    if env:
        both = [tuple(int(v) for v in part.split(",")) for part in env.split(";")]
        return both[2 if which == "gat_edge_in" else 0 if which.endswith("in") else 1]
    return _DEFAULT_LIMITS[which]


def _pad4(n):
    return (int(n) + 3) & ~3


def _i32p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class _Layout:
    """Word offsets of the record sections (mirror of rec_layout in csrc/gts_spmm_cluster.hip)."""

    def __init__(self, max_rows, max_srcs, loc_words, tagged):
        self.rows = 4
        self.srcs = self.rows + _pad4(max_rows)
        self.eoff = self.srcs + _pad4(max_srcs)
        self.loc = self.eoff + _pad4(max_rows)
        self.tag = self.loc + loc_words
        self.words = self.tag + (loc_words if tagged else 0)


class ClusterSchedule:
    __slots__ = ("limits", "tagged", "loc_words", "n_rows", "n_edges", "staged_rows", "rec", "owner")

    def __init__(self, lim, tagged, loc_words, n_rows, n_edges, staged_rows, rec, owner=None):
        self.owner = owner               # the (page-locked) torch tensor `rec` is a view of, when concat() staged it for upload
        self.limits = tuple(int(v) for v in lim)
        self.tagged, self.loc_words = bool(tagged), int(loc_words)
        self.n_rows, self.n_edges, self.staged_rows = int(n_rows), int(n_edges), int(staged_rows)
        self.rec = rec if (isinstance(rec, np.ndarray) and rec.dtype == np.int32 and rec.flags.c_contiguous) \
            else np.ascontiguousarray(rec, dtype=np.int32)
        assert self.rec.ndim == 2 and self.rec.shape[1] == self.layout.words

    @property
    def n_clusters(self):
        return self.rec.shape[0]

    @property
    def layout(self):
        return _Layout(self.limits[0], self.limits[1], self.loc_words, self.tagged)

    def worthwhile(self, max_degree=None, which=None):
        """Does the clustered kernel beat the plain one on this graph?  It must stage clearly fewer neighbour rows than
        the plain kernel fetches (one per edge), and the graph must be sparse enough for its reduction (see above)."""
        if self.n_edges == 0 or self.staged_rows > WORTHWHILE * self.n_edges:
            return False
        if which is not None and which.startswith("gat_"):
            return max_degree is None or max_degree <= MAX_DEGREE_GAT
        if self.tagged:      # K2
            return max_degree is None or max_degree <= MAX_DEGREE_BACKWARD
        return self.n_edges <= MAX_MEAN_DEGREE_FORWARD * self.n_rows

    def with_loc_words(self, loc_words):
        """The same schedule with the per-edge sections re-packed to `loc_words` words (>= what the edges need)."""
        if loc_words == self.loc_words:
            return self
        old = self.layout
        new = _Layout(self.limits[0], self.limits[1], loc_words, self.tagged)
        rec = np.zeros((self.n_clusters, new.words), dtype=np.int32)
        rec[:, :old.loc] = self.rec[:, :old.loc]
        keep = min(loc_words, self.loc_words)
        rec[:, new.loc:new.loc + keep] = self.rec[:, old.loc:old.loc + keep]
        if self.tagged:
            rec[:, new.tag:new.tag + keep] = self.rec[:, old.tag:old.tag + keep]
        return ClusterSchedule(self.limits, self.tagged, loc_words, self.n_rows, self.n_edges, self.staged_rows, rec)

    @staticmethod
    def build(indptr, indices, t_indptr, t_indices, edge_tag, lim):
        """Schedule of the CSR (indptr, indices); (t_indptr, t_indices) is its transpose (row lists of every
        neighbour).  Returns None when a row does not fit a cluster (degree above the limits) or a tag does not
        fit a byte."""
        lib = _lib.load()
        if not (1 <= lim[0] and 1 <= lim[1] <= 256 and 1 <= lim[2] <= 65535):
            raise ValueError(f"cluster limits {tuple(lim)}: neighbours are byte-indexed (<= 256), edge offsets 16-bit")
        n, e = int(indptr.size - 1), int(indices.size)
        arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in (indptr, indices, t_indptr, t_indices)]
        tag = None if edge_tag is None else np.ascontiguousarray(edge_tag, dtype=np.int32)
        lw_max = _pad4((lim[2] + 3) // 4)
        words = int(lib.gts_cluster_record_words(int(lim[0]), int(lim[1]), lw_max, 0 if tag is None else 1))
        if words > 512:
            raise ValueError(f"cluster limits {tuple(lim)}: records of {words} words (at most 512)")
        capacity = 2 * n // max(1, int(lim[0])) + 64        # clusters are nearly full on real graphs; retried otherwise
        while True:
            rec = np.empty((capacity, words), dtype=np.int32)
            n_cl, staged, e_max = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int32(0)
            code = lib.gts_cluster_schedule(*map(_i32p, arrs), _i32p(tag), n, int(lim[0]), int(lim[1]), int(lim[2]),
                                            _i32p(rec), capacity, ctypes.byref(n_cl), ctypes.byref(staged),
                                            ctypes.byref(e_max))
            if code == -2:      # GTS_ERR_SHAPE: a row's degree (or a tag) is beyond what a cluster can hold
                return None
            _lib.check(code, "gts_cluster_schedule")
            if n_cl.value <= capacity:
                break
            capacity = int(n_cl.value)
        full = ClusterSchedule(lim, tag is not None, lw_max, n, e, staged.value, rec[:n_cl.value])
        return full.with_loc_words(_pad4((int(e_max.value) + 3) // 4))

    @staticmethod
    def concat(parts, node_offsets):
        """Schedule of the block-diagonal union (gts.batch): member clusters in member order, ids shifted."""
        lim, tagged = parts[0].limits, parts[0].tagged
        if any(p.limits != lim or p.tagged != tagged for p in parts):
            raise ValueError("schedules built with different limits cannot be concatenated")
        from .graph import host_staging_int32

        lw = max(p.loc_words for p in parts)
        parts = [p.with_loc_words(lw) for p in parts]
        lay = parts[0].layout
        counts = [p.n_clusters for p in parts]
        host, owner = host_staging_int32(sum(counts) * lay.words)      # page-locked: one asynchronous upload
        rec = host[:sum(counts) * lay.words].reshape(sum(counts), lay.words)
        np.concatenate([p.rec for p in parts], out=rec)
        shift = np.repeat(np.asarray(node_offsets[:len(parts)], dtype=np.int32), counts)
        rec[:, lay.rows:lay.eoff] += shift[:, None]                   # row ids and neighbour ids of every cluster
        return ClusterSchedule(lim, tagged, lw, sum(p.n_rows for p in parts), sum(p.n_edges for p in parts),
                               sum(p.staged_rows for p in parts), rec,
                               owner=owner[:rec.size] if owner is not None else None)

    def lds_bytes(self, kind):
        return int(_lib.load().gts_cluster_lds_bytes(self.limits[0], self.limits[1], self.loc_words, kind))

    # ---- host-side view used by the tests (the schedule is an exact cover, edges in CSR order)
    def decode(self):
        """[(rows, neighbours, [(row, [neighbour ids in slot order], [tags] or None) ...])] per cluster."""
        lay, out = self.layout, []
        for r in self.rec:
            n_rows, n_srcs, n_edges = int(r[0]), int(r[1]), int(r[2])
            rows = r[lay.rows:lay.rows + n_rows]
            srcs = r[lay.srcs:lay.srcs + n_srcs]
            info = r[lay.eoff:lay.eoff + n_rows].view(np.uint32)
            loc = r[lay.loc:lay.loc + self.loc_words].view(np.uint8)
            tag = r[lay.tag:lay.tag + self.loc_words].view(np.uint8) if self.tagged else None
            per_row, at = [], 0
            for i, row in enumerate(rows):
                first, deg = 8 * int(info[i] & 0xFFFF), int(info[i] >> 16)
                assert first == at, "rows take whole 8-edge chunks, back to back"
                at += (deg + 7) & ~7
                assert all(loc[k] == loc[first + deg - 1] for k in range(first + deg, at)), "pads repeat the last edge"
                ks = range(first, first + deg)
                per_row.append((int(row), [int(srcs[loc[k]]) for k in ks], None if tag is None else [int(tag[k]) for k in ks]))
            assert at == n_edges
            out.append((rows, srcs, per_row))
        return out
