"""Offline model of what the clustered GAT aggregation fetches through ONE XCD's L2 (tools/diag: not part of the product).

Replays the unit order of gat_cluster_stream_kernel (csrc/gts_gat_cluster.hip) over the 'gat_in' schedule of the C3 batch
(4 lattice graphs of 15 000 nodes) through a 4 MiB / 16-way / 128-byte-line LRU cache and counts the lines that miss, for
  * the table layouts: row-major [N, H, 256] (4 KiB rows, a unit's slice = 512 bytes of every row) and slice-major
    [2 H][N][128] (a slice column is contiguous);
  * the set index: plain address bits, or a hash of all line-address bits (the unknown of the real L2);
  * the walk: `group` clusters through all their slices (0 = the XCD's whole span slice by slice).
Units are replayed in issue order (the 64 persistent workgroups of an XCD take consecutive units), stores as write-allocate
lines of the output rows unless --nt.
Usage: python tools/diag/l2_halo_sim.py [--kind gat_in] [--graphs 4] [--cache-mb 4]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "gnn-tumor-seg_amd"))


def build(kind, graphs, limits=None):
    from gts import synth, schedule, graph as G
    g = synth.lattice_graph()
    lim = limits or schedule.limits(kind)
    if kind.endswith("out"):
        s = schedule.ClusterSchedule.build(g.t_indptr, g.t_indices, g.indptr, g.indices, None, lim)
    else:
        s = schedule.ClusterSchedule.build(g.indptr, g.indices, g.t_indptr, g.t_indices, None, lim)
    lay = s.layout
    rows, srcs = [], []
    for m in range(graphs):
        for r in s.rec:
            rows.append(r[lay.rows:lay.rows + r[0]] + m * g.n)
            srcs.append(r[lay.srcs:lay.srcs + r[1]] + m * g.n)
    return rows, srcs, g.n * graphs


class Cache:
    def __init__(self, mbytes, ways=16, line=128, hashed=False):
        self.sets = int(mbytes * (1 << 20)) // (ways * line)
        self.ways, self.hashed = ways, hashed
        self.tags = [dict() for _ in range(self.sets)]   # insertion-ordered dict as LRU
        self.miss = self.hit = 0

    def index(self, line_addr):
        if self.hashed:
            x = (line_addr * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
            return (x >> 40) % self.sets
        return line_addr % self.sets

    def touch(self, line_addr, count=True):
        s = self.tags[self.index(line_addr)]
        if line_addr in s:
            del s[line_addr]
            s[line_addr] = 1
            if count:
                self.hit += 1
            return
        if count:
            self.miss += 1
        if len(s) >= self.ways:
            del s[next(iter(s))]
        s[line_addr] = 1


def unit_order(span, subs, group, per_xcd):
    """(cluster offset in span, sub) in global issue order i = 0 .. span*subs-1 (the kernel's `unit` lambda)."""
    group = group if 0 < group < span else span
    out = []
    for i in range(span * subs):
        per_group = group * subs
        gi, r = divmod(i, per_group)
        first = gi * group
        size = min(group, span - first)
        s, c = divmod(r, size)
        out.append((first + c, s))
    return out


def simulate(rows, srcs, n, heads, layout, hashed, group, cache_mb, nt, xcd=3, slice_bytes=512, out_base=1 << 34, pitch=None):
    ncl = len(rows)
    clo, chi = ncl * xcd // 8, ncl * (xcd + 1) // 8
    subs = heads * (1024 // slice_bytes)
    row_bytes = pitch or heads * 1024
    cache = Cache(cache_mb, hashed=hashed)
    lines = slice_bytes // 128
    for c, s in unit_order(chi - clo, subs, group, 64):
        for node in srcs[clo + c]:
            a = (s * n * slice_bytes + int(node) * slice_bytes) if layout == "slice" else (int(node) * row_bytes + s * slice_bytes)
            for l in range(lines):
                cache.touch(a // 128 + l)
        if not nt:
            for node in rows[clo + c]:
                a = out_base + int(node) * row_bytes + s * slice_bytes
                for l in range(lines):
                    cache.touch(a // 128 + l, count=False)
    span_rows = sum(len(rows[c]) for c in range(clo, chi))
    compulsory = span_rows * heads * 1024
    return cache.miss * 128 / compulsory, (cache.miss + cache.hit) * 128 / compulsory


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="gat_in")
    ap.add_argument("--graphs", type=int, default=4)
    ap.add_argument("--heads", type=int, default=4)
    ap.add_argument("--cache-mb", type=float, default=4.0)
    ap.add_argument("--limits", default=None)
    ap.add_argument("--slice-bytes", type=int, default=512)
    ap.add_argument("--pitch", type=int, default=None, help="row pitch in bytes of the row-major table (default heads KiB)")
    ap.add_argument("--only", default=None, help="layout,hashed,group,nt of the one case to run")
    args = ap.parse_args()
    lim = tuple(int(v) for v in args.limits.split(",")) if args.limits else None
    rows, srcs, n = build(args.kind, args.graphs, lim)
    staged = sum(len(s) for s in srcs) / sum(len(r) for r in rows)
    print(f"{len(rows)} clusters, staged rows per row {staged:.3f}", flush=True)
    if args.only:
        layout, hashed, group, nt = args.only.split(",")
        miss, total = simulate(rows, srcs, n, args.heads, layout, hashed == "1", int(group), args.cache_mb, nt == "1", slice_bytes=args.slice_bytes, pitch=args.pitch)
        print(f"layout {layout} hashed {hashed} group {group} nt {nt} pitch {args.pitch}: fetched / compulsory = {miss:.3f} (staged {total:.3f})")
        sys.exit(0)
    for layout in ("row", "slice"):
        for hashed in (False, True):
            for group in (0, 16, 64):
                for nt in (True, False):
                    miss, total = simulate(rows, srcs, n, args.heads, layout, hashed, group, args.cache_mb, nt, slice_bytes=args.slice_bytes)
                    print(f"layout {layout:5s} hashed {int(hashed)} group {group:3d} nt {int(nt)}: fetched / compulsory = {miss:.3f} (staged {total:.3f})", flush=True)


def simulate_drift(rows, srcs, n, heads, group, cache_mb, sigma, dynamic, xcd=3, slice_bytes=512, wgs=64, seed=0, jitter=0.1):
    """Event-driven replay: `wgs` persistent workgroups with their own speeds (1 + sigma * N(0,1), fixed per workgroup) and
    per-unit jitter; static dealing (workgroup j takes units j, j + wgs, ...) or dynamic (the next free unit)."""
    import heapq
    rng = np.random.default_rng(seed)
    ncl = len(rows)
    clo, chi = ncl * xcd // 8, ncl * (xcd + 1) // 8
    subs = heads * (1024 // slice_bytes)
    order = unit_order(chi - clo, subs, group, wgs)
    speed = 1.0 + sigma * rng.standard_normal(wgs)
    cache = Cache(cache_mb, hashed=True)
    lines = slice_bytes // 128
    nxt = [j for j in range(wgs)]          # static: next unit index of workgroup j
    counter = 0
    heap = [(0.0, j) for j in range(wgs)]
    heapq.heapify(heap)
    while heap:
        t, j = heapq.heappop(heap)
        if dynamic:
            i = counter
            counter += 1
        else:
            i = nxt[j]
            nxt[j] += wgs
        if i >= len(order):
            continue
        c, s = order[i]
        for node in srcs[clo + c]:
            a = int(node) * heads * 1024 + s * slice_bytes
            for l in range(lines):
                cache.touch(a // 128 + l)
        heapq.heappush(heap, (t + speed[j] * (1.0 + jitter * rng.standard_normal()), j))
    span_rows = sum(len(rows[c]) for c in range(clo, chi))
    return cache.miss * 128 / (span_rows * heads * 1024)


def simulate_pending(rows, srcs, n, heads, group, cache_mb, latency, chunk=1, xcd=3, slice_bytes=512, wgs=64, merge=False):
    """Units in lockstep generations of `wgs`; a line requested in generation g is usable from generation g + latency on.
    merge = False: a request for a line that is still in flight goes to the fabric again (no miss merging in the L2).
    chunk: consecutive units one workgroup takes back to back (1 = round-robin dealing)."""
    ncl = len(rows)
    clo, chi = ncl * xcd // 8, ncl * (xcd + 1) // 8
    subs = heads * (1024 // slice_bytes)
    order = unit_order(chi - clo, subs, group, wgs)
    cache = Cache(cache_mb, hashed=True)
    lines = slice_bytes // 128
    pending = {}                      # line -> generation it lands in
    fetched = 0
    # generation g: workgroup j runs unit index (g // chunk) * wgs * chunk + j * chunk + g % chunk
    n_gen = (len(order) + wgs - 1) // wgs + chunk
    for g in range(n_gen):
        landed = [l for l, when in pending.items() if when <= g]
        for l in landed:
            del pending[l]
            cache.touch(l, count=False)
        for j in range(wgs):
            i = (g // chunk) * wgs * chunk + j * chunk + g % chunk
            if i >= len(order):
                continue
            c, s = order[i]
            for node in srcs[clo + c]:
                a = (int(node) * heads * 1024 + s * slice_bytes) // 128
                for l in range(a, a + lines):
                    if l in pending:
                        if not merge:
                            fetched += 1
                        continue
                    st = cache.tags[cache.index(l)]
                    if l in st:
                        cache.touch(l, count=False)
                    else:
                        fetched += 1
                        pending[l] = g + latency
    span_rows = sum(len(rows[c]) for c in range(clo, chi))
    return fetched * 128 / (span_rows * heads * 1024)
