// GATConv's weighted neighbour aggregation (K5 - K7) and the source pass of its backward (K8) at D = 256 over a
// CLUSTER ROW SCHEDULE: the LDS-staged neighbour tiles of gts_spmm_cluster.hip, four times wider.
//
// The plain kernels (gts_gat.hip) give one wave to every (node, head) row and fetch one 1 KiB slice per (edge,
// head) through the CU's L1 out of a table of N x H KiB (245 MB at the C3 shapes) behind the row's index chain:
// 0.30 - 0.36 of the HBM peak.  Here a unit of work is one 512-byte column half of one head of one cluster of
// the graph's schedule (gts/schedule.py, the records of gts_cluster_schedule): 2 H units per cluster, the unit's
// distinct neighbour slices staged once in LDS by LDS-DMA, persistent workgroups (two per CU), gathers one unit
// ahead of the reduction — the streaming form of spmm_cluster_stream_kernel, see there for the pipeline.
//
// What differs from the max-pool reducers is the arithmetic of a row: a weighted sum, weights = the row's edge
// softmax.  They come from a pass of their own (`gat_attn_kernel`: the same expressions in the same association
// as gat_fwd_kernel, so attn [E, H] is bit-identical) and are laid out per (cluster, head) in the order of the
// record's 8-edge chunks (`gat_arrange_kernel`) so that a unit's weights arrive by ONE more LDS-DMA with its
// record; the reduction then adds `w_k * slice_k` in CSR slot order exactly as the plain kernels do.  Bias + ELU
// (forward) and `gel * attn_l + ger * attn_r` (backward) ride in the row's epilogue out of a third small LDS-DMA.
// Rows keep their place: the schedule only decides which workgroup computes which row.
#include <algorithm>
#include <mutex>
#include <vector>

#include "gts_cluster.h"

namespace gts {
namespace {

__device__ __forceinline__ float leaky_relu(float x, float slope) { return x > 0.0f ? x : x * slope; }

template <int G>
__device__ __forceinline__ float lanes_sum(float x) {   // xor butterfly inside the aligned group of G lanes (as group_sum of gts_gat.hip)
#pragma unroll
  for (int m = G / 2; m >= 1; m >>= 1) x += __shfl_xor(x, m, kWave);
  return x;
}
template <int G>
__device__ __forceinline__ float lanes_max(float x) {
#pragma unroll
  for (int m = G / 2; m >= 1; m >>= 1) x = fmaxf(x, __shfl_xor(x, m, kWave));
  return x;
}

// ---- the edge softmax of every (node, head) row: attn [E, H] --------------------------------------------------
// G lanes per row, G >= the largest in-degree (host-checked).  gat_fwd_kernel reduces over the 64 lanes of a wave
// with the row's edges in lanes 0 .. deg-1 and neutral elements elsewhere; the steps of its butterfly that reach
// beyond an aligned group of G lanes add exact zeros (the maximum: -inf), so the G-lane butterfly gives the same
// bits with 64 / G rows per wave.
template <int G>
__global__ __launch_bounds__(kBlock) void gat_attn_kernel(const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                          const float* __restrict__ el, const float* __restrict__ er, float slope,
                                                          float* __restrict__ attn, int n_rows, int heads) {
  const int gl = threadIdx.x % G;
  const long long r = (static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x) / G;
  const bool live = r < n_rows;
  const int v = live ? static_cast<int>(r / heads) : 0, h = live ? static_cast<int>(r - static_cast<long long>(v) * heads) : 0;
  const int beg = live ? indptr[v] : 0, deg = live ? indptr[v + 1] - beg : 0;
  const bool mine = gl < deg;
  const int idx = mine ? indices[beg + gl] : 0;
  const float el_u = mine ? el[static_cast<size_t>(idx) * heads + h] : 0.0f;
  const float er_v = live ? er[r] : 0.0f;
  const float score = mine ? leaky_relu(el_u + er_v, slope) : -INFINITY;
  const float m = lanes_max<G>(score);
  const float den = lanes_sum<G>(mine ? expf(score - m) : 0.0f);
  const float w = mine ? expf(score - m) / den : 0.0f;
  if (mine) attn[static_cast<size_t>(beg + gl) * heads + h] = w;
}

// ---- weights (and per-row scalars) in the order of the schedule's records -------------------------------------
// side [n_clusters][H][side_floats]: chunk weights [chunk_slots][8] (the record's 8-edge chunks; pads 0), then for the
// backward pass per row of the cluster (gel, ger).  One thread per (cluster, row slot, head).
//   forward : weight of the row's k-th in-edge  = attn[(indptr[v] + k) H + h]
//   backward: weight of the row's k-th out-edge = attn[t_pos[t_indptr[u] + k] H + h];  gel[u, h] = sum_k ge[t_pos[..] H + h]
//             in edge order (as gat_bwd_src_kernel adds them), also written to gel [N, H]
template <bool BWD>
__global__ __launch_bounds__(kBlock) void gat_arrange_kernel(const int32_t* __restrict__ rec, RecLayout layout, int n_clusters,
                                                             int max_rows, int heads, const int32_t* __restrict__ row_ptr,
                                                             const int32_t* __restrict__ t_pos, const float* __restrict__ attn,
                                                             const float* __restrict__ ge, const float* __restrict__ ger,
                                                             float* __restrict__ gel, float* __restrict__ side, int side_floats,
                                                             int chunk_slots, unsigned* __restrict__ counters) {
  if (blockIdx.x == 0) counters[threadIdx.x] = 0u;   // the streaming kernel behind this pass deals its units off them (kBlock = 256 words)
  const long long tid = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x;
  const int per_cluster = max_rows * heads;
  const long long c = tid / per_cluster;
  if (c >= n_clusters) return;
  const int rem = static_cast<int>(tid - c * per_cluster), j = rem / heads, h = rem - j * heads;
  const int32_t* r = rec + static_cast<size_t>(c) * layout.words;
  if (j >= r[0]) return;
  const int row = r[layout.rows + j];
  const uint32_t info = static_cast<uint32_t>(r[layout.eoff + j]);
  const int c0 = info & 0xFFFF, deg = info >> 16;
  const int beg = row_ptr[row];
  float* block = side + (static_cast<size_t>(c) * heads + h) * side_floats;
  float* dst = block + c0 * 8;
  float gel_r = 0.0f;
  const int padded = (deg + 7) & ~7;
  for (int k = 0; k < padded; ++k) {
    float w = 0.0f;
    if (k < deg) {
      const size_t pos = static_cast<size_t>(BWD ? t_pos[beg + k] : beg + k) * heads + h;
      w = attn[pos];
      if constexpr (BWD) gel_r += ge[pos];
    }
    dst[k] = w;
  }
  if constexpr (BWD) {
    const size_t at = static_cast<size_t>(row) * heads + h;
    gel[at] = gel_r;
    float* scal = block + chunk_slots * 8 + j * 2;
    scal[0] = gel_r;
    scal[1] = ger != nullptr ? ger[at] : 0.0f;
  }
}

// The two passes above in one, for graphs whose rows fit ONE 8-edge chunk (every graph the default rules send here): eight
// lanes per (cluster, head, row slot) = the row's edges, walked in the ORDER OF THE RECORDS, so the weights land in their
// blocks as whole 32-byte pieces next to each other (a record's rows own consecutive chunks).  Forward: the softmax itself
// (lanes_max<8> / lanes_sum<8>: gat_fwd_kernel's bits, see gat_attn_kernel), attn [E, H] written on the way.  Backward: the
// weights through t_pos, and gel[u, h] = the row's ge added in edge order across the eight lanes.
template <bool BWD>
__global__ __launch_bounds__(kBlock) void gat_weights_one_chunk_kernel(const int32_t* __restrict__ rec, RecLayout layout, int n_clusters,
                                                                       int max_rows, int heads, const int32_t* __restrict__ row_ptr,
                                                                       const int32_t* __restrict__ indices_or_pos,
                                                                       const float* __restrict__ el, const float* __restrict__ er,
                                                                       float slope, float* __restrict__ attn_out,
                                                                       const float* __restrict__ attn_in, const float* __restrict__ ge,
                                                                       const float* __restrict__ ger, float* __restrict__ gel,
                                                                       float* __restrict__ side, int side_floats, int chunk_slots,
                                                                       unsigned* __restrict__ counters) {
  if (blockIdx.x == 0) counters[threadIdx.x] = 0u;   // the streaming kernel behind this pass deals its units off them (kBlock = 256 words)
  // eight lanes per (cluster, row slot); every lane walks the heads (their values sit next to each other in el / er / attn / ge:
  // one dependent chain of index loads per row instead of one per (row, head))
  const unsigned idx = blockIdx.x * static_cast<unsigned>(kBlock) + threadIdx.x;   // < 2^31 (host-checked)
  const int k = static_cast<int>(idx & 7);
  const unsigned grp = idx >> 3;
  const unsigned c = grp / static_cast<unsigned>(max_rows);
  const int j = static_cast<int>(grp - c * max_rows);
  const int32_t* r = rec + static_cast<size_t>(c < static_cast<unsigned>(n_clusters) ? c : 0u) * layout.words;
  const bool live = c < static_cast<unsigned>(n_clusters) && j < r[0];
  const int row = live ? r[layout.rows + j] : 0;
  const uint32_t info = live ? static_cast<uint32_t>(r[layout.eoff + j]) : 0u;
  const int c0 = info & 0xFFFF, deg = info >> 16;          // deg <= 8 (host-checked)
  const int beg = live ? row_ptr[row] : 0;
  const bool mine = k < deg;
  const size_t at = static_cast<size_t>(row) * heads;
  const size_t edge = mine ? static_cast<size_t>(BWD ? indices_or_pos[beg + k] : beg + k) * heads : 0;   // this lane's row of attn / ge
  // the edge's source id out of the record itself (its position byte -> the cluster's neighbour list): the el fetch then waits for
  // the record only, not for indptr -> indices
  const int at_src = mine ? reinterpret_cast<const uint8_t*>(r + layout.loc)[c0 * 8 + k] : 0;
  const size_t src = (!BWD && mine) ? static_cast<size_t>(r[layout.srcs + at_src]) * heads : 0;           // ... and of el
  float* block = side + static_cast<size_t>(c) * heads * side_floats;
  const int base = (threadIdx.x & (kWave - 1)) & ~7;
#pragma unroll 4
  for (int h = 0; h < heads; ++h) {
    float w = 0.0f;
    if constexpr (!BWD) {
      const float el_u = mine ? el[src + h] : 0.0f;
      const float er_v = live ? er[at + h] : 0.0f;
      const float score = mine ? leaky_relu(el_u + er_v, slope) : -INFINITY;
      const float m = lanes_max<8>(score);
      const float den = lanes_sum<8>(mine ? expf(score - m) : 0.0f);
      w = mine ? expf(score - m) / den : 0.0f;
      if (mine) attn_out[edge + h] = w;
    } else {
      w = mine ? attn_in[edge + h] : 0.0f;
      const float g = mine ? ge[edge + h] : 0.0f;
      float gel_r = 0.0f;                                   // edge order; the lanes past the row's end add +0.0
#pragma unroll
      for (int q = 0; q < 8; ++q) gel_r += __shfl(g, base + q, kWave);
      if (live && k == 0) {
        gel[at + h] = gel_r;
        float* scal = block + static_cast<size_t>(h) * side_floats + chunk_slots * 8 + j * 2;
        scal[0] = gel_r;
        scal[1] = ger != nullptr ? ger[at + h] : 0.0f;
      }
    }
    if (live && deg > 0) block[static_cast<size_t>(h) * side_floats + c0 * 8 + k] = w;
  }
}

// Step 2 of the clustered edge pass (rows of one 8-edge chunk): the two half dot products of every edge are added (the last
// step of gat_bwd_edge_kernel's dot_lanes), then the row's closing formula exactly as gat_bwd_edge_kernel writes it:
//   ge_k = a_k (ga_k - sum_j a_j ga_j) leaky'(el[src_k] + er[v]),  ger[v, h] = sum_k ge_k   (sums in edge order).
// Eight lanes per (cluster, row slot), heads inside, in the order of the records.
__global__ __launch_bounds__(kBlock) void gat_edge_finish_kernel(const int32_t* __restrict__ rec, RecLayout layout, int n_clusters, int max_rows,
                                                                 int heads, const int32_t* __restrict__ indptr,
                                                                 const int32_t* __restrict__ indices, const float* __restrict__ el,
                                                                 const float* __restrict__ er, const float* __restrict__ attn, float slope,
                                                                 const float* __restrict__ halves, float* __restrict__ ge,
                                                                 float* __restrict__ ger, int chunk_slots) {
  const unsigned idx = blockIdx.x * static_cast<unsigned>(kBlock) + threadIdx.x;   // < 2^31 (host-checked)
  const int k = static_cast<int>(idx & 7);
  const unsigned grp = idx >> 3;
  const unsigned c = grp / static_cast<unsigned>(max_rows);
  const int j = static_cast<int>(grp - c * max_rows);
  const int32_t* r = rec + static_cast<size_t>(c < static_cast<unsigned>(n_clusters) ? c : 0u) * layout.words;
  const bool live = c < static_cast<unsigned>(n_clusters) && j < r[0];
  const int row = live ? r[layout.rows + j] : 0;
  const uint32_t info = live ? static_cast<uint32_t>(r[layout.eoff + j]) : 0u;
  const int c0 = info & 0xFFFF, deg = info >> 16;          // deg <= 8 (host-checked)
  const int beg = live ? indptr[row] : 0;
  const bool mine = k < deg;
  const size_t at = static_cast<size_t>(row) * heads;
  const size_t edge = mine ? static_cast<size_t>(beg + k) * heads : 0;
  const int at_src = mine ? reinterpret_cast<const uint8_t*>(r + layout.loc)[c0 * 8 + k] : 0;   // source id out of the record (see above)
  const size_t src = mine ? static_cast<size_t>(r[layout.srcs + at_src]) * heads : 0;
  const size_t half_stride = static_cast<size_t>(n_clusters) * heads * chunk_slots * 8;
  const int base = (threadIdx.x & (kWave - 1)) & ~7;
#pragma unroll 4
  for (int h = 0; h < heads; ++h) {
    const size_t slot = (static_cast<size_t>(c) * heads + h) * chunk_slots * 8 + c0 * 8 + k;
    const float ga = mine ? halves[slot] + halves[half_stride + slot] : 0.0f;
    const float a_l = mine ? attn[edge + h] : 0.0f;
    const float pre = mine ? el[src + h] + er[at + h] : 0.0f;
    float dot_sum = 0.0f;
#pragma unroll
    for (int q = 0; q < 8; ++q) dot_sum += __shfl(a_l, base + q, kWave) * __shfl(ga, base + q, kWave);   // lanes past the row's end: + 0 * 0
    const float g_e = mine ? a_l * (ga - dot_sum) * (pre > 0.0f ? 1.0f : slope) : 0.0f;
    if (mine) ge[edge + h] = g_e;
    float ger_acc = 0.0f;
#pragma unroll
    for (int q = 0; q < 8; ++q) ger_acc += __shfl(g_e, base + q, kWave);
    if (live && k == 0) ger[at + h] = ger_acc;
  }
}

// ---- the persistent streaming kernel ----------------------------------------------------------------------------
struct GatClusterArgs {
  const int32_t* rec;      // [n_clusters][layout.words]
  RecLayout layout;
  int n_clusters, max_srcs;
  const float* table;      // forward: ft [N, H, 256];  source pass: g_pre [N, H, 256];  edge pass: ft
  const float* own;        // edge pass: g_pre [N, H, 256] (every row's own slice, staged behind the neighbours')
  const float* side;       // gat_arrange_kernel's blocks
  const float* vec;        // forward: bias [H, 256] or null;  backward: attn_l | attn_r [2][H, 256] or null
  float* out;              // forward: out [N, H, 256];  backward: gft
  unsigned table_bytes, side_bytes, vec_bytes;
  unsigned row_bytes, own_row_bytes;   // bytes between the rows of `table` / `own` (4 H KiB)
  unsigned own_bytes;
  int heads, act, nt;
  int side_floats, chunk_slots, side_pieces;
  int rec_bytes, side_slot_bytes, image_bytes;   // LDS: 3 record slots | 3 side slots | 3 vector slots of 1 KiB | 2 images | 64 B of dealt units
  int deal_off;
  int own_off;             // edge pass: byte offset of the own-row section inside an image
  int group;               // clusters of an XCD's span walked together through all their (head, half) slices (see the kernel)
  unsigned* counters;      // dynamic dealing: one unit counter per XCD, 32 words apart, zero at launch (null: static round-robin dealing)
};

// the rows of one unit out of LDS: half a wave per row, 16 B per lane; acc = w_k * slice_k + acc in slot order (`mad`), one
// straight-line body per exact edge count of an 8-edge chunk.  The kernel is bound by the vector instructions it issues
// (profiles/r04: without its gathers it takes 3 / 4 of its time), so the common pair of rows — equal degrees, one chunk — runs
// a body without per-edge masks: the exact count is both rows' count.  Returns the number of store instructions issued.
template <bool BWD, bool NOSTORE = false>   // NOSTORE: tools/diag what-if runs only
__device__ __forceinline__ int reduce_wsum_rows(const GatClusterArgs& a, const int32_t* l_rec, const unsigned char* image,
                                                const float* l_side, const float* l_vec, int sub, int first, int step) {
  const int lane = threadIdx.x & (kWave - 1), half = lane >> 5, hl = lane & 31;
  const int n_rows = l_rec[0];
  const uint32_t* info = reinterpret_cast<const uint32_t*>(l_rec + a.layout.eoff);
  const uint2* loc = reinterpret_cast<const uint2*>(l_rec + a.layout.loc);
  const unsigned char* mine = image + hl * 16;
  int trips = 0;
  for (int j0 = 2 * first; j0 < n_rows; j0 += 2 * step, ++trips) {   // j0 is wave-uniform: the even row of the pair
    const int j = j0 + half;
    const bool have = j < n_rows;
    const uint32_t ri = have ? info[j] : 0u;
    const int c0 = ri & 0xFFFF, deg = ri >> 16;
    const int deg_a = __builtin_amdgcn_readlane(deg, 0), deg_b = __builtin_amdgcn_readlane(deg, 32);
    const int deg_w = max(deg_a, deg_b);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (deg_a == deg_b && deg_w <= 8) {
      if (deg_w > 0) {
        const uint2 w = loc[c0];
        const float4 wa = *reinterpret_cast<const float4*>(l_side + c0 * 8), wb = *reinterpret_cast<const float4*>(l_side + c0 * 8 + 4);
        const float wts[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
        for_count(deg_w, [&](auto cnt_c) {
          constexpr int CNT = decltype(cnt_c)::value;
          float4 val[CNT];
#pragma unroll
          for (int q = 0; q < CNT; ++q) val[q] = *reinterpret_cast<const float4*>(mine + chunk_byte(w, q) * kHalfBytes);
#pragma unroll
          for (int q = 0; q < CNT; ++q) {
            acc[0] = mad(wts[q], val[q].x, acc[0]);
            acc[1] = mad(wts[q], val[q].y, acc[1]);
            acc[2] = mad(wts[q], val[q].z, acc[2]);
            acc[3] = mad(wts[q], val[q].w, acc[3]);
          }
        });
      }
    } else {
      for (int c = 0; 8 * c < deg_w; ++c) {
        const int at = 8 * c < deg ? c0 + c : c0;                 // the shorter row of the pair re-reads its first chunk, masked below
        const uint2 w = loc[at];
        const float4 wa = *reinterpret_cast<const float4*>(l_side + at * 8), wb = *reinterpret_cast<const float4*>(l_side + at * 8 + 4);
        const float wts[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
        for_count(min(8, deg_w - 8 * c), [&](auto cnt_c) {
          constexpr int CNT = decltype(cnt_c)::value;
          float4 val[CNT];
#pragma unroll
          for (int q = 0; q < CNT; ++q) val[q] = *reinterpret_cast<const float4*>(mine + chunk_byte(w, q) * kHalfBytes);
#pragma unroll
          for (int q = 0; q < CNT; ++q) {
            const bool live = 8 * c + q < deg;
            const float v4[4] = {val[q].x, val[q].y, val[q].z, val[q].w};
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = live ? mad(wts[q], v4[t], acc[t]) : acc[t];
          }
        });
      }
    }
    Vec<4> o{{acc[0], acc[1], acc[2], acc[3]}};
    if constexpr (BWD) {
      if (a.vec != nullptr) {   // el = <ft, attn_l>, er = <ft, attn_r>: their gradient w.r.t. ft
        const float2 sc = have ? *reinterpret_cast<const float2*>(l_side + a.chunk_slots * 8 + j * 2) : float2{0.f, 0.f};
        const float4 al = *reinterpret_cast<const float4*>(l_vec + hl * 4), ar = *reinterpret_cast<const float4*>(l_vec + 128 + hl * 4);
        const float al4[4] = {al.x, al.y, al.z, al.w}, ar4[4] = {ar.x, ar.y, ar.z, ar.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) o.v[t] = mad(sc.y, ar4[t], mad(sc.x, al4[t], o.v[t]));
      }
    } else {
      if (a.vec != nullptr) {
        const float4 bs = *reinterpret_cast<const float4*>(l_vec + hl * 4);
        o.v[0] += bs.x, o.v[1] += bs.y, o.v[2] += bs.z, o.v[3] += bs.w;
      }
      if (a.act == 1) {   // ELU: both pairs through elu_expm1_pair (gts_rows.h), then the selects — NaN included
        const v2f lo = elu_expm1_pair(v2f{o.v[0], o.v[1]}), hi = elu_expm1_pair(v2f{o.v[2], o.v[3]});
        o.v[0] = o.v[0] > 0.0f ? o.v[0] : lo.x;
        o.v[1] = o.v[1] > 0.0f ? o.v[1] : lo.y;
        o.v[2] = o.v[2] > 0.0f ? o.v[2] : hi.x;
        o.v[3] = o.v[3] > 0.0f ? o.v[3] : hi.y;
      }
    }
    if constexpr (NOSTORE) {
      if (o.v[0] == 123.456f) a.out[0] = o.v[1] + o.v[2] + o.v[3];
    } else if (have) {   // streamed out (GTS_OPT_CLUSTER_STREAMING, default on): the rows just written do not push the halo slices out of the XCD's L2
      float* dst = a.out + (static_cast<size_t>(l_rec[a.layout.rows + j]) * a.heads * kF + sub * (kF / 2) + hl * 4);
      if (a.nt) o.store_nt(dst); else o.store(dst);
    }
  }
  return NOSTORE ? 0 : trips;
}

// Edge pass of the backward (K8, step 1): ga_k = <g_pre[v, h, :], ft[src_k, h, :]> for every in-edge.  A unit holds one column
// half, so it produces HALF dot products: each lane adds its four products in order, the 32 lanes of the row's half wave
// combine with the butterfly 16, 8, 4, 2, 1 — the first five steps of gat_bwd_edge_kernel's `dot_lanes` (gts_gat.hip), whose
// last step (x + x^32) adds the two halves: gat_edge_finish_kernel does exactly that addition, so ga has the plain kernel's
// bits.  Partials go to side[part][cluster][head][chunk][8] (the record's chunk order): 32 B per row and chunk.
__device__ __forceinline__ int reduce_edge_rows(const GatClusterArgs& a, const int32_t* l_rec, const unsigned char* image,
                                                float* out_block, int first, int step) {
  const int lane = threadIdx.x & (kWave - 1), half = lane >> 5, hl = lane & 31;
  const int n_rows = l_rec[0];
  const uint32_t* info = reinterpret_cast<const uint32_t*>(l_rec + a.layout.eoff);
  const uint2* loc = reinterpret_cast<const uint2*>(l_rec + a.layout.loc);
  const unsigned char* mine = image + hl * 16;
  int stores = 0;
  for (int j0 = 2 * first; j0 < n_rows; j0 += 2 * step) {
    const int j = j0 + half;
    const bool have = j < n_rows;
    const uint32_t ri = have ? info[j] : 0u;
    const int c0 = ri & 0xFFFF, deg = ri >> 16;
    const int deg_w = max(__builtin_amdgcn_readlane(deg, 0), __builtin_amdgcn_readlane(deg, 32));
    const float4 g = *reinterpret_cast<const float4*>(image + a.own_off + (have ? j : 0) * kHalfBytes + hl * 16);
    for (int c = 0; 8 * c < deg_w; ++c, ++stores) {
      const bool mine_too = 8 * c < deg;
      const uint2 w = loc[mine_too ? c0 + c : c0];
      float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for_count(min(8, deg_w - 8 * c), [&](auto cnt_c) {
        constexpr int CNT = decltype(cnt_c)::value;
        float4 val[CNT];
#pragma unroll
        for (int q = 0; q < CNT; ++q) val[q] = *reinterpret_cast<const float4*>(mine + chunk_byte(w, q) * kHalfBytes);
#pragma unroll
        for (int q = 0; q < CNT; ++q) {
          part[q] = mad(g.x, val[q].x, part[q]);
          part[q] = mad(g.y, val[q].y, part[q]);
          part[q] = mad(g.z, val[q].z, part[q]);
          part[q] = mad(g.w, val[q].w, part[q]);
        }
      });
      // The butterfly 16, 8, 4, 2, 1 of all eight sums at once: at each of the first three steps a lane keeps the half of
      // its values its lane bit selects and hands the other half to its partner, which adds them to the ones IT keeps —
      // every sum still meets the same partners in the same order (9 cross-lane moves instead of 40; fp32 addition
      // commutes).  Lane hl ends up with the sum of edge 4 (hl / 16 % 2) + 2 (hl / 8 % 2) + hl / 4 % 2.
      const bool b16 = (hl & 16) != 0, b8 = (hl & 8) != 0, b4 = (hl & 4) != 0;
      float k4[4], k2[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) k4[i] = (b16 ? part[i + 4] : part[i]) + __shfl_xor(b16 ? part[i] : part[i + 4], 16, kWave);
#pragma unroll
      for (int i = 0; i < 2; ++i) k2[i] = (b8 ? k4[i + 2] : k4[i]) + __shfl_xor(b8 ? k4[i] : k4[i + 2], 8, kWave);
      float k1 = (b4 ? k2[1] : k2[0]) + __shfl_xor(b4 ? k2[0] : k2[1], 4, kWave);
      k1 += __shfl_xor(k1, 2, kWave);
      k1 += __shfl_xor(k1, 1, kWave);
      const int q_mine = (b16 ? 4 : 0) + (b8 ? 2 : 0) + (b4 ? 1 : 0);
      if (have && mine_too && (hl & 3) == 0) out_block[(c0 + c) * 8 + q_mine] = k1;
    }
  }
  return stores;
}

// gridDim.x is a multiple of 8: the workgroups with blockIdx % 8 == x (one XCD under round-robin placement; speed only)
// share the x-th eighth of the CLUSTERS and walk its units sub-unit by sub-unit (all clusters of the span for head 0 /
// left half, then head 0 / right half, ...), round-robin: the units in flight on an XCD at one time are the same
// column half of neighbouring clusters, whose halo slices meet in its L2.
// MINW = 8: up to 16 waves per workgroup at <= 64 registers; 6: up to 12 waves at <= 80 (two workgroups per CU either way)
// MODE 0: forward aggregation;  1: source pass of the backward;  2: edge pass of the backward (half dot products)
// DEPTH: units the gathers run ahead of the reduction (DEPTH + 1 images).  A unit's reduction is short beside the round trip
// of its gathers, so with one unit ahead a workgroup spends most of an iteration waiting for the LAST of ~35 KB to land and the
// kernel's rate is (bytes in flight) / (loaded latency), whatever the L2 hits (profiles/r04: the time does not move with the
// bytes fetched); deeper images keep more of the LDS in flight.  Records run max(1, DEPTH - 1) units in front of the gathers.
// WHATIF (tools/diag only): 1 = no neighbour gathers, 2 = no reduction and no stores, 3 = reduction without stores
template <int MODE, int MINW, int DEPTH = 1, int WHATIF = 0>
__global__ __launch_bounds__(1024, MINW) void gat_cluster_stream_kernel(const GatClusterArgs a) {
  constexpr bool BWD = MODE == 1;
  constexpr int AHEAD = DEPTH > 1 ? DEPTH - 1 : 1;   // records are fetched this many units in front of their gathers
  constexpr int R = DEPTH + AHEAD + 1;               // record / weight / vector slots
  constexpr int IMAGES = DEPTH + 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int n_waves = blockDim.x / kWave;
  const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  const int subs = 2 * a.heads;
  const int clo = static_cast<int>(static_cast<long long>(a.n_clusters) * xcd / 8);
  const int span = static_cast<int>(static_cast<long long>(a.n_clusters) * (xcd + 1) / 8) - clo;
  const long long n_local = static_cast<long long>(span) * subs;
  // Dealing.  STATIC (a.counters == nullptr): unit i of the span to workgroup i % per_xcd.  A workgroup that falls ONE unit behind
  // its neighbours is then per_xcd units behind in the walk — a whole generation — and the halo slices it shares with them have
  // left the L2 before it asks: with the reduction in the loop the kernel fetched 1.57 x the table where the same walk without it
  // (all workgroups in step) fetched 1.20 x (profiles/r04/gat_l2_replay.log).  DYNAMIC (the default): the workgroups of an XCD
  // take their units from ONE counter (an agent-scope atomic add by the workgroup's last wave, one iteration before the unit's
  // record is fetched; the value goes round through LDS with the iteration's barrier), so the units in flight are always the
  // latest per_xcd ones of the walk, whatever the pace of each workgroup.  Which workgroup computes a unit changes nothing in what
  // it computes.
  const bool dynamic = a.counters != nullptr;
  const int n_static = jw < n_local ? static_cast<int>((n_local - jw + per_xcd - 1) / per_xcd) : 0;
  if (!dynamic && n_static == 0) return;
  const int words = a.layout.words;
  const int rec_pieces = a.rec_bytes / 1024;
  unsigned char* side_slots = lds + R * a.rec_bytes;
  unsigned char* vec_slots = side_slots + R * a.side_slot_bytes;
  unsigned char* images = MODE == 2 ? lds + R * a.rec_bytes : vec_slots + R * 1024;   // the edge pass keeps no weight / vector slots
  const RawDma rr(a.rec, static_cast<unsigned>(a.n_clusters) * words * 4u);
  const RawDma rt(a.table, a.table_bytes);
  const RawDma rs(a.side, a.side_bytes);
  const RawDma rv(a.vec, a.vec != nullptr ? a.vec_bytes : 0);
  const RawDma ro(MODE == 2 ? a.own : a.table, MODE == 2 ? a.own_bytes : a.table_bytes);
  // Unit order inside the span: groups of `a.group` consecutive clusters, a group walked through ALL its (head, half) slices
  // before the next one — consecutive units (= what the XCD's workgroups hold at one time) are the same slice of the group's
  // clusters, then the next slice of the same clusters.
  const unsigned group = static_cast<unsigned>(a.group > 0 && a.group < span ? a.group : span);
  // The walk is kept as scalar state and advanced by per_xcd units at a time (a few scalar operations: the two integer divisions
  // of the closed form, taken three times per iteration, were a fifth of the kernel's time — profiles/r04): `at` = the unit
  // fetch_record takes next, (first, size) its group, s its slice, c its cluster inside the group.
  unsigned w_first, w_size, w_s, w_c;
  {
    const unsigned i = static_cast<unsigned>(jw);                                  // this workgroup's first unit: the closed form, once
    const unsigned per_group = group * static_cast<unsigned>(subs);
    const unsigned gi = i / per_group, r = i - gi * per_group;
    w_first = gi * group;
    w_size = min(group, static_cast<unsigned>(span) - w_first);
    w_s = r / w_size;
    w_c = r - w_s * w_size;
  }
  auto advance = [&]() {                     // per_xcd units on; past the span's end the state is never used (size kept >= 1)
    w_c += static_cast<unsigned>(per_xcd);
    while (w_c >= w_size) {
      w_c -= w_size;
      if (++w_s == static_cast<unsigned>(subs)) {
        w_s = 0;
        w_first += group;
        w_size = w_first < static_cast<unsigned>(span) ? min(group, static_cast<unsigned>(span) - w_first) : 0x40000000u;
      }
    }
  };
  // (cluster, sub) of the units in the pipeline: [0] = the one fetch_record took last ... [PIPE - 1] = the one being reduced;
  // cluster < 0: no unit (the span is exhausted)
  constexpr int PIPE = DEPTH + AHEAD + 1;
  int u_cluster[PIPE], u_sub[PIPE];
#pragma unroll
  for (int q = 0; q < PIPE; ++q) u_cluster[q] = -1, u_sub[q] = 0;
  int32_t* l_deal = reinterpret_cast<int32_t*>(lds + a.deal_off);   // dynamic dealing: 8 x (cluster, sub); 0 .. PIPE - 2 the prologue's, 6 / 7 the loop's
  int entered = 0;
  auto next_unit = [&](int deal_slot) {      // shift the pipeline and enter the next unit at [0]
#pragma unroll
    for (int q = PIPE - 1; q > 0; --q) u_cluster[q] = u_cluster[q - 1], u_sub[q] = u_sub[q - 1];
    if (dynamic) {
      u_cluster[0] = __builtin_amdgcn_readfirstlane(l_deal[2 * deal_slot]);
      u_sub[0] = __builtin_amdgcn_readfirstlane(l_deal[2 * deal_slot + 1]);
    } else if (entered < n_static) {
      u_cluster[0] = clo + static_cast<int>(w_first + w_c), u_sub[0] = static_cast<int>(w_s);
      advance();
      ++entered;
    } else {
      u_cluster[0] = -1;
    }
  };
  // the dealer (the workgroup's last wave, which fetches no record piece): `count` units off the XCD's counter (take) into
  // l_deal[slot0 ..] (publish).  In the loop the add is issued in one iteration and published at the top of the next, behind the
  // wait every wave makes there anyway: the dealer never waits for the counter's round trip on its own.
  auto take = [&](int count) {
    unsigned base = 0;
    if (lane == 0) base = __hip_atomic_fetch_add(a.counters + 32 * xcd, static_cast<unsigned>(count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return base;
  };
  auto publish = [&](unsigned taken, int count, int slot0) {
    const unsigned base = __builtin_amdgcn_readfirstlane(taken);
    for (int q = 0; q < count; ++q) {
      const unsigned idx = base + q;
      const bool ok = idx < static_cast<unsigned long long>(n_local);
      const unsigned sl = idx / static_cast<unsigned>(max(span, 1));   // the whole span slice by slice (an empty span deals nothing)
      if (lane == 0) {
        l_deal[2 * (slot0 + q)] = ok ? clo + static_cast<int>(idx - sl * static_cast<unsigned>(span)) : -1;
        l_deal[2 * (slot0 + q) + 1] = ok ? static_cast<int>(sl) : 0;
      }
    }
  };
  // record, weights and epilogue vectors of unit t -> their slots, one LDS-DMA per piece and wave.  Returns this wave's count
  auto fetch_record = [&](int t, int cluster_id, int sub) {
    const unsigned cluster = static_cast<unsigned>(cluster_id);
    const int slot = t % R;
    if (wave < rec_pieces) {
      const int word = 256 * wave + 4 * lane;
      rr(lds + slot * a.rec_bytes + 1024 * wave, word < words ? (cluster * static_cast<unsigned>(words) + word) * 4u : 0xFFFFFFF0u);
      return 1;
    } else if (MODE == 2) {
      // the edge pass fetches nothing but the record
    } else if (wave < rec_pieces + a.side_pieces) {
      const int p = wave - rec_pieces, word = 256 * p + 4 * lane;
      const unsigned base = (cluster * static_cast<unsigned>(a.heads) + static_cast<unsigned>(sub >> 1)) * static_cast<unsigned>(a.side_floats);
      rs(side_slots + slot * a.side_slot_bytes + 1024 * p, word < a.side_floats ? (base + word) * 4u : 0xFFFFFFF0u);
      return 1;
    } else if (wave == rec_pieces + a.side_pieces && a.vec != nullptr) {
      // 128 floats of this (head, half) from each vector: lanes 0-31 the first (bias / attn_l), lanes 32-63 the second (attn_r)
      const unsigned first = static_cast<unsigned>(sub) * 128u + (lane & 31) * 4u;
      const unsigned off = (lane < 32 ? first : BWD ? static_cast<unsigned>(a.heads) * kF + first : 0x3FFFFFFCu) * 4u;
      rv(vec_slots + slot * 1024, off);
      return 1;
    }
    return 0;
  };
  auto issue_gathers = [&](int t, int sub) {   // this wave's share of unit t's gathers; its record is in LDS.  Returns their count
    const int32_t* l_rec = reinterpret_cast<const int32_t*>(lds + (t % R) * a.rec_bytes);
    unsigned char* image = images + (t % IMAGES) * a.image_bytes;
    const int n_srcs = l_rec[1];
    // lane l keeps the id of row 2 (wave + n_waves (l / 2)) + l % 2: every row pair this wave fetches, one LDS read
    const int last = pad4(a.max_srcs) - 1;
    const int32_t ids = l_rec[a.layout.srcs + min(2 * (wave + n_waves * (lane >> 1)) + (lane & 1), last)];
    const unsigned row_bytes = a.row_bytes;
    const unsigned col = static_cast<unsigned>(sub) * kHalfBytes + (lane & 31) * 16u;
    int k = 0;
    if constexpr (WHATIF != 1) {
      for (int i = 2 * wave; i < n_srcs; i += 2 * n_waves, ++k) {
        const int s0 = __builtin_amdgcn_readlane(ids, 2 * k), s1 = __builtin_amdgcn_readlane(ids, 2 * k + 1);
        rt(image + i * kHalfBytes, static_cast<unsigned>(lane < 32 ? s0 : s1) * row_bytes + col);
      }
    }
    int k2 = 0;
    if constexpr (MODE == 2 && WHATIF != 1) {   // ... and of its rows' own gradient slices, the same way out of the other table
      const int n_rows = l_rec[0];
      const int32_t own_ids = l_rec[a.layout.rows + min(2 * (wave + n_waves * (lane >> 1)) + (lane & 1), n_rows - 1)];
      for (int i = 2 * wave; i < n_rows; i += 2 * n_waves, ++k2) {
        const int s0 = __builtin_amdgcn_readlane(own_ids, 2 * k2), s1 = __builtin_amdgcn_readlane(own_ids, 2 * k2 + 1);
        ro(image + a.own_off + i * kHalfBytes, static_cast<unsigned>(lane < 32 ? s0 : s1) * a.own_row_bytes + col);
      }
    }
    return k + k2;
  };

  // Iteration `it` issues, in this order: [record / weights / vectors of unit it + DEPTH + AHEAD] [gathers of unit it + DEPTH]
  // [stores of unit it].  At its top the gathers of unit `it` and the record of unit it + DEPTH must have landed; vector-memory
  // operations retire in order, and the record of unit it + DEPTH was issued in front of the gathers of unit it + 1 (AHEAD =
  // DEPTH - 1; for DEPTH 1 in front of those of unit `it`), so everything YOUNGER than that record fetch may stay in flight:
  // the gathers of units it + 1 .. it + DEPTH - 1, the stores of the last DEPTH - 1 iterations (for DEPTH 1: of the last
  // one), and the record fetches of the last DEPTH - 2 iterations.
  constexpr int HIST = DEPTH > 1 ? DEPTH - 1 : 1;
  int g_hist[HIST], s_hist[HIST], r_hist[HIST];   // [0] = youngest; wave-uniform counts of this wave's operations
#pragma unroll
  for (int q = 0; q < HIST; ++q) g_hist[q] = s_hist[q] = r_hist[q] = 0;
  if (dynamic) {                               // the first PIPE units of this workgroup in one add: PIPE - 1 for the prologue, one for iteration 0
    if (wave == n_waves - 1) {
      const unsigned first = take(PIPE);      // ONE round trip for the prologue's units and the one iteration 0 enters
      publish(first, PIPE - 1, 0);
      publish(first + (PIPE - 1), 1, 6);
    }
    barrier_all();
  }
  unsigned taken = 0;                          // the dealer's add of the iteration before
#pragma unroll
  for (int t = 0; t < DEPTH + AHEAD; ++t) {   // afterwards u_*[DEPTH + AHEAD - 1 - t] is unit t
    next_unit(t);
    if (u_cluster[0] >= 0) fetch_record(t, u_cluster[0], u_sub[0]);
  }
  barrier_all();
#pragma unroll
  for (int t = 0; t < DEPTH; ++t) {
    const int cnt = u_cluster[DEPTH + AHEAD - 1 - t] >= 0 ? issue_gathers(t, u_sub[DEPTH + AHEAD - 1 - t]) : 0;
    if (t >= 1) {                              // the gathers of units 1 .. DEPTH - 1 stay in flight past the first wait
#pragma unroll
      for (int q = HIST - 1; q > 0; --q) g_hist[q] = g_hist[q - 1];
      g_hist[0] = cnt;
    }
  }
  // WHATIF == 9 (tools/diag only): shader-clock stamps of wave 0 around the phases of an iteration, summed per workgroup into
  // `out` (then a buffer of 8 x uint64 per workgroup, no rows are stored): wait for gathers | barrier | issue | reduce
  unsigned long long phase[4] = {0, 0, 0, 0}, stamp = 0;
  auto lap = [&](int which) {
    if constexpr (WHATIF == 9) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      phase[which] += now - stamp;
      stamp = now;
    }
  };
  if constexpr (WHATIF == 9) stamp = __builtin_amdgcn_s_memtime();
  int it = 0;
  for (;; ++it) {
    int keep = 0;
    if constexpr (DEPTH == 1) {
      keep = s_hist[0];
    } else {
#pragma unroll
      for (int q = 0; q < DEPTH - 1; ++q) keep += g_hist[q] + s_hist[q];
#pragma unroll
      for (int q = 0; q < DEPTH - 2; ++q) keep += r_hist[q];
    }
    wait_vm_all_but(keep);
    if (dynamic && it > 0 && wave == n_waves - 1) publish(taken, 1, 6 + (it & 1));   // this slot was last read two barriers ago
    lap(0);
    barrier_lds();                          // ... and everyone else's; the oldest image and the oldest slots are free
    lap(1);
    next_unit(6 + (it & 1));                // u_*[0] = unit it + DEPTH + AHEAD, [AHEAD] = unit it + DEPTH, [PIPE - 1] = unit it
    if (u_cluster[PIPE - 1] < 0) break;     // units come in walk order: nothing behind an empty slot
    if (dynamic && wave == n_waves - 1) taken = take(1);   // published at the top of the next iteration
    const int recs = u_cluster[0] >= 0 ? fetch_record(it + DEPTH + AHEAD, u_cluster[0], u_sub[0]) : 0;
    const int gathers = u_cluster[AHEAD] >= 0 ? issue_gathers(it + DEPTH, u_sub[AHEAD]) : 0;
    lap(2);
    const int sub = u_sub[PIPE - 1], cluster = u_cluster[PIPE - 1];
    const int slot = it % R;
    int stores = 0;
    if constexpr (WHATIF == 2) {
      (void)cluster;
    } else if constexpr (MODE == 2) {
      // partial dot products of this (cluster, head) for column half `sub & 1`
      float* block = a.out + (static_cast<size_t>(sub & 1) * a.n_clusters * a.heads + static_cast<size_t>(cluster) * a.heads + (sub >> 1)) *
                                 (a.chunk_slots * 8);
      stores = reduce_edge_rows(a, reinterpret_cast<const int32_t*>(lds + slot * a.rec_bytes), images + (it % IMAGES) * a.image_bytes, block,
                                wave, n_waves);
    } else {
      stores = reduce_wsum_rows<BWD, WHATIF == 3 || WHATIF == 9>(a, reinterpret_cast<const int32_t*>(lds + slot * a.rec_bytes),
                                                   images + (it % IMAGES) * a.image_bytes,
                                                   reinterpret_cast<const float*>(side_slots + slot * a.side_slot_bytes),
                                                   reinterpret_cast<const float*>(vec_slots + slot * 1024), sub, wave, n_waves);
    }
#pragma unroll
    for (int q = HIST - 1; q > 0; --q) g_hist[q] = g_hist[q - 1], s_hist[q] = s_hist[q - 1], r_hist[q] = r_hist[q - 1];
    g_hist[0] = gathers, s_hist[0] = stores, r_hist[0] = recs;
    lap(3);
  }
  if constexpr (WHATIF == 9) {
    if (threadIdx.x == 0) {
      unsigned long long* dbg = reinterpret_cast<unsigned long long*>(a.out) + 8 * static_cast<size_t>(blockIdx.x);
      for (int q = 0; q < 4; ++q) dbg[q] = phase[q];
      dbg[4] = static_cast<unsigned long long>(it);
    }
  }
}

struct GatPlan {
  int chunk_slots, side_floats, side_pieces, rec_bytes, side_slot_bytes, image_bytes, waves, depth;
  int64_t wg_lds;
};
// depth: units the gathers run ahead.  The library runs depth 1 (two images, two workgroups per CU); 2 and 3 are instantiated
// by tools/diag/gat_whatif.hip only (measured: no gain — the kernels are bound by the vector instructions they issue, and the
// deeper images leave room for one workgroup per CU; profiles/r04/gat_depth_ab.log)
inline GatPlan gat_plan(int max_rows, int max_srcs, int loc_words, bool tagged, bool bwd, bool edge = false, int depth = 1) {
  GatPlan p;
  p.chunk_slots = loc_words / 2;
  p.side_floats = p.chunk_slots * 8 + (bwd ? max_rows * 2 : 0);
  p.side_pieces = (p.side_floats + 255) / 256;
  p.rec_bytes = 1024 * ((rec_layout(max_rows, max_srcs, loc_words, tagged).words + 255) / 256);
  p.side_slot_bytes = 1024 * p.side_pieces;
  p.image_bytes = (((max_srcs + 1) & ~1) + (edge ? (max_rows + 1) & ~1 : 0)) * kHalfBytes;   // edge pass: the rows' own slices behind the neighbours'
  p.waves = g_gat_cluster_waves > 0 ? std::max(4, std::min(16, g_gat_cluster_waves)) : 12;
  for (p.depth = std::max(1, std::min(3, depth));; --p.depth) {
    const int64_t slots = p.depth + std::max(1, p.depth - 1) + 1;
    p.wg_lds = slots * p.rec_bytes + (edge ? 0 : slots * (p.side_slot_bytes + 1024)) + (p.depth + 1LL) * p.image_bytes + 64;
    if (p.wg_lds <= kMaxLds || p.depth == 1) break;
  }
  return p;
}

inline bool bad_gat_cluster(int64_t n_clusters, int32_t max_rows, int32_t max_srcs, int32_t loc_words, bool tagged, bool bwd,
                            int64_t n, int64_t heads, int64_t dim, bool edge = false) {
  if (n_clusters < 0 || n_clusters >= (1 << 28) || max_rows < 1 || max_srcs < 1 || max_srcs > 256 || loc_words < 2 ||
      loc_words % 4 != 0 || n < 0 || heads < 1 || heads > 64 || dim != kF)
    return true;
  if (n * heads * kF * 4 >= (1LL << 32) || n_clusters * heads * 2 + 4096 >= (1LL << 31)) return true;   // 32-bit byte offsets into the table; unit ids
  if (n_clusters * rec_layout(max_rows, max_srcs, loc_words, tagged).words * 4 >= (1LL << 32)) return true;
  if (n_clusters * max_rows * 8 + 4096 >= (1LL << 31)) return true;                               // thread ids of the weight pass
  const GatPlan p = gat_plan(max_rows, max_srcs, loc_words, tagged, bwd, edge);
  if (n_clusters * heads * p.side_floats * 4 * (edge ? 2 : 1) >= (1LL << 32)) return true;     // ... and into the weight blocks
  if (rec_layout(max_rows, max_srcs, loc_words, tagged).words > 512 || p.wg_lds > kMaxLds) return true;
  if (p.rec_bytes / 1024 + p.side_pieces + 1 > p.waves || max_srcs > 64 * p.waves || max_rows > 2 * 64) return true;
  return false;
}

template <int MODE, int WHATIF = 0, int MAXDEPTH = 1>
int launch_gat_cluster(GatClusterArgs a, const GatPlan& p, hipStream_t st, int64_t grid_override = 0) {   // override: tools/diag only
  a.side_floats = p.side_floats, a.chunk_slots = p.chunk_slots, a.side_pieces = p.side_pieces;
  a.rec_bytes = p.rec_bytes, a.side_slot_bytes = p.side_slot_bytes, a.image_bytes = p.image_bytes;
  a.own_off = ((a.max_srcs + 1) & ~1) * kHalfBytes;
  a.deal_off = static_cast<int>(p.wg_lds) - 64;
  if (g_gat_cluster_dealing == 1) a.counters = nullptr;   // static round-robin dealing (A/B runs)
  a.group = g_gat_cluster_group > 0 ? g_gat_cluster_group : 16;   // profiles/r04: source pass 152 -> 146 us, the other two passes unchanged
  const int per_cu = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>({g_cluster_per_cu > 0 ? g_cluster_per_cu : 2, kMaxLds / p.wg_lds,
                                                                               static_cast<int64_t>(32 / p.waves)})));
  const int64_t units = 2LL * a.heads * a.n_clusters;
  int64_t grid = static_cast<int64_t>(device_cus()) * per_cu;
  grid = std::max<int64_t>(8, std::min(grid, (units + 7) / 8 * 8)) / 8 * 8;
  if (grid_override > 0) grid = (grid_override + 7) / 8 * 8;
  auto go = [&](auto kernel) {
    static std::mutex guard;               // the LDS attribute once per kernel, not per launch
    static std::vector<const void*> allowed;
    {
      std::lock_guard<std::mutex> lock(guard);
      const void* id = reinterpret_cast<const void*>(kernel);
      if (std::find(allowed.begin(), allowed.end(), id) == allowed.end()) allow_big_lds(kernel), allowed.push_back(id);
    }
    kernel<<<dim3(static_cast<unsigned>(grid)), p.waves * kWave, p.wg_lds, st>>>(a);
    return launch_status();
  };
  if constexpr (MAXDEPTH >= 3) {
    if (p.depth == 3) return p.waves > 12 ? go(gat_cluster_stream_kernel<MODE, 8, 3, WHATIF>) : go(gat_cluster_stream_kernel<MODE, 6, 3, WHATIF>);
  }
  if constexpr (MAXDEPTH >= 2) {
    if (p.depth == 2) return p.waves > 12 ? go(gat_cluster_stream_kernel<MODE, 8, 2, WHATIF>) : go(gat_cluster_stream_kernel<MODE, 6, 2, WHATIF>);
  }
  if (p.depth != 1) return GTS_ERR_ARGKIND;
  return p.waves > 12 ? go(gat_cluster_stream_kernel<MODE, 8, 1, WHATIF>) : go(gat_cluster_stream_kernel<MODE, 6, 1, WHATIF>);
}

constexpr int64_t kCounterBytes = 1024;   // 8 XCDs x 32 words: every counter on a line of its own
inline int64_t weight_block_bytes(int64_t n_clusters, int64_t heads, int64_t side_floats) {
  return (n_clusters * heads * side_floats * static_cast<int64_t>(sizeof(float)) + 255) / 256 * 256;
}

template <typename Launch>
int for_group_size(int64_t max_degree, Launch&& launch) {
  if (max_degree <= 8) return launch(IC<8>{});
  if (max_degree <= 16) return launch(IC<16>{});
  if (max_degree <= 32) return launch(IC<32>{});
  return launch(IC<64>{});
}

}  // namespace
}  // namespace gts

extern "C" int64_t gts_gat_cluster_workspace(int64_t n_clusters, int32_t max_rows, int32_t loc_words, int64_t heads,
                                             int32_t backward) {
  if (n_clusters < 0 || max_rows < 1 || loc_words < 2 || heads < 1) return 0;
  const int64_t side_floats = (loc_words / 2) * 8 + (backward ? max_rows * 2 : 0);
  return gts::weight_block_bytes(n_clusters, heads, side_floats) + gts::kCounterBytes;   // the unit counters of the streaming kernel behind the blocks
}

extern "C" int32_t gts_gat_attn_f32(const int32_t* indptr, const int32_t* indices, const float* el, const float* er,
                                    float negative_slope, float* attn, int64_t n, int64_t heads, int64_t max_degree,
                                    void* stream) {
  using namespace gts;
  if (!indptr || !el || !er || !attn) return GTS_ERR_NULL;
  if (n < 0 || heads < 1 || n * heads >= (1LL << 31) || max_degree < 0 || max_degree > kWave) return GTS_ERR_SHAPE;
  if (n == 0 || max_degree == 0) return GTS_OK;
  if (!indices) return GTS_ERR_NULL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nr = static_cast<int>(n * heads), nh = static_cast<int>(heads);
  return for_group_size(max_degree, [&](auto g_c) {
    constexpr int G = decltype(g_c)::value;
    const int64_t threads = static_cast<int64_t>(nr) * G;
    gat_attn_kernel<G><<<static_cast<unsigned>((threads + kBlock - 1) / kBlock), kBlock, 0, st>>>(indptr, indices, el, er, negative_slope,
                                                                                                 attn, nr, nh);
    return launch_status();
  });
}

extern "C" int32_t gts_gat_fwd_cluster_f32(const int32_t* indptr, const int32_t* indices, const int32_t* rec, int64_t n_clusters,
                                           int32_t max_rows, int32_t max_srcs, int32_t loc_words, int32_t tagged, const float* ft,
                                           const float* el, const float* er, float negative_slope, const float* bias,
                                           int32_t activation, float* out, float* attn, float* workspace, int64_t workspace_bytes,
                                           int64_t n, int64_t heads, int64_t dim, int64_t max_degree, void* stream) {
  using namespace gts;
  if (!indptr || !rec || !ft || !el || !er || !out || !attn || !workspace) return GTS_ERR_NULL;
  if (bad_gat_cluster(n_clusters, max_rows, max_srcs, loc_words, tagged != 0, false, n, heads, dim)) return GTS_ERR_SHAPE;
  if (activation != 0 && activation != 1) return GTS_ERR_ARGKIND;
  if (workspace_bytes < gts_gat_cluster_workspace(n_clusters, max_rows, loc_words, heads, 0)) return GTS_ERR_SHAPE;
  if (n == 0 || n_clusters == 0) return GTS_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const GatPlan p = gat_plan(max_rows, max_srcs, loc_words, tagged != 0, false);
  const RecLayout lay = rec_layout(max_rows, max_srcs, loc_words, tagged != 0);
  const int nh = static_cast<int>(heads);
  const int64_t threads = n_clusters * max_rows * heads;
  unsigned* counters = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(workspace) + weight_block_bytes(n_clusters, heads, p.side_floats));
  int rc = GTS_OK;
  if (max_degree <= 8) {
    if (!indices && max_degree > 0) return GTS_ERR_NULL;
    gat_weights_one_chunk_kernel<false><<<static_cast<unsigned>((8 * (threads / heads) + kBlock - 1) / kBlock), kBlock, 0, st>>>(
        rec, lay, static_cast<int>(n_clusters), max_rows, nh, indptr, indices, el, er, negative_slope, attn, nullptr, nullptr, nullptr,
        nullptr, workspace, p.side_floats, p.chunk_slots, counters);
  } else {
    rc = gts_gat_attn_f32(indptr, indices, el, er, negative_slope, attn, n, heads, max_degree, stream);
    if (rc != GTS_OK) return rc;
    gat_arrange_kernel<false><<<static_cast<unsigned>((threads + kBlock - 1) / kBlock), kBlock, 0, st>>>(
        rec, lay, static_cast<int>(n_clusters), max_rows, nh, indptr, nullptr, attn, nullptr, nullptr, nullptr, workspace, p.side_floats,
        p.chunk_slots, counters);
  }
  rc = launch_status();
  if (rc != GTS_OK) return rc;
  GatClusterArgs a{};
  a.rec = rec, a.layout = lay, a.n_clusters = static_cast<int>(n_clusters), a.max_srcs = max_srcs;
  a.table = ft, a.side = workspace, a.vec = bias, a.out = out, a.counters = counters;
  a.row_bytes = static_cast<unsigned>(heads * kF * 4);
  a.table_bytes = static_cast<unsigned>(n * a.row_bytes);
  a.side_bytes = static_cast<unsigned>(n_clusters * heads * p.side_floats * 4);
  a.vec_bytes = static_cast<unsigned>(heads * kF * 4);
  a.heads = nh, a.act = activation, a.nt = g_cluster_nt < 0 ? 1 : (g_cluster_nt & 1);
  return launch_gat_cluster<0>(a, p, st);
}

extern "C" int32_t gts_gat_bwd_src_cluster_f32(const int32_t* t_indptr, const int32_t* t_pos, const int32_t* rec, int64_t n_clusters,
                                               int32_t max_rows, int32_t max_srcs, int32_t loc_words, int32_t tagged,
                                               const float* attn, const float* ge, const float* gout, const float* attn_lr,
                                               const float* ger, float* gft, float* gel, float* workspace, int64_t workspace_bytes,
                                               int64_t n, int64_t heads, int64_t dim, int64_t max_degree, void* stream) {
  using namespace gts;
  if (!t_indptr || !rec || !attn || !ge || !gout || !gft || !gel || !workspace) return GTS_ERR_NULL;
  if ((attn_lr == nullptr) != (ger == nullptr)) return GTS_ERR_NULL;
  if (bad_gat_cluster(n_clusters, max_rows, max_srcs, loc_words, tagged != 0, true, n, heads, dim)) return GTS_ERR_SHAPE;
  if (workspace_bytes < gts_gat_cluster_workspace(n_clusters, max_rows, loc_words, heads, 1)) return GTS_ERR_SHAPE;
  if (n == 0 || n_clusters == 0) return GTS_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const GatPlan p = gat_plan(max_rows, max_srcs, loc_words, tagged != 0, true);
  const RecLayout lay = rec_layout(max_rows, max_srcs, loc_words, tagged != 0);
  const int nh = static_cast<int>(heads);
  const int64_t threads = n_clusters * max_rows * heads;
  unsigned* counters = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(workspace) + weight_block_bytes(n_clusters, heads, p.side_floats));
  if (max_degree >= 0 && max_degree <= 8)
    gat_weights_one_chunk_kernel<true><<<static_cast<unsigned>((8 * (threads / heads) + kBlock - 1) / kBlock), kBlock, 0, st>>>(
        rec, lay, static_cast<int>(n_clusters), max_rows, nh, t_indptr, t_pos, nullptr, nullptr, 0.0f, nullptr, attn, ge, ger, gel,
        workspace, p.side_floats, p.chunk_slots, counters);
  else
    gat_arrange_kernel<true><<<static_cast<unsigned>((threads + kBlock - 1) / kBlock), kBlock, 0, st>>>(
        rec, lay, static_cast<int>(n_clusters), max_rows, nh, t_indptr, t_pos, attn, ge, ger, gel, workspace, p.side_floats, p.chunk_slots,
        counters);
  const int rc = launch_status();
  if (rc != GTS_OK) return rc;
  GatClusterArgs a{};
  a.rec = rec, a.layout = lay, a.n_clusters = static_cast<int>(n_clusters), a.max_srcs = max_srcs;
  a.table = gout, a.side = workspace, a.vec = attn_lr, a.out = gft, a.counters = counters;
  a.row_bytes = static_cast<unsigned>(heads * kF * 4);
  a.table_bytes = static_cast<unsigned>(n * a.row_bytes);
  a.side_bytes = static_cast<unsigned>(n_clusters * heads * p.side_floats * 4);
  a.vec_bytes = static_cast<unsigned>(2 * heads * kF * 4);
  a.heads = nh, a.act = 0, a.nt = g_cluster_nt < 0 ? 1 : (g_cluster_nt & 1);
  return launch_gat_cluster<1>(a, p, st);
}

extern "C" int32_t gts_gat_bwd_edge_cluster_f32(const int32_t* indptr, const int32_t* indices, const int32_t* rec, int64_t n_clusters,
                                                int32_t max_rows, int32_t max_srcs, int32_t loc_words, int32_t tagged, const float* ft,
                                                const float* el, const float* er, const float* attn, const float* gout,
                                                float negative_slope, float* ge, float* ger, float* workspace, int64_t workspace_bytes,
                                                int64_t n, int64_t heads, int64_t dim, int64_t max_degree, void* stream) {
  using namespace gts;
  if (!indptr || !rec || !ft || !el || !er || !attn || !gout || !ge || !ger || !workspace) return GTS_ERR_NULL;
  if (bad_gat_cluster(n_clusters, max_rows, max_srcs, loc_words, tagged != 0, false, n, heads, dim, true)) return GTS_ERR_SHAPE;
  if (max_degree < 0 || max_degree > 8) return GTS_ERR_SHAPE;     // rows of one 8-edge chunk (the finishing pass)
  if (workspace_bytes < 2 * gts_gat_cluster_workspace(n_clusters, max_rows, loc_words, heads, 0)) return GTS_ERR_SHAPE;
  if (n == 0 || n_clusters == 0) return GTS_OK;
  if (!indices && max_degree > 0) return GTS_ERR_NULL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const GatPlan p = gat_plan(max_rows, max_srcs, loc_words, tagged != 0, false, true);
  const RecLayout lay = rec_layout(max_rows, max_srcs, loc_words, tagged != 0);
  GatClusterArgs a{};
  a.rec = rec, a.layout = lay, a.n_clusters = static_cast<int>(n_clusters), a.max_srcs = max_srcs;
  // the two halves' partial dot products, then the unit counters (no pass of ours runs in front: a memset node zeroes them)
  unsigned* counters = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(workspace) + 2 * weight_block_bytes(n_clusters, heads, p.side_floats));
  if (g_gat_cluster_dealing != 1 && hipMemsetAsync(counters, 0, kCounterBytes, st) != hipSuccess) return launch_status();
  a.table = ft, a.own = gout, a.out = workspace, a.counters = counters;
  a.row_bytes = a.own_row_bytes = static_cast<unsigned>(heads * kF * 4);
  a.table_bytes = static_cast<unsigned>(n * a.row_bytes), a.own_bytes = static_cast<unsigned>(n * a.own_row_bytes);
  a.heads = static_cast<int>(heads);
  int rc = launch_gat_cluster<2>(a, p, st);
  if (rc != GTS_OK) return rc;
  const int64_t threads = n_clusters * max_rows * 8;
  gat_edge_finish_kernel<<<static_cast<unsigned>((threads + kBlock - 1) / kBlock), kBlock, 0, st>>>(
      rec, lay, static_cast<int>(n_clusters), max_rows, static_cast<int>(heads), indptr, indices, el, er, attn, negative_slope, workspace, ge,
      ger, p.chunk_slots);
  return launch_status();
}
