// K5-K8: GATConv attention (u_add_v + leaky_relu + edge softmax) fused with the weighted
// neighbour aggregation, forward and backward, for gfx950.
//
// "Row" here is a (node, head) pair: row r = v*H + h owns the D-wide slice ft[v,h,:], so the
// skeleton is the same lane-group row gather as gts_spmm.hip (D=256: one wave per
// (node, head), 1 KiB per gathered slice).  The softmax statistics of a destination row are
// a handful of scalars (in-degree ~6): every lane recomputes them from el/er (L1/L2
// resident, [N,H] fp32) instead of exchanging them, which keeps the edge pass free of
// LDS and barriers.  HBM-bound: no MFMA.
#include "gts_rows.h"

namespace gts {
namespace {

__device__ __forceinline__ float leaky(float x, float slope) { return x > 0.0f ? x : x * slope; }

// Which (node, head) row the w-th unit of work is.  Head-major (all nodes of head 0, then head 1, ...): the rows a
// workgroup's neighbours gather then lie 1 KiB apart per node instead of H KiB, so the window of nodes whose slices
// the XCD's L2 holds is H times longer — the +-x neighbours of a supervoxel lattice stay inside it, as they do for K1.
// Node-major (w = row index) is kept for A/B runs.  The arithmetic of a row does not depend on the walk.
__device__ __forceinline__ int walk_row(int w, int n_rows, int heads, int head_major) {
  if (w < 0 || !head_major) return w;
  const int n_nodes = n_rows / heads;
  const int h = w / n_nodes;
  return (w - h * n_nodes) * heads + h;
}

// sum over the LPR lanes that share a row (xor butterfly stays inside the aligned group)
template <int LPR>
__device__ __forceinline__ float group_sum(float x) {
#pragma unroll
  for (int m = LPR / 2; m >= 1; m >>= 1) x += __shfl_xor(x, m, kWave);
  return x;
}

// The dot products of the backward's edge pass: over a whole wave the two halves of the row are reduced first (16, 8, 4, 2, 1)
// and added last (32) — the clustered edge pass (gts_gat_cluster.hip) computes a column half per unit of work and adds
// the two half sums in its finishing pass: the same tree, the same bits.
template <int LPR>
__device__ __forceinline__ float dot_lanes(float x) {
  if constexpr (LPR == kWave) {
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) x += __shfl_xor(x, m, kWave);
    return x + __shfl_xor(x, 32, kWave);
  } else {
    return group_sum<LPR>(x);
  }
}

template <int LPR>
__device__ __forceinline__ float group_max(float x) {
#pragma unroll
  for (int m = LPR / 2; m >= 1; m >>= 1) x = fmaxf(x, __shfl_xor(x, m, kWave));
  return x;
}

// epilogue of GATConv: + res_fc(h) + bias, then the activation (ELU), all on the way out
template <int VEC>
__device__ __forceinline__ void gat_row_epilogue(const float (&acc)[VEC], float* __restrict__ out, const float* __restrict__ bias,
                                                 const float* __restrict__ residual, int act, int r, int h, int dim, int c,
                                                 bool active) {
  Vec<VEC> o;
#pragma unroll
  for (int t = 0; t < VEC; ++t) o.v[t] = acc[t];
  if (residual != nullptr) {
    const Vec<VEC> rs = Vec<VEC>::load(residual + static_cast<size_t>(r) * dim + c);
#pragma unroll
    for (int t = 0; t < VEC; ++t) o.v[t] += rs.v[t];
  }
  if (bias != nullptr) {
    const Vec<VEC> bs = Vec<VEC>::load(bias + static_cast<size_t>(h) * dim + c);
#pragma unroll
    for (int t = 0; t < VEC; ++t) o.v[t] += bs.v[t];
  }
  if (act == 1) {
#pragma unroll
    for (int t = 0; t < VEC; ++t) o.v[t] = o.v[t] > 0.0f ? o.v[t] : elu_expm1(o.v[t]);
  }
  if (active) o.store(out + static_cast<size_t>(r) * dim + c);
}

// ------------------------------------------------------------------ forward
template <int VEC, int LPR>
__global__ __launch_bounds__(kBlock) void gat_fwd_kernel(
    const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
    const float* __restrict__ ft, const float* __restrict__ el, const float* __restrict__ er,
    float slope, float* __restrict__ out, float* __restrict__ attn,
    const float* __restrict__ bias, const float* __restrict__ residual, int act, int n_rows,
    int heads, int dim, int seq, int head_major) {
  const int gl = (threadIdx.x & (kWave - 1)) % LPR;
  for (int s = 0; s < seq; ++s) {
    const int r = walk_row(owned_row<LPR>(s, seq, n_rows), n_rows, heads, head_major);
    if (r < 0) continue;
    const int v = r / heads, h = r - v * heads;
    const int beg = indptr[v], end = indptr[v + 1];
    const float er_v = er[r];
    if constexpr (LPR == kWave) {
      // One wave per (node, head) and at most 64 in-edges: lane k keeps edge k's source id, score and weight, so
      // the wave's life is three memory round trips (row extent -> edge ids -> scores + source slices) instead
      // of seven (ids and scores were fetched again for the denominator and once more for the gather).  Same
      // arithmetic in the same order as the general path below.
      const int deg = end - beg;
      if (deg <= kWave) {
        const bool mine = gl < deg;
        const int idx = mine ? indices[beg + gl] : 0;
        const float el_u = mine ? el[static_cast<size_t>(idx) * heads + h] : 0.0f;
        float w = 0.0f;
        bool have_weights = false;
        for_columns<VEC, LPR>(dim, [&](int c, bool active) {
          float acc[VEC];
#pragma unroll
          for (int t = 0; t < VEC; ++t) acc[t] = 0.0f;
          for_chunks<LPR>(0, deg, [&](auto cnt_c, int k0) {
            constexpr int CNT = decltype(cnt_c)::value;
            Vec<VEC> val[CNT];
#pragma unroll
            for (int j = 0; j < CNT; ++j) {
              const size_t urow = static_cast<size_t>(__builtin_amdgcn_readlane(idx, k0 + j)) * heads + h;
              val[j] = Vec<VEC>::load(ft + urow * dim + c);
            }
            if (!have_weights) {   // the softmax of the row, while the first source slices are on their way
              const float score = mine ? leaky(el_u + er_v, slope) : -INFINITY;
              const float m = group_max<LPR>(score);
              const float den = group_sum<LPR>(mine ? expf(score - m) : 0.0f);
              w = mine ? expf(score - m) / den : 0.0f;
              if (mine) attn[static_cast<size_t>(beg + gl) * heads + h] = w;
              have_weights = true;
            }
#pragma unroll
            for (int j = 0; j < CNT; ++j) {
              const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w), k0 + j));
#pragma unroll
              for (int t = 0; t < VEC; ++t) acc[t] = mad(a, val[j].v[t], acc[t]);
            }
          });
          gat_row_epilogue<VEC>(acc, out, bias, residual, act, r, h, dim, c, active);
        });
        continue;
      }
    }
    // pass 1: row maximum and softmax denominator; the group's lanes stride over the row's
    // edges and combine with an xor butterfly (all lanes of a group share r: convergent)
    float m = -INFINITY;
    for (int k = beg + gl; k < end; k += LPR)
      m = fmaxf(m, leaky(el[static_cast<size_t>(indices[k]) * heads + h] + er_v, slope));
    m = group_max<LPR>(m);
    float den = 0.0f;
    for (int k = beg + gl; k < end; k += LPR)
      den += expf(leaky(el[static_cast<size_t>(indices[k]) * heads + h] + er_v, slope) - m);
    den = group_sum<LPR>(den);
    // pass 2: weights + weighted gather
    for_columns<VEC, LPR>(dim, [&](int c, bool active) {
      float acc[VEC];
#pragma unroll
      for (int t = 0; t < VEC; ++t) acc[t] = 0.0f;
      for_chunks<LPR>(beg, end, [&](auto cnt_c, int k) {
        constexpr int CNT = decltype(cnt_c)::value;
        const Chunk<LPR, CNT> src(indices, k, end);
        Vec<VEC> val[CNT];
        float a[CNT];
#pragma unroll
        for (int j = 0; j < CNT; ++j) {
          const size_t urow = static_cast<size_t>(src[j]) * heads + h;
          a[j] = el[urow];
          val[j] = Vec<VEC>::load(ft + urow * dim + c);
        }
#pragma unroll
        for (int j = 0; j < CNT; ++j) {
          if (src.valid(j)) {
            a[j] = expf(leaky(a[j] + er_v, slope) - m) / den;
            if (active && c == 0) attn[static_cast<size_t>(k + j) * heads + h] = a[j];
#pragma unroll
            for (int t = 0; t < VEC; ++t) acc[t] = mad(a[j], val[j].v[t], acc[t]);
          }
        }
      });
      gat_row_epilogue<VEC>(acc, out, bias, residual, act, r, h, dim, c, active);
    });
  }
}

// ------------------------------------------------------------------ attention scores
// el[n,h] = <ft[n,h,:], attn_l[h,:]>, er likewise: one pass over ft for both.
template <int VEC, int LPR>
__global__ __launch_bounds__(kBlock) void gat_scores_kernel(
    const float* __restrict__ ft, const float* __restrict__ attn_l,
    const float* __restrict__ attn_r, float* __restrict__ el, float* __restrict__ er, int n_rows,
    int heads, int dim, int seq) {
  const int gl = (threadIdx.x & (kWave - 1)) % LPR;
  for (int s = 0; s < seq; ++s) {
    const int r = owned_row<LPR>(s, seq, n_rows);
    const bool live = r >= 0;          // dead groups still join the wave-wide shuffles with zeros
    const int h = live ? r % heads : 0;
    float sl = 0.0f, sr = 0.0f;
    if (live) {
      for (int c = gl * VEC; c < dim; c += LPR * VEC) {
        const Vec<VEC> f = Vec<VEC>::load(ft + static_cast<size_t>(r) * dim + c);
        const Vec<VEC> al = Vec<VEC>::load(attn_l + static_cast<size_t>(h) * dim + c);
        const Vec<VEC> ar = Vec<VEC>::load(attn_r + static_cast<size_t>(h) * dim + c);
#pragma unroll
        for (int t = 0; t < VEC; ++t) sl += f.v[t] * al.v[t], sr += f.v[t] * ar.v[t];
      }
    }
    sl = group_sum<LPR>(sl);
    sr = group_sum<LPR>(sr);
    if (live && gl == 0) el[r] = sl, er[r] = sr;
  }
}

// ------------------------------------------------------------------ backward, edge pass
// ge[pos,h] first holds ga_k = <gout[v,h,:], ft[src_k,h,:]>, then is overwritten with
// ge_k = a_k (ga_k - sum_j a_j ga_j) leaky'(.) by the same lane (lane 0 of the group).
template <int VEC, int LPR>
__global__ __launch_bounds__(kBlock) void gat_bwd_edge_kernel(
    const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
    const float* __restrict__ ft, const float* __restrict__ el, const float* __restrict__ er,
    const float* __restrict__ attn, const float* __restrict__ gout, float slope,
    float* __restrict__ ge, float* __restrict__ ger, int n_rows, int heads, int dim, int seq, int head_major) {
  const int gl = (threadIdx.x & (kWave - 1)) % LPR;
  for (int s = 0; s < seq; ++s) {
    const int r = walk_row(owned_row<LPR>(s, seq, n_rows), n_rows, heads, head_major);
    // all lanes of a group share r, so the shuffles below are convergent per group;
    // groups past the end still take part in the wave-wide shuffle with zeros.
    const bool live = r >= 0;
    const int v = live ? r / heads : 0, h = live ? r - v * heads : 0;
    const int beg = live ? indptr[v] : 0, end = live ? indptr[v + 1] : 0;
    int maxdeg = end - beg;
    if constexpr (LPR != kWave) {
#pragma unroll
      for (int m = kWave / 2; m >= LPR; m >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, m, kWave));
    }
    float dot_sum = 0.0f;  // sum_j a_j ga_j
    if constexpr (LPR == kWave) {
      const int deg = end - beg;
      if (live && deg <= kWave) {
        // lane k keeps edge k's id, weight, score and dot product: no per-edge scalar is fetched twice and the
        // closing per-edge formula runs on all edges at once instead of one lane walking them with dependent loads
        const bool mine = gl < deg;
        const int idx = mine ? indices[beg + gl] : 0;
        const size_t my_pos = static_cast<size_t>(beg + gl) * heads + h;
        const float a_l = mine ? attn[my_pos] : 0.0f;
        const float pre = mine ? el[static_cast<size_t>(idx) * heads + h] + er[r] : 0.0f;
        float ga = 0.0f;
        for_chunks<LPR>(0, deg, [&](auto cnt_c, int k0) {
          constexpr int CNT = decltype(cnt_c)::value;
          float part[CNT];
#pragma unroll
          for (int j = 0; j < CNT; ++j) part[j] = 0.0f;
          for (int c0 = 0; c0 < dim; c0 += LPR * VEC) {
            const int c = c0 + gl * VEC;
            const bool active = c < dim;
            const int cc = active ? c : 0;
            const Vec<VEC> g = Vec<VEC>::load(gout + static_cast<size_t>(r) * dim + cc);
            Vec<VEC> f[CNT];
#pragma unroll
            for (int j = 0; j < CNT; ++j)
              f[j] = Vec<VEC>::load(ft + (static_cast<size_t>(__builtin_amdgcn_readlane(idx, k0 + j)) * heads + h) * dim + cc);
            if (active) {
#pragma unroll
              for (int j = 0; j < CNT; ++j)
#pragma unroll
                for (int t = 0; t < VEC; ++t) part[j] = mad(g.v[t], f[j].v[t], part[j]);
            }
          }
#pragma unroll
          for (int j = 0; j < CNT; ++j) part[j] = dot_lanes<LPR>(part[j]);
#pragma unroll
          for (int j = 0; j < CNT; ++j) {
            dot_sum += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a_l), k0 + j)) * part[j];
            ga = gl == k0 + j ? part[j] : ga;
          }
        });
        const float g_e = mine ? a_l * (ga - dot_sum) * (pre > 0.0f ? 1.0f : slope) : 0.0f;
        if (mine) ge[my_pos] = g_e;
        float ger_acc = 0.0f;
        for (int k = 0; k < deg; ++k) ger_acc += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g_e), k));
        if (gl == 0) ger[r] = ger_acc;
        continue;
      }
      // one wave per (node, head): the row's source slices are fetched in chunks of <= 8 so that
      // their loads are in flight together, then the chunk's dot products are reduced with
      // independent butterflies (the wave is convergent: r is wave-uniform)
      for_chunks<LPR>(beg, end, [&](auto cnt_c, int k) {
        constexpr int CNT = decltype(cnt_c)::value;
        const Chunk<LPR, CNT> src(indices, k, end);
        float part[CNT];
#pragma unroll
        for (int j = 0; j < CNT; ++j) part[j] = 0.0f;
        for (int c0 = 0; c0 < dim; c0 += LPR * VEC) {
          const int c = c0 + gl * VEC;
          const bool active = c < dim;
          const int cc = active ? c : 0;
          const Vec<VEC> g = Vec<VEC>::load(gout + static_cast<size_t>(r) * dim + cc);
          Vec<VEC> f[CNT];
#pragma unroll
          for (int j = 0; j < CNT; ++j)
            f[j] = Vec<VEC>::load(ft + (static_cast<size_t>(src[j]) * heads + h) * dim + cc);
          if (active) {
#pragma unroll
            for (int j = 0; j < CNT; ++j)
#pragma unroll
              for (int t = 0; t < VEC; ++t) part[j] = mad(g.v[t], f[j].v[t], part[j]);
          }
        }
#pragma unroll
        for (int j = 0; j < CNT; ++j) part[j] = dot_lanes<LPR>(part[j]);
#pragma unroll
        for (int j = 0; j < CNT; ++j) {
          const size_t pos = static_cast<size_t>(k + j) * heads + h;
          dot_sum += attn[pos] * part[j];
          if (gl == 0) ge[pos] = part[j];
        }
      });
    } else {
    for (int i = 0; i < maxdeg; ++i) {
      const int k = beg + i;
      const bool on = k < end;
      float part = 0.0f;
      if (on) {
        const size_t urow = static_cast<size_t>(indices[k]) * heads + h;
        for (int c = gl * VEC; c < dim; c += LPR * VEC) {
          const Vec<VEC> g = Vec<VEC>::load(gout + static_cast<size_t>(r) * dim + c);
          const Vec<VEC> f = Vec<VEC>::load(ft + urow * dim + c);
#pragma unroll
          for (int t = 0; t < VEC; ++t) part = mad(g.v[t], f.v[t], part);
        }
      }
      const float ga = group_sum<LPR>(part);
      if (on) {
        const size_t pos = static_cast<size_t>(k) * heads + h;
        dot_sum += attn[pos] * ga;
        if (gl == 0) ge[pos] = ga;
      }
    }
    }
    if (live && gl == 0) {
      const float er_v = er[r];
      float ger_acc = 0.0f;
      for (int k = beg; k < end; ++k) {
        const size_t pos = static_cast<size_t>(k) * heads + h;
        const float pre = el[static_cast<size_t>(indices[k]) * heads + h] + er_v;
        const float g_e = attn[pos] * (ge[pos] - dot_sum) * (pre > 0.0f ? 1.0f : slope);
        ge[pos] = g_e;
        ger_acc += g_e;
      }
      ger[r] = ger_acc;
    }
  }
}

// ------------------------------------------------------------------ backward, source pass
template <int VEC, int LPR>
__global__ __launch_bounds__(kBlock) void gat_bwd_src_kernel(
    const int32_t* __restrict__ t_indptr, const int32_t* __restrict__ t_indices,
    const int32_t* __restrict__ t_pos, const float* __restrict__ attn,
    const float* __restrict__ ge, const float* __restrict__ gout, float* __restrict__ gft,
    float* __restrict__ gel, const float* __restrict__ attn_l, const float* __restrict__ attn_r,
    const float* __restrict__ ger, int n_rows, int heads, int dim, int seq, int head_major) {
  const int gl = (threadIdx.x & (kWave - 1)) % LPR;
  for (int s = 0; s < seq; ++s) {
    const int r = walk_row(owned_row<LPR>(s, seq, n_rows), n_rows, heads, head_major);
    if (r < 0) continue;
    const int u = r / heads, h = r - u * heads;
    const int beg = t_indptr[u], end = t_indptr[u + 1];
    const float ger_r = ger != nullptr ? ger[r] : 0.0f;
    if constexpr (LPR == kWave) {
      const int deg = end - beg;
      if (deg <= kWave) {
        // lane k keeps out-edge k's destination, weight and score gradient (one round of loads for all of them)
        const bool mine = gl < deg;
        const int idx = mine ? t_indices[beg + gl] : 0;
        const size_t my_pos = mine ? static_cast<size_t>(t_pos[beg + gl]) * heads + h : 0;
        const float a_l = mine ? attn[my_pos] : 0.0f;
        const float ge_l = mine ? ge[my_pos] : 0.0f;
        float gel_r = 0.0f;
        bool have_gel = false;
        for_columns<VEC, LPR>(dim, [&](int c, bool active) {
          float acc[VEC];
#pragma unroll
          for (int t = 0; t < VEC; ++t) acc[t] = 0.0f;
          for_chunks<LPR>(0, deg, [&](auto cnt_c, int k0) {
            constexpr int CNT = decltype(cnt_c)::value;
            Vec<VEC> val[CNT];
#pragma unroll
            for (int j = 0; j < CNT; ++j)
              val[j] = Vec<VEC>::load(gout + (static_cast<size_t>(__builtin_amdgcn_readlane(idx, k0 + j)) * heads + h) * dim + c);
            if (!have_gel) {   // in edge order, while the first gradient slices are on their way
              for (int k = 0; k < deg; ++k)
                gel_r += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ge_l), k));
              if (gl == 0) gel[r] = gel_r;
              have_gel = true;
            }
#pragma unroll
            for (int j = 0; j < CNT; ++j) {
              const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a_l), k0 + j));
#pragma unroll
              for (int t = 0; t < VEC; ++t) acc[t] = mad(a, val[j].v[t], acc[t]);
            }
          });
          Vec<VEC> o;
#pragma unroll
          for (int t = 0; t < VEC; ++t) o.v[t] = acc[t];
          if (attn_l != nullptr) {
            const Vec<VEC> al = Vec<VEC>::load(attn_l + static_cast<size_t>(h) * dim + c);
            const Vec<VEC> ar = Vec<VEC>::load(attn_r + static_cast<size_t>(h) * dim + c);
#pragma unroll
            for (int t = 0; t < VEC; ++t) o.v[t] = mad(ger_r, ar.v[t], mad(gel_r, al.v[t], o.v[t]));
          }
          if (active) o.store(gft + static_cast<size_t>(r) * dim + c);
        });
        if (!have_gel && gl == 0) gel[r] = 0.0f;   // a node without out-edges
        continue;
      }
    }
    float gel_r = 0.0f;   // every lane of the group adds the same few scalars (no exchange needed)
    for (int k = beg; k < end; ++k) gel_r += ge[static_cast<size_t>(t_pos[k]) * heads + h];
    if (gl == 0) gel[r] = gel_r;
    for_columns<VEC, LPR>(dim, [&](int c, bool active) {
      float acc[VEC];
#pragma unroll
      for (int t = 0; t < VEC; ++t) acc[t] = 0.0f;
      for_chunks<LPR>(beg, end, [&](auto cnt_c, int k) {
        constexpr int CNT = decltype(cnt_c)::value;
        const Chunk<LPR, CNT> dst(t_indices, k, end);
        const Chunk<LPR, CNT> pos(t_pos, k, end);
        Vec<VEC> val[CNT];
        float a[CNT];
#pragma unroll
        for (int j = 0; j < CNT; ++j) {
          a[j] = attn[static_cast<size_t>(pos[j]) * heads + h];
          val[j] = Vec<VEC>::load(gout + (static_cast<size_t>(dst[j]) * heads + h) * dim + c);
        }
#pragma unroll
        for (int j = 0; j < CNT; ++j) {
          if (dst.valid(j)) {
#pragma unroll
            for (int t = 0; t < VEC; ++t) acc[t] = mad(a[j], val[j].v[t], acc[t]);
          }
        }
      });
      Vec<VEC> o;
#pragma unroll
      for (int t = 0; t < VEC; ++t) o.v[t] = acc[t];
      if (attn_l != nullptr) {   // el = <ft, attn_l>, er = <ft, attn_r>: their gradient w.r.t. ft
        const Vec<VEC> al = Vec<VEC>::load(attn_l + static_cast<size_t>(h) * dim + c);
        const Vec<VEC> ar = Vec<VEC>::load(attn_r + static_cast<size_t>(h) * dim + c);
#pragma unroll
        for (int t = 0; t < VEC; ++t) o.v[t] = mad(ger_r, ar.v[t], mad(gel_r, al.v[t], o.v[t]));
      }
      if (active) o.store(gft + static_cast<size_t>(r) * dim + c);
    });
  }
}

inline bool bad_gat_shape(int64_t n, int64_t heads, int64_t dim) {
  return n < 0 || heads <= 0 || dim <= 0 || heads > 4096 || dim >= (1LL << 24) ||
         n * heads >= (1LL << 31);
}

}  // namespace
}  // namespace gts

extern "C" int32_t gts_gat_fwd_f32(const int32_t* indptr, const int32_t* indices,
                                   const float* ft, const float* el, const float* er,
                                   float negative_slope, const float* bias,
                                   const float* residual, int32_t activation, float* out,
                                   float* attn, int64_t n, int64_t heads, int64_t dim,
                                   void* stream) {
  using namespace gts;
  if (!indptr || !ft || !el || !er || !out || !attn) return GTS_ERR_NULL;
  if (bad_gat_shape(n, heads, dim)) return GTS_ERR_SHAPE;
  if (activation != 0 && activation != 1) return GTS_ERR_ARGKIND;
  if (n == 0) return GTS_OK;
  const Geometry g = make_geometry(n * heads, dim);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nr = static_cast<int>(n * heads), nh = static_cast<int>(heads), nd = static_cast<int>(dim);
  const int hm = g_gat_walk;
  GTS_DISPATCH_GEOM(g, {
    gat_fwd_kernel<VEC, LPR><<<g.grid, kBlock, 0, st>>>(indptr, indices, ft, el, er, negative_slope, out, attn, bias, residual, activation, nr, nh, nd, g.seq, hm);
  })
  return launch_status();
}

extern "C" int32_t gts_gat_bwd_edge_f32(const int32_t* indptr, const int32_t* indices,
                                        const float* ft, const float* el, const float* er,
                                        const float* attn, const float* gout,
                                        float negative_slope, float* ge, float* ger, int64_t n,
                                        int64_t heads, int64_t dim, void* stream) {
  using namespace gts;
  if (!indptr || !ft || !el || !er || !attn || !gout || !ge || !ger) return GTS_ERR_NULL;
  if (bad_gat_shape(n, heads, dim)) return GTS_ERR_SHAPE;
  if (n == 0) return GTS_OK;
  const Geometry g = make_geometry(n * heads, dim);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nr = static_cast<int>(n * heads), nh = static_cast<int>(heads), nd = static_cast<int>(dim);
  const int hm = g_gat_walk;
  GTS_DISPATCH_GEOM(g, {
    gat_bwd_edge_kernel<VEC, LPR><<<g.grid, kBlock, 0, st>>>(indptr, indices, ft, el, er, attn, gout, negative_slope, ge, ger, nr, nh, nd, g.seq, hm);
  })
  return launch_status();
}

extern "C" int32_t gts_gat_bwd_src_f32(const int32_t* t_indptr, const int32_t* t_indices,
                                       const int32_t* t_pos, const float* attn,
                                       const float* ge, const float* gout,
                                       const float* attn_l, const float* attn_r,
                                       const float* ger, float* gft, float* gel, int64_t n,
                                       int64_t heads, int64_t dim, void* stream) {
  using namespace gts;
  if (!t_indptr || !attn || !ge || !gout || !gft || !gel) return GTS_ERR_NULL;
  if ((attn_l == nullptr) != (attn_r == nullptr) || (attn_l == nullptr) != (ger == nullptr))
    return GTS_ERR_NULL;
  if (bad_gat_shape(n, heads, dim)) return GTS_ERR_SHAPE;
  if (n == 0) return GTS_OK;
  const Geometry g = make_geometry(n * heads, dim);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nr = static_cast<int>(n * heads), nh = static_cast<int>(heads), nd = static_cast<int>(dim);
  const int hm = g_gat_walk;
  GTS_DISPATCH_GEOM(g, {
    gat_bwd_src_kernel<VEC, LPR><<<g.grid, kBlock, 0, st>>>(t_indptr, t_indices, t_pos, attn, ge, gout, gft, gel, attn_l, attn_r, ger, nr, nh, nd, g.seq, hm);
  })
  return launch_status();
}

extern "C" int32_t gts_gat_scores_f32(const float* ft, const float* attn_l, const float* attn_r,
                                      float* el, float* er, int64_t n, int64_t heads, int64_t dim,
                                      void* stream) {
  using namespace gts;
  if (!ft || !attn_l || !attn_r || !el || !er) return GTS_ERR_NULL;
  if (bad_gat_shape(n, heads, dim)) return GTS_ERR_SHAPE;
  if (n == 0) return GTS_OK;
  const Geometry g = make_geometry(n * heads, dim);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nr = static_cast<int>(n * heads), nh = static_cast<int>(heads), nd = static_cast<int>(dim);
  GTS_DISPATCH_GEOM(g, {
    gat_scores_kernel<VEC, LPR><<<g.grid, kBlock, 0, st>>>(ft, attn_l, attn_r, el, er, nr, nh, nd, g.seq);
  })
  return launch_status();
}
