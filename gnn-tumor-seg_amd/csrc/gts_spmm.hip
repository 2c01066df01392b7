// K1-K4: CSR neighbour gather + segmented max / sum reducers for gfx950 (MI355X).
//
// HBM-bound integer-indexed row gathers: no MFMA here.  Layout decisions:
//   * one lane owns VEC=4 consecutive fp32 columns (16 B/lane); LPR lanes cover one
//     feature row, so an F=256 row is exactly one 1 KiB wave-wide load (64 x 16 B);
//   * F=256 (LPR=64): the row index is wave-uniform, so indptr / indices travel through
//     the scalar path (s_load) and only feature rows use vector memory;
//   * up to UNROLL neighbour rows are requested before the first one is consumed, and
//     each wave walks `seq` consecutive destination rows: with 16-32 waves per CU that is
//     >100 KiB of row loads in flight per CU;
//   * workgroup -> row-tile mapping is XCD-contiguous (gts_common.h) so that the
//     re-reads of neighbouring rows hit the XCD's own L2.
#include "gts_rows.h"

namespace gts {
namespace {

template <int ARGB>
struct ArgTraits;
template <>
struct ArgTraits<1> {
  using T = uint8_t;
  static constexpr int kNone = 0xFF;
};
template <>
struct ArgTraits<4> {
  using T = int32_t;
  static constexpr int kNone = -1;
};

template <int VEC, int ARGB>
__device__ __forceinline__ void store_slots(void* arg, size_t off, const int (&slot)[VEC]) {
  if constexpr (ARGB == 1) {
    uint8_t* p = static_cast<uint8_t*>(arg) + off;
    if constexpr (VEC == 4) {
      const uint32_t w = (slot[0] & 0xFF) | ((slot[1] & 0xFF) << 8) | ((slot[2] & 0xFF) << 16) |
                         (static_cast<uint32_t>(slot[3] & 0xFF) << 24);
      *reinterpret_cast<uint32_t*>(p) = w;
    } else {
      p[0] = static_cast<uint8_t>(slot[0]);
    }
  } else if constexpr (ARGB == 4) {
    int32_t* p = static_cast<int32_t*>(arg) + off;
    if constexpr (VEC == 4) {
      *reinterpret_cast<int4*>(p) = make_int4(slot[0], slot[1], slot[2], slot[3]);
    } else {
      p[0] = slot[0];
    }
  }
}

template <int VEC, int ARGB>
__device__ __forceinline__ void load_slots(const void* arg, size_t off, int (&slot)[VEC]) {
  if constexpr (ARGB == 1) {
    const uint8_t* p = static_cast<const uint8_t*>(arg) + off;
    if constexpr (VEC == 4) {
      const uint32_t w = *reinterpret_cast<const uint32_t*>(p);
      slot[0] = w & 0xFF;
      slot[1] = (w >> 8) & 0xFF;
      slot[2] = (w >> 16) & 0xFF;
      slot[3] = w >> 24;
    } else {
      slot[0] = p[0];
    }
  } else {
    const int32_t* p = static_cast<const int32_t*>(arg) + off;
    if constexpr (VEC == 4) {
      const int4 w = *reinterpret_cast<const int4*>(p);
      slot[0] = w.x, slot[1] = w.y, slot[2] = w.z, slot[3] = w.w;
    } else {
      slot[0] = p[0];
    }
  }
}

// ------------------------------------------------------------------ K1: max forward
template <int VEC, int LPR, int ARGB>
__global__ __launch_bounds__(kBlock) void spmm_max_fwd_kernel(
    const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
    const float* __restrict__ x, float* __restrict__ out, void* __restrict__ arg, int n_dst,
    int n_feat, int seq, int nt, int relu_input) {
  for (int s = 0; s < seq; ++s) {
    const int v = owned_row<LPR>(s, seq, n_dst);
    if (v < 0) continue;
    const int beg = indptr[v], end = indptr[v + 1];
    for_columns<VEC, LPR>(n_feat, [&](int c, bool active) {
      float best[VEC];
      int slot[VEC];
#pragma unroll
      for (int t = 0; t < VEC; ++t) best[t] = -INFINITY, slot[t] = -1;
      for_chunks<LPR>(beg, end, [&](auto cnt_c, int k) {
        constexpr int CNT = decltype(cnt_c)::value;
        const Chunk<LPR, CNT> src(indices, k, end);
        Vec<VEC> val[CNT];
#pragma unroll
        for (int j = 0; j < CNT; ++j)
          val[j] = Vec<VEC>::load(x + static_cast<size_t>(src[j]) * n_feat + c);
#pragma unroll
        for (int j = 0; j < CNT; ++j) {
          if (src.valid(j)) {
#pragma unroll
            for (int t = 0; t < VEC; ++t) {
              if (best[t] < val[j].v[t]) best[t] = val[j].v[t], slot[t] = k + j - beg;
            }
          }
        }
      });
      Vec<VEC> o;
#pragma unroll
      for (int t = 0; t < VEC; ++t) {
        const bool dead = isinf(best[t]);  // empty row (-inf) or a +-inf maximum -> 0, no winner
        o.v[t] = dead ? 0.0f : best[t];
        // x = relu(.): a maximum that is not positive carries no gradient (relu'(0) = 0), so it
        // is recorded as "no winner" and the backward never has to look at x again
        slot[t] = (dead || (relu_input && !(best[t] > 0.0f))) ? -1 : slot[t];
      }
      const size_t off = static_cast<size_t>(v) * n_feat + c;
      if (active) {
        if (nt & 1) o.store_nt(out + off); else o.store(out + off);
        if constexpr (ARGB != 0) store_slots<VEC, ARGB>(arg, off, slot);
      }
    });
  }
}

// ------------------------------------------------------------------ K2: max backward
// A stored "no winner" (0xFF / -1) never equals a real slot (ARGB = 1 needs in-degree <= 254).
template <int VEC, int LPR, int ARGB>
__global__ __launch_bounds__(kBlock) void spmm_max_bwd_kernel(
    const int32_t* __restrict__ t_indptr, const int32_t* __restrict__ t_indices,
    const int32_t* __restrict__ t_slot, const float* __restrict__ gout,
    const void* __restrict__ arg, const float* __restrict__ relu_src, float* __restrict__ gx,
    int n_src, int n_feat, int seq, int nt) {
  for (int s = 0; s < seq; ++s) {
    const int u = owned_row<LPR>(s, seq, n_src);
    if (u < 0) continue;
    const int beg = t_indptr[u], end = t_indptr[u + 1];
    for_columns<VEC, LPR>(n_feat, [&](int c, bool active) {
      float acc[VEC];
#pragma unroll
      for (int t = 0; t < VEC; ++t) acc[t] = 0.0f;
      for_chunks<LPR>(beg, end, [&](auto cnt_c, int k) {
        constexpr int CNT = decltype(cnt_c)::value;
        const Chunk<LPR, CNT> dst(t_indices, k, end);
        const Chunk<LPR, CNT> slt(t_slot, k, end);
        Vec<VEC> g[CNT];
        int win[CNT][VEC];
        int want[CNT];
#pragma unroll
        for (int j = 0; j < CNT; ++j) {
          const size_t off = static_cast<size_t>(dst[j]) * n_feat + c;
          want[j] = slt[j];
          g[j] = Vec<VEC>::load(gout + off);
          load_slots<VEC, ARGB>(arg, off, win[j]);
        }
#pragma unroll
        for (int j = 0; j < CNT; ++j) {
          if (dst.valid(j)) {
#pragma unroll
            for (int t = 0; t < VEC; ++t) acc[t] += (win[j][t] == want[j]) ? g[j].v[t] : 0.0f;
          }
        }
      });
      const size_t off = static_cast<size_t>(u) * n_feat + c;
      Vec<VEC> o;
      if (relu_src != nullptr) {
        const Vec<VEC> p = (nt & 2) ? Vec<VEC>::load_nt(relu_src + off) : Vec<VEC>::load(relu_src + off);
#pragma unroll
        for (int t = 0; t < VEC; ++t) o.v[t] = p.v[t] > 0.0f ? acc[t] : 0.0f;
      } else {
#pragma unroll
        for (int t = 0; t < VEC; ++t) o.v[t] = acc[t];
      }
      if (active) { if (nt & 1) o.store_nt(gx + off); else o.store(gx + off); }
    });
  }
}

// ------------------------------------------------------------------ K3/K4: sum family
template <int VEC, int LPR, bool DIV_IN>
__global__ __launch_bounds__(kBlock) void spmm_sum_kernel(
    const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
    const float* __restrict__ x, float* __restrict__ out, const float* __restrict__ div_in,
    const float* __restrict__ div_out, const float* __restrict__ accum, int add_self, int n_out, int n_feat,
    int seq) {
  for (int s = 0; s < seq; ++s) {
    const int v = owned_row<LPR>(s, seq, n_out);
    if (v < 0) continue;
    const int beg = indptr[v], end = indptr[v + 1];
    for_columns<VEC, LPR>(n_feat, [&](int c, bool active) {
      float acc[VEC];
#pragma unroll
      for (int t = 0; t < VEC; ++t) acc[t] = 0.0f;
      for_chunks<LPR>(beg, end, [&](auto cnt_c, int k) {
        constexpr int CNT = decltype(cnt_c)::value;
        const Chunk<LPR, CNT> src(indices, k, end);
        Vec<VEC> val[CNT];
        float d[CNT];
#pragma unroll
        for (int j = 0; j < CNT; ++j) {
          const int u = src[j];
          d[j] = DIV_IN ? div_in[u] : 1.0f;
          val[j] = Vec<VEC>::load(x + static_cast<size_t>(u) * n_feat + c);
        }
#pragma unroll
        for (int j = 0; j < CNT; ++j) {
          if (src.valid(j)) {
#pragma unroll
            for (int t = 0; t < VEC; ++t) acc[t] += DIV_IN ? val[j].v[t] / d[j] : val[j].v[t];
          }
        }
      });
      const size_t off = static_cast<size_t>(v) * n_feat + c;
      if (add_self) {
        const Vec<VEC> self = Vec<VEC>::load(x + off);
        const float d = DIV_IN ? div_in[v] : 1.0f;
#pragma unroll
        for (int t = 0; t < VEC; ++t) acc[t] += DIV_IN ? self.v[t] / d : self.v[t];
      }
      Vec<VEC> o;
      const float dv = div_out != nullptr ? div_out[v] : 1.0f;
#pragma unroll
      for (int t = 0; t < VEC; ++t) o.v[t] = div_out != nullptr ? acc[t] / dv : acc[t];
      if (accum != nullptr) {   // the gradient that reached this row by another path (fc_self beside the neighbour term)
        const Vec<VEC> other = Vec<VEC>::load(accum + off);
#pragma unroll
        for (int t = 0; t < VEC; ++t) o.v[t] = other.v[t] + o.v[t];
      }
      if (active) o.store(out + off);
    });
  }
}

}  // namespace
}  // namespace gts

extern "C" int32_t gts_spmm_max_fwd_f32(const int32_t* indptr, const int32_t* indices,
                                        const float* x, float* out, void* arg,
                                        int32_t arg_bytes, int32_t relu_input, int64_t n_dst,
                                        int64_t n_feat, void* stream) {
  using namespace gts;
  if (!indptr || !x || !out || (arg_bytes != 0 && !arg)) return GTS_ERR_NULL;
  if (bad_shape(n_dst, n_feat)) return GTS_ERR_SHAPE;
  if (arg_bytes != 0 && arg_bytes != 1 && arg_bytes != 4) return GTS_ERR_ARGKIND;
  if (n_dst == 0) return GTS_OK;
  const Geometry g = make_geometry(n_dst, n_feat, /*preferred_seq=*/2);
  const int nt = g_spmm_nt < 0 ? 1 : g_spmm_nt;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nd = static_cast<int>(n_dst), nf = static_cast<int>(n_feat);
  GTS_DISPATCH_GEOM(g, {
    if (arg_bytes == 0)
      spmm_max_fwd_kernel<VEC, LPR, 0><<<g.grid, kBlock, 0, st>>>(indptr, indices, x, out, arg, nd, nf, g.seq, nt, relu_input);
    else if (arg_bytes == 1)
      spmm_max_fwd_kernel<VEC, LPR, 1><<<g.grid, kBlock, 0, st>>>(indptr, indices, x, out, arg, nd, nf, g.seq, nt, relu_input);
    else
      spmm_max_fwd_kernel<VEC, LPR, 4><<<g.grid, kBlock, 0, st>>>(indptr, indices, x, out, arg, nd, nf, g.seq, nt, relu_input);
  })
  return launch_status();
}

extern "C" int32_t gts_spmm_max_bwd_f32(const int32_t* t_indptr, const int32_t* t_indices,
                                        const int32_t* t_slot, const float* gout,
                                        const void* arg, int32_t arg_bytes,
                                        const float* relu_src, float* gx, int64_t n_src,
                                        int64_t n_feat, void* stream) {
  using namespace gts;
  if (!t_indptr || !gout || !arg || !gx) return GTS_ERR_NULL;
  if (bad_shape(n_src, n_feat)) return GTS_ERR_SHAPE;
  if (arg_bytes != 1 && arg_bytes != 4) return GTS_ERR_ARGKIND;
  if (n_src == 0) return GTS_OK;
  const Geometry g = make_geometry(n_src, n_feat, /*preferred_seq=*/1);
  const int nt = g_spmm_nt < 0 ? 0 : g_spmm_nt;  // in the training step plain stores of gx are 0.5 % faster (the next GEMM reads it)
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int ns = static_cast<int>(n_src), nf = static_cast<int>(n_feat);
  GTS_DISPATCH_GEOM(g, {
    if (arg_bytes == 1)
      spmm_max_bwd_kernel<VEC, LPR, 1><<<g.grid, kBlock, 0, st>>>(t_indptr, t_indices, t_slot, gout, arg, relu_src, gx, ns, nf, g.seq, nt);
    else
      spmm_max_bwd_kernel<VEC, LPR, 4><<<g.grid, kBlock, 0, st>>>(t_indptr, t_indices, t_slot, gout, arg, relu_src, gx, ns, nf, g.seq, nt);
  })
  return launch_status();
}

extern "C" int32_t gts_spmm_sum_f32(const int32_t* indptr, const int32_t* indices,
                                    const float* x, float* out, const float* div_in,
                                    const float* div_out, const float* accum, int32_t add_self,
                                    int64_t n_out, int64_t n_feat, void* stream) {
  using namespace gts;
  if (!indptr || !x || !out) return GTS_ERR_NULL;
  if (bad_shape(n_out, n_feat)) return GTS_ERR_SHAPE;
  if (n_out == 0) return GTS_OK;
  const Geometry g = make_geometry(n_out, n_feat);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int no = static_cast<int>(n_out), nf = static_cast<int>(n_feat);
  GTS_DISPATCH_GEOM(g, {
    if (div_in != nullptr)
      spmm_sum_kernel<VEC, LPR, true><<<g.grid, kBlock, 0, st>>>(indptr, indices, x, out, div_in, div_out, accum, add_self, no, nf, g.seq);
    else
      spmm_sum_kernel<VEC, LPR, false><<<g.grid, kBlock, 0, st>>>(indptr, indices, x, out, div_in, div_out, accum, add_self, no, nf, g.seq);
  })
  return launch_status();
}
