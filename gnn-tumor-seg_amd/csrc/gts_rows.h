// Row-gather skeleton shared by the CSR kernels (spmm max/sum, GAT): lane groups, neighbour
// chunks, launch geometry and the (VEC, LPR) dispatch.
#pragma once
#include <type_traits>

#include "gts_common.h"

namespace gts {

constexpr int kUnroll = 8;

// w * v + acc in ONE instruction and one rounding (v_fma_f32; the library is built with -ffp-contract=off, so this is the only
// place products and sums fuse).  GATConv's weighted sums and dot products (K5 - K8, plain and clustered kernels alike, so the
// two stay bit-identical): the clustered kernels are bound by the vector instructions they issue, and a separate multiply and
// add per element were half of a neighbour's cost (profiles/r04).  DGL's own CPU kernel leaves the contraction to its compiler;
// the oracle comparison holds at the stated tolerance either way (one rounding instead of two per term).
__device__ __forceinline__ float mad(float w, float v, float acc) { return __builtin_fmaf(w, v, acc); }

// exp(x) - 1 for the negative branch of ELU (GATConv's activation, model/networks.py:52 -> F.elu; torch computes expm1), x <= 0:
//   x >= -0.35: x + x^2 (1/2 + x/6 + ... + x^5 / 5040) — the series, cut where its next term is a quarter of an ulp of x;
//   x <  -0.35: 2^(x log2 e) - 1 on v_exp_f32 (the result is <= -0.29, the subtraction costs at most a bit);
// about 2 ulp from expm1 at worst (tests/test_gpu_kernels.py compares with torch's at every magnitude), -inf -> -1, NaN -> NaN.
// The clustered forward kernel issues vector instructions for longer than its memory traffic takes (profiles/r04), and ocml's
// expm1f was half of them: this form is a third as long and pairs up on v_pk_fma_f32 (`elu_expm1_pair`: the same operations on two
// elements, so the plain kernels' scalar form and the clustered kernels' paired form give the same bits).
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr float kEluSplit = -0.35f, kLog2e = 1.44269504088896340736f;
constexpr float kEluC2 = 0.5f, kEluC3 = 1.0f / 6.0f, kEluC4 = 1.0f / 24.0f, kEluC5 = 1.0f / 120.0f, kEluC6 = 1.0f / 720.0f,
                kEluC7 = 1.0f / 5040.0f;
__device__ __forceinline__ float elu_expm1(float x) {
  float q = __builtin_fmaf(kEluC7, x, kEluC6);
  q = __builtin_fmaf(q, x, kEluC5);
  q = __builtin_fmaf(q, x, kEluC4);
  q = __builtin_fmaf(q, x, kEluC3);
  q = __builtin_fmaf(q, x, kEluC2);
  const float small = __builtin_fmaf(x * x, q, x);
  const float large = __builtin_amdgcn_exp2f(x * kLog2e) - 1.0f;
  return x < kEluSplit ? large : small;
}
__device__ __forceinline__ v2f elu_expm1_pair(v2f x) {
  v2f q = __builtin_elementwise_fma(v2f{kEluC7, kEluC7}, x, v2f{kEluC6, kEluC6});
  q = __builtin_elementwise_fma(q, x, v2f{kEluC5, kEluC5});
  q = __builtin_elementwise_fma(q, x, v2f{kEluC4, kEluC4});
  q = __builtin_elementwise_fma(q, x, v2f{kEluC3, kEluC3});
  q = __builtin_elementwise_fma(q, x, v2f{kEluC2, kEluC2});
  const v2f small = __builtin_elementwise_fma(x * x, q, x);
  const v2f arg = x * v2f{kLog2e, kLog2e};
  const v2f large = v2f{__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)} - v2f{1.0f, 1.0f};
  return v2f{x.x < kEluSplit ? large.x : small.x, x.y < kEluSplit ? large.y : small.y};
}

// Row owned by this lane group for sequential step s; -1 when past the end.
template <int LPR>
__device__ __forceinline__ int owned_row(int s, int seq, int n_rows) {
  constexpr int kRowsPerWave = kWave / LPR;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int tile = xcd_contiguous_tile(blockIdx.x, gridDim.x);
  int v = ((tile * kWavesPerBlock + wave) * seq + s) * kRowsPerWave + lane / LPR;
  if constexpr (LPR == kWave) v = __builtin_amdgcn_readfirstlane(v);
  return v < n_rows ? v : -1;
}

// Column walk of one row.  The trip count is the same for every lane (wave-uniform for
// LPR == 64, where the chunk code below broadcasts indices with v_readlane and therefore
// needs all 64 lanes alive); a lane whose columns lie past the end runs on column 0 and
// must not store (`active` false).
template <int VEC, int LPR, typename Body>
__device__ __forceinline__ void for_columns(int n_feat, Body&& body) {
  const int gl = (threadIdx.x & (kWave - 1)) % LPR;
  for (int c0 = 0; c0 < n_feat; c0 += LPR * VEC) {
    const int c = c0 + gl * VEC;
    const bool active = c < n_feat;
    body(active ? c : 0, active);
  }
}

// Neighbour chunks.  A row's in-edges are consumed in chunks of <= kUnroll so that all of a
// chunk's row loads are in flight together.  Every chunk body is straight-line code:
//  * LPR == 64 (row is wave-uniform): the exact count is a template parameter (switch on
//    the remainder), lanes 0..CNT-1 fetch the chunk's indices with ONE vector load and each
//    index is broadcast with v_readlane.  (Per-index s_load, or loads under `if (j < cnt)`,
//    make hipcc wait vmcnt(0) in front of every row load.)
//  * LPR < 64 (several rows per wave, divergent): always kUnroll entries, positions clamped
//    to the row's last edge and the update guarded by `valid(j)`.
template <int N>
using IC = std::integral_constant<int, N>;

template <int LPR, typename Body>
__device__ __forceinline__ void for_chunks(int beg, int end, Body&& body) {
  if constexpr (LPR == kWave) {
    int k = beg;
    for (; k + kUnroll <= end; k += kUnroll) body(IC<kUnroll>{}, k);
    switch (end - k) {
      case 1: body(IC<1>{}, k); break;
      case 2: body(IC<2>{}, k); break;
      case 3: body(IC<3>{}, k); break;
      case 4: body(IC<4>{}, k); break;
      case 5: body(IC<5>{}, k); break;
      case 6: body(IC<6>{}, k); break;
      case 7: body(IC<7>{}, k); break;
      default: break;
    }
  } else {
    for (int k = beg; k < end; k += kUnroll) body(IC<kUnroll>{}, k);
  }
}

// entry j of the chunk starting at edge position k (of an int32 per-edge array)
template <int LPR, int CNT>
struct Chunk {
  int reg;  // LPR == 64: lane l holds entry l
  const int32_t* __restrict__ p;
  int k, last;
  __device__ __forceinline__ Chunk(const int32_t* __restrict__ arr, int k_, int end) : p(arr), k(k_), last(end - 1) {
    if constexpr (LPR == kWave) {
      const int lane = threadIdx.x & (kWave - 1);
      reg = lane < CNT ? arr[k_ + lane] : 0;
    } else {
      reg = 0;
    }
  }
  __device__ __forceinline__ int operator[](int j) const {
    if constexpr (LPR == kWave) {
      return __builtin_amdgcn_readlane(reg, j);
    } else {
      return p[min(k + j, last)];
    }
  }
  __device__ __forceinline__ bool valid(int j) const {
    if constexpr (LPR == kWave) return true;
    return k + j <= last;
  }
};

// rows each wave walks sequentially: enough tiles to fill 256 CUs several times over,
// few enough that index prefetch amortises.
// Tuning knobs (gts_set_option).  Defaults come from tools/tune_spmm.py on 4 x 15k-node lattice
// graphs at F = 256 (profiles/r01_tune_spmm.log): K1 is fastest with 2 rows per wave and streaming
// (non-temporal) stores of out/argmax; K2 with 1 row per wave.  K2's streaming stores win 2 % in
// isolation but lose 0.5 % inside the training step (the next GEMM reads gx), streaming LOADS of its
// relu_src rows cost 40 % (bit 1 of the knob): K2 keeps ordinary loads and stores.
inline int g_spmm_seq = 0;   // rows per wave; 0 = the kernel's own default
inline int g_project_nt = 1;  // K12: non-temporal stores of the projected rows
inline int g_spmm_nt = -1;   // streaming stores/loads of write-once / read-once rows; -1 = default
inline int g_cluster_nt = -1;  // clustered K1 / K2 (gts_spmm_cluster.hip): bit 0 = streaming stores of out / gx; -1 = default
inline int g_gat_walk = 1;           // K5-K8: 1 = head-major walk over the (node, head) rows, 0 = node-major
inline int g_cluster_ring = 0;       // form 0: units the gathers run ahead (0 = automatic: 2 if it fits half a CU's LDS, else 1); form 2: ring slots
inline int g_cluster_per_cu = 0;     // persistent workgroups per CU (0 = automatic)
inline int g_gat_cluster_waves = 0;  // clustered GAT kernels: waves per workgroup (0 = default)
inline int g_cluster_dealing = 0;      // clustered K1 / K2: 0 = automatic, 1 = units dealt off the caller's counters, 2 = static round-robin
inline int g_gat_cluster_dealing = 0;  // clustered GAT kernels: 0 = units dealt off a counter per XCD (default), 1 = static round-robin
inline int g_gat_cluster_group = 0;  // clustered GAT kernels: clusters walked together through all their slices (0 = the whole span of an XCD)
inline int g_cluster_consumers = 0;  // form 0: waves per workgroup; form 2: consumer waves (0 = automatic)

inline int pick_seq(int64_t n_rows, int rows_per_wave_step, int preferred) {
  if (g_spmm_seq > 0) return g_spmm_seq;
  if (preferred > 0) return preferred;
  const int64_t steps = (n_rows + rows_per_wave_step - 1) / rows_per_wave_step;
  // aim for >= 8192 waves (256 CUs x 32) before lengthening the per-wave walk
  int seq = 1;
  while (seq < 4 && steps / (seq * 2) >= 8192) seq *= 2;
  return seq;
}

struct Geometry {
  int vec, lpr, seq;
  dim3 grid;
};

inline Geometry make_geometry(int64_t n_rows, int64_t n_feat, int preferred_seq = 0) {
  Geometry g;
  g.vec = (n_feat % 4 == 0) ? 4 : 1;
  g.lpr = lanes_per_row(n_feat / g.vec);
  const int rows_per_step = kWave / g.lpr;
  g.seq = pick_seq(n_rows, rows_per_step, preferred_seq);
  const int64_t rows_per_block = static_cast<int64_t>(rows_per_step) * g.seq * kWavesPerBlock;
  g.grid = dim3(static_cast<unsigned>((n_rows + rows_per_block - 1) / rows_per_block));
  return g;
}

inline bool bad_shape(int64_t n_rows, int64_t n_feat) {
  return n_rows < 0 || n_feat <= 0 || n_rows >= (1LL << 31) || n_feat >= (1LL << 24);
}

}  // namespace gts

// Dispatch on (VEC, LPR): LPR in {1,2,4,...,64} for VEC=4 and VEC=1.
#define GTS_DISPATCH_LPR(VEC_, LPR_VAL, ...)                \
  switch (LPR_VAL) {                                        \
    case 1: { constexpr int LPR = 1; __VA_ARGS__; } break;  \
    case 2: { constexpr int LPR = 2; __VA_ARGS__; } break;  \
    case 4: { constexpr int LPR = 4; __VA_ARGS__; } break;  \
    case 8: { constexpr int LPR = 8; __VA_ARGS__; } break;  \
    case 16: { constexpr int LPR = 16; __VA_ARGS__; } break; \
    case 32: { constexpr int LPR = 32; __VA_ARGS__; } break; \
    default: { constexpr int LPR = 64; __VA_ARGS__; } break; \
  }

#define GTS_DISPATCH_GEOM(G, ...)                                   \
  if ((G).vec == 4) {                                               \
    constexpr int VEC = 4;                                          \
    GTS_DISPATCH_LPR(4, (G).lpr, __VA_ARGS__)                       \
  } else {                                                          \
    constexpr int VEC = 1;                                          \
    GTS_DISPATCH_LPR(1, (G).lpr, __VA_ARGS__)                       \
  }

