// K11: dense fp32 layer GEMMs on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32,
// 157 TF peak — the ONLY MFMA use of the path; gfx950 has no xf32/TF32).
//
// One tiled kernel serves the three GEMMs of a layer; what differs is only how each operand
// is laid out with respect to the reduction index kk:
//     C[ra, rb] = sum_kk  A(ra, kk) * B(rb, kk)
//   forward      out[M,N] = act[M,K] . W[N,K]^T (+ second pair) + bias, ReLU
//                A = act  (kk contiguous),  B = W   (kk contiguous)
//   input grad   gin[M,K] = g[M,N] . W[N,K]   (+ second pair)
//                A = g    (kk contiguous),  B = W   (kk strided: B(k, n) = W[n*K + k])
//   weight grad  gw[N,K]  = g[M,N]^T . act[M,K]   (reduction over the 60 000 nodes, split
//                over blockIdx.z into per-split slabs that a second kernel sums in a fixed
//                order -> bitwise reproducible, no float atomics)
//                A = g    (kk strided: A(n, m) = g[m*N + n]),  B = act (kk strided)
//                + the bias gradient (column sums of g) from the A fragments on the way.
//
// Tile: BM x BN outputs per 256-thread workgroup (2x2 waves, each (BM/2)x(BN/2) = TMxTN
// 32x32 MFMA tiles), reduction in steps of 32.  Global -> registers (16 B/lane, issued one
// tile ahead, in flight under the MFMAs) -> LDS (ds_write_b128) -> fragments.  LDS images:
//   kk-contiguous operand: [rows][36]  (32 + 4 pad floats: ds_read_b128 of 4 consecutive kk per
//                          lane is conflict-free for any 16 rows distinct mod 16);
//   kk-strided operand:    [32][rows]  (ds_read_b32, lanes on consecutive addresses).
// The reduction index consumed by MFMA step (g, j) on lane-half h is 8g + 4h + j for both
// operands — a permutation of kk inside each 8-block, free for a sum, chosen so that the
// contiguous operand needs ONE 16-byte LDS read per four MFMAs.
// Two workgroups per CU (<= 256 VGPR, <= 56 KB LDS): while one waits at its barrier the
// other keeps the matrix pipe busy.
#include "gts_common.h"

namespace gts {
namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kBK = 32;        // reduction elements per LDS tile
constexpr int kKcLd = kBK + 4; // padded row of a kk-contiguous LDS image

struct GemmArgs {
  const float* a[2];
  const float* b[2];
  int lda[2], ldb[2];
  int kseg[2];          // reduction length of each (a, b) pair; kseg[1] = 0 when unused
  int ra, rb;           // output rows / cols
  float* c;             // [ra, rb] (or [splits][ra, rb] slabs)
  int ldc;
  const float* bias;    // [rb] or null
  int relu;
  float* colsum;        // [splits][ra] column sums of the strided A operand, or null
  int tiles_per_split;  // reduction tiles handled by one blockIdx.z
};

template <int ROWS, bool KC>
struct OperandTile {
  static constexpr int kFloats = KC ? ROWS * kKcLd : kBK * ROWS;
  static constexpr int kVec = ROWS * kBK / 4 / kBlock;  // float4 per thread per tile
  static_assert(ROWS * kBK / 4 % kBlock == 0, "tile must divide over the workgroup");

  // global -> registers.  `row0` first row of the tile, `k0` first reduction index.
  __device__ __forceinline__ static void load(v4f (&reg)[kVec], const float* __restrict__ p, int ld,
                                              int row0, int k0, int n_rows, int n_k) {
#pragma unroll
    for (int q = 0; q < kVec; ++q) {
      const int idx = threadIdx.x + kBlock * q;
      int r, kk;
      if constexpr (KC) {
        r = idx >> 3, kk = (idx & 7) * 4;            // 8 float4 per 32-wide row
      } else {
        kk = idx / (ROWS / 4), r = (idx % (ROWS / 4)) * 4;
      }
      const int gr = row0 + r, gk = k0 + kk;
      const bool ok = gr < n_rows && gk < n_k;        // dims are multiples of 4: all-or-nothing
      const size_t off = KC ? static_cast<size_t>(gr) * ld + gk : static_cast<size_t>(gk) * ld + gr;
      reg[q] = ok ? *reinterpret_cast<const v4f*>(p + off) : v4f{0.f, 0.f, 0.f, 0.f};
    }
  }

  __device__ __forceinline__ static void store(const v4f (&reg)[kVec], float* lds) {
#pragma unroll
    for (int q = 0; q < kVec; ++q) {
      const int idx = threadIdx.x + kBlock * q;
      int off;
      if constexpr (KC) {
        off = (idx >> 3) * kKcLd + (idx & 7) * 4;
      } else {
        off = (idx / (ROWS / 4)) * ROWS + (idx % (ROWS / 4)) * 4;
      }
      *reinterpret_cast<v4f*>(lds + off) = reg[q];
    }
  }

  // fragment for the 32-row MFMA tile starting at `row` of the image, k-group g:
  // out[j] feeds MFMA step j (reduction index 8g + 4h + j)
  __device__ __forceinline__ static void fragment(float (&out)[4], const float* lds, int row, int g) {
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    if constexpr (KC) {
      const v4f t = *reinterpret_cast<const v4f*>(lds + (row + i) * kKcLd + g * 8 + 4 * h);
      out[0] = t[0], out[1] = t[1], out[2] = t[2], out[3] = t[3];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) out[j] = lds[(g * 8 + 4 * h + j) * ROWS + row + i];
    }
  }
};

template <int BM, int BN, bool AKC, bool BKC>
__global__ __launch_bounds__(kBlock, 2) void gemm_kernel(const GemmArgs p) {
  using TA = OperandTile<BM, AKC>;
  using TB = OperandTile<BN, BKC>;
  constexpr int WTM = BM / 2, WTN = BN / 2;      // wave tile
  constexpr int TM = WTM / 32, TN = WTN / 32;    // MFMA tiles per wave
  __shared__ float lds[TA::kFloats + TB::kFloats];
  float* lds_a = lds;
  float* lds_b = lds + TA::kFloats;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

  const int nt0 = (p.kseg[0] + kBK - 1) / kBK;
  const int nt1 = (p.kseg[1] + kBK - 1) / kBK;
  const int t_beg = blockIdx.z * p.tiles_per_split;
  const int t_end = min(nt0 + nt1, t_beg + p.tiles_per_split);

  v16f acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
  float csum[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) csum[tm] = 0.f;
  const bool want_colsum = !AKC && p.colsum != nullptr && blockIdx.y == 0 && wn == 0;

  v4f ra[TA::kVec], rb[TB::kVec];
  auto fetch = [&](int t) {
    const int s = t >= nt0 ? 1 : 0;
    const int k0 = (s ? t - nt0 : t) * kBK;
    TA::load(ra, p.a[s], p.lda[s], m0, k0, p.ra, p.kseg[s]);
    TB::load(rb, p.b[s], p.ldb[s], n0, k0, p.rb, p.kseg[s]);
  };

  if (t_beg < t_end) {
    fetch(t_beg);
    TA::store(ra, lds_a);
    TB::store(rb, lds_b);
    __syncthreads();
  }
  for (int t = t_beg; t < t_end; ++t) {
    const bool more = t + 1 < t_end;
    if (more) fetch(t + 1);  // in flight under the MFMAs below
#pragma unroll
    for (int g = 0; g < kBK / 8; ++g) {
      float af[TM][4], bf[TN][4];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) TA::fragment(af[tm], lds_a, wm * WTM + tm * 32, g);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) TB::fragment(bf[tn], lds_b, wn * WTN + tn * 32, g);
      if (want_colsum) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) csum[tm] += (af[tm][0] + af[tm][1]) + (af[tm][2] + af[tm][3]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[tm][j], bf[tn][j], acc[tm][tn], 0, 0, 0);
    }
    __syncthreads();  // every wave is done reading this tile
    if (more) {
      TA::store(ra, lds_a);
      TB::store(rb, lds_b);
      __syncthreads();
    }
  }

  // epilogue: C/D layout of the 32x32 MFMA — col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int i = lane & 31, h = lane >> 5;
  float* c = p.c + static_cast<size_t>(blockIdx.z) * p.ra * p.ldc;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = n0 + wn * WTN + tn * 32 + i;
    const float bias = (p.bias != nullptr && col < p.rb) ? p.bias[col] : 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WTM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        float val = acc[tm][tn][r] + bias;
        if (p.relu) val = fmaxf(val, 0.f);
        if (row < p.ra && col < p.rb) c[static_cast<size_t>(row) * p.ldc + col] = val;
      }
    }
  }
  if (want_colsum) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const float total = csum[tm] + __shfl_xor(csum[tm], 32, kWave);  // the two kk halves
      const int row = m0 + wm * WTM + tm * 32 + i;
      if (h == 0 && row < p.ra) p.colsum[static_cast<size_t>(blockIdx.z) * p.ra + row] = total;
    }
  }
}

// out[i] = sum_s slab[s][i] in split order (deterministic); float4 granularity when n % 4 == 0
__global__ __launch_bounds__(kBlock) void reduce_slabs_kernel(const float* __restrict__ slabs,
                                                             float* __restrict__ out, int64_t n,
                                                             int splits) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  float acc = 0.f;
  for (int s = 0; s < splits; ++s) acc += slabs[static_cast<size_t>(s) * n + i];
  out[i] = acc;
}

inline bool aligned4(int64_t x) { return (x & 3) == 0; }

template <bool AKC, bool BKC>
int launch_gemm(const GemmArgs& p, int splits, hipStream_t st, bool square_tile = false) {
  // BN = 256 when the output is wide enough to fill it, else the narrowest tile that covers rb;
  // the split-reduction GEMM uses 128x128 tiles (more output tiles -> fewer, smaller slabs)
  if (square_tile && p.rb > 64) {
    dim3 grid((p.ra + 127) / 128, (p.rb + 127) / 128, splits);
    gemm_kernel<128, 128, AKC, BKC><<<grid, kBlock, 0, st>>>(p);
  } else if (p.rb > 128) {
    dim3 grid((p.ra + 127) / 128, (p.rb + 255) / 256, splits);
    gemm_kernel<128, 256, AKC, BKC><<<grid, kBlock, 0, st>>>(p);
  } else if (p.rb > 64) {
    dim3 grid((p.ra + 127) / 128, 1, splits);
    gemm_kernel<128, 128, AKC, BKC><<<grid, kBlock, 0, st>>>(p);
  } else {
    dim3 grid((p.ra + 127) / 128, 1, splits);
    gemm_kernel<128, 64, AKC, BKC><<<grid, kBlock, 0, st>>>(p);
  }
  return launch_status();
}

}  // namespace
}  // namespace gts

extern "C" int32_t gts_linear_fwd_f32(const float* a0, const float* w0, const float* a1,
                                      const float* w1, const float* bias, float* out, int64_t m,
                                      int64_t n, int64_t k0, int64_t k1, int32_t relu,
                                      void* stream) {
  using namespace gts;
  if (!a0 || !w0 || !out || ((a1 == nullptr) != (w1 == nullptr))) return GTS_ERR_NULL;
  if (m < 0 || n <= 0 || k0 <= 0 || k1 < 0 || m >= (1LL << 31) || n >= (1 << 20) ||
      k0 >= (1 << 20) || k1 >= (1 << 20) || !aligned4(k0) || !aligned4(k1) || (a1 && k1 == 0))
    return GTS_ERR_SHAPE;
  if (m == 0) return GTS_OK;
  GemmArgs p{};
  p.a[0] = a0, p.b[0] = w0, p.lda[0] = static_cast<int>(k0), p.ldb[0] = static_cast<int>(k0);
  p.kseg[0] = static_cast<int>(k0);
  p.a[1] = a1 ? a1 : a0, p.b[1] = w1 ? w1 : w0;
  p.lda[1] = p.ldb[1] = static_cast<int>(k1), p.kseg[1] = a1 ? static_cast<int>(k1) : 0;
  p.ra = static_cast<int>(m), p.rb = static_cast<int>(n), p.c = out, p.ldc = static_cast<int>(n);
  p.bias = bias, p.relu = relu, p.colsum = nullptr;
  p.tiles_per_split = (p.kseg[0] + kBK - 1) / kBK + (p.kseg[1] + kBK - 1) / kBK;
  return launch_gemm<true, true>(p, 1, static_cast<hipStream_t>(stream));
}

extern "C" int32_t gts_linear_bwd_input_f32(const float* g0, const float* w0, const float* g1,
                                            const float* w1, float* gin, int64_t m, int64_t k,
                                            int64_t n0, int64_t n1, void* stream) {
  using namespace gts;
  if (!g0 || !w0 || !gin || ((g1 == nullptr) != (w1 == nullptr))) return GTS_ERR_NULL;
  if (m < 0 || k <= 0 || n0 <= 0 || n1 < 0 || m >= (1LL << 31) || k >= (1 << 20) ||
      n0 >= (1 << 20) || n1 >= (1 << 20) || !aligned4(k) || !aligned4(n0) || !aligned4(n1) ||
      (g1 && n1 == 0))
    return GTS_ERR_SHAPE;
  if (m == 0) return GTS_OK;
  GemmArgs p{};
  // C[m, k] = sum_n g[m, n] * W[n, k]:  A = g (reduction contiguous), B(k, n) = W[n*K + k]
  p.a[0] = g0, p.b[0] = w0, p.lda[0] = static_cast<int>(n0), p.ldb[0] = static_cast<int>(k);
  p.kseg[0] = static_cast<int>(n0);
  p.a[1] = g1 ? g1 : g0, p.b[1] = w1 ? w1 : w0;
  p.lda[1] = static_cast<int>(n1), p.ldb[1] = static_cast<int>(k);
  p.kseg[1] = g1 ? static_cast<int>(n1) : 0;
  p.ra = static_cast<int>(m), p.rb = static_cast<int>(k), p.c = gin, p.ldc = static_cast<int>(k);
  p.bias = nullptr, p.relu = 0, p.colsum = nullptr;
  p.tiles_per_split = (p.kseg[0] + kBK - 1) / kBK + (p.kseg[1] + kBK - 1) / kBK;
  return launch_gemm<true, false>(p, 1, static_cast<hipStream_t>(stream));
}

extern "C" int64_t gts_linear_bwd_weight_workspace(int64_t m, int64_t n, int64_t k) {
  using namespace gts;
  if (m <= 0 || n <= 0 || k <= 0) return 0;
  const int64_t tiles = (m + kBK - 1) / kBK;
  const int64_t out_tiles = ((n + 127) / 128) * ((k + 127) / 128);
  int64_t splits = (512 + out_tiles - 1) / out_tiles;        // ~2 workgroups per CU in flight
  if (splits > tiles) splits = tiles;
  if (splits < 1) splits = 1;
  return splits * (n * k + n) * static_cast<int64_t>(sizeof(float));
}

extern "C" int32_t gts_linear_bwd_weight_f32(const float* g, const float* a, float* gw, float* gb,
                                             float* workspace, int64_t workspace_bytes, int64_t m,
                                             int64_t n, int64_t k, void* stream) {
  using namespace gts;
  if (!g || !a || !gw || !workspace) return GTS_ERR_NULL;
  if (m <= 0 || n <= 0 || k <= 0 || m >= (1LL << 31) || n >= (1 << 20) || k >= (1 << 20) ||
      !aligned4(n) || !aligned4(k))
    return GTS_ERR_SHAPE;
  const int64_t need = gts_linear_bwd_weight_workspace(m, n, k);
  if (workspace_bytes < need) return GTS_ERR_SHAPE;
  const int splits = static_cast<int>(need / ((n * k + n) * static_cast<int64_t>(sizeof(float))));
  const int tiles = static_cast<int>((m + kBK - 1) / kBK);
  hipStream_t st = static_cast<hipStream_t>(stream);
  GemmArgs p{};
  // C[n, k] = sum_m g[m, n] * act[m, k]: both operands reduction-strided
  p.a[0] = g, p.b[0] = a, p.lda[0] = static_cast<int>(n), p.ldb[0] = static_cast<int>(k);
  p.kseg[0] = static_cast<int>(m);
  p.a[1] = g, p.b[1] = a, p.lda[1] = p.lda[0], p.ldb[1] = p.ldb[0], p.kseg[1] = 0;
  p.ra = static_cast<int>(n), p.rb = static_cast<int>(k);
  p.c = workspace, p.ldc = static_cast<int>(k);
  p.bias = nullptr, p.relu = 0;
  p.colsum = workspace + static_cast<size_t>(splits) * n * k;
  p.tiles_per_split = (tiles + splits - 1) / splits;
  int rc = launch_gemm<false, false>(p, splits, st, /*square_tile=*/true);
  if (rc != GTS_OK) return rc;
  const int64_t nk = n * k;
  reduce_slabs_kernel<<<static_cast<unsigned>((nk + kBlock - 1) / kBlock), kBlock, 0, st>>>(workspace, gw, nk, splits);
  if (gb != nullptr)
    reduce_slabs_kernel<<<static_cast<unsigned>((n + kBlock - 1) / kBlock), kBlock, 0, st>>>(p.colsum, gb, n, splits);
  return launch_status();
}
