// K11: dense fp32 layer GEMMs on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32,
// 157 TF peak — the ONLY MFMA use of the path; gfx950 has no xf32/TF32).
//
// One tiled kernel serves the three GEMMs of a layer; what differs is only how each operand
// is laid out with respect to the reduction index kk:
//     C[ra, rb] = sum_kk  A(ra, kk) * B(rb, kk)
//   forward      out[M,N] = act[M,K] . W[N,K]^T (+ second pair) + bias, ReLU
//                A = act  (kk contiguous),  B = W   (kk contiguous)
//   input grad   gin[M,K] = g[M,N] . W[N,K]   (+ second pair), optional ReLU mask on gin
//                A = g    (kk contiguous),  B = W   (kk strided: B(k, n) = W[n*K + k])
//   weight grad  gw[N,K]  = g[M,N]^T . act[M,K]   (reduction over the 60 000 nodes, split
//                over blockIdx.z into per-split slabs that a second kernel sums in a fixed
//                order -> bitwise reproducible, no float atomics); up to 4 same-shape
//                problems share one launch (the three weight gradients of a SAGE layer);
//                A = g    (kk strided: A(n, m) = g[m*N + n]),  B = act (kk strided)
//                + the bias gradient (column sums of g) from the A fragments on the way.
//
// Tile: BM x BN outputs per workgroup of WM x WN waves, each wave (BM/WM)x(BN/WN) = TMxTN
// 32x32 MFMA tiles, reduction in steps of 32.  Global -> registers (16 B/lane, issued one
// tile ahead, in flight under the MFMAs) -> LDS (ds_write_b128) -> fragments.  LDS images:
//   kk-contiguous operand: [rows][36]  (32 + 4 pad floats: ds_read_b128 of 4 consecutive kk per
//                          lane is conflict-free for any 16 rows distinct mod 16);
//   kk-strided operand:    [32][rows]  (ds_read_b32, lanes on consecutive addresses).
// The reduction index consumed by MFMA step (g, j) on lane-half h is 8g + 4h + j for both
// operands — a permutation of kk inside each 8-block, free for a sum, chosen so that the
// contiguous operand needs ONE 16-byte LDS read per four MFMAs.
#include <type_traits>

#include "gts_rows.h"

namespace gts {
namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kBK = 32;         // reduction elements per LDS tile
constexpr int kKcLd = kBK + 4;  // padded row of a kk-contiguous LDS image
constexpr int kMaxProblems = 32;
int g_gemm_sched = 1;           // GemmArgs::sched (GTS_OPT_GEMM_SCHED); bit 0 on: -0.9 % on the 19-problem weight-gradient launch

struct GemmArgs {
  const float* a[2];    // forward / input grad: the two (a, b) reduction segments
  const float* b[2];
  int lda[2], ldb[2];
  int kseg[2];          // reduction length of each segment; kseg[1] = 0 when unused
  int ra, rb;           // output rows / cols
  float* c;             // [ra, rb]  (weight grad: slabs [problem][split][ra, rb])
  int ldc;
  const float* bias;    // [rb] or null
  int relu;
  const float* mask;    // [ra, rb] or null: output zeroed where mask <= 0 (fused ReLU backward)
  // split-reduction (weight gradient) only
  const float* pa[kMaxProblems];  // per-problem operands (same shapes)
  const float* pb[kMaxProblems];
  float* colsum;        // [problem][split][ra] column sums of A, or null
  unsigned colsum_mask; // bit q: problem q wants its column sums (wgrad_stream_kernel; the other kernels sum them all)
  int n_problems;       // 0 -> plain GEMM
  int tiles_n;          // output tiles along rb per problem
  int n_splits;
  int tiles_per_split;  // reduction tiles handled by one blockIdx.z
  int sched;            // tuning bits (GTS_OPT_GEMM_SCHED): 1 = waves further into a tile yield MFMA issue
  // chained second GEMM of the panel kernels (c2 != null): c2[ra, rb2] = act2(c[ra, rb] . b2[rb2, rb]^T + bias2),
  // computed by the workgroup that has just produced those rows of c (rb <= 256: one workgroup per row panel)
  const float* b2;
  const float* bias2;
  float* c2;
  int rb2, ldb2, ldc2, relu2;
  // attention scores riding in the epilogue of the panel kernels (GATConv: el / er = <ft[n, h, :], attn_l/r[h, :]>):
  // per output row and 64-column block the partial dot products with sc_l / sc_r [rb] go to sc_el / sc_er [ra, rb / 64]
  const float* sc_l;
  const float* sc_r;
  float* sc_el;
  float* sc_er;
  // ReLU masks as bits (layout: gts_relu_bits_bytes in gts_hip.h).  bits_out: c > 0 is recorded while c is stored;
  // bits_in: the same mask as `mask`, read by the kernels that can (the others read the floats of `mask`)
  unsigned long long* bits_out;
  const unsigned long long* bits_in;
  // the activation backward of the layer BELOW riding in an input gradient's epilogue (panel kernels only; GATConv stacks):
  // mask_kind 1 = ELU through its output: c *= mask > 0 ? 1 : mask + 1 (mask = that layer's output, read as floats);
  // col_partial [row blocks][rb]: column sums of the rows each wave stored (the bias gradient of the layer below, summed
  // over the row blocks in fixed order by sum_chunks)
  int mask_kind;
  float* col_partial;
  // the same weights in FRAGMENT ORDER (gts_pack_weights_f32; panel kernels only, null = read b / b2 as they are):
  // bp[seg] for b[seg], bp2 for b2.  One buffer_load_dwordx4 of a 16-row weight fragment then reads 1 KiB of
  // consecutive bytes (16 accesses of the vector L1) instead of 64 pieces of 16 bytes 1 KiB apart (64 accesses).
  const float* bp[2];
  const float* bp2;
};

// Phase probe of the kernel (start / operands staged / main loop done / tile stored).  The
// library only ever instantiates NoProbe, which compiles to nothing; tools/diag/gemm_probe.hip
// includes this file and instantiates the same kernels with a probe that records timestamps.
struct NoProbe {
  __device__ __forceinline__ static void mark(int /*phase*/) {}
};

template <int ROWS, bool KC, int THREADS>
struct OperandTile {
  static constexpr int kFloats = KC ? ROWS * kKcLd : kBK * ROWS;
  static constexpr int kVec = ROWS * kBK / 4 / THREADS;  // float4 per thread per tile
  static_assert(ROWS * kBK / 4 % THREADS == 0 && kVec >= 1, "tile must divide over the workgroup");

  // global -> registers.  `row0` first row of the tile, `k0` first reduction index.
  __device__ __forceinline__ static void load(v4f (&reg)[kVec], const float* __restrict__ p, int ld,
                                              int row0, int k0, int n_rows, int n_k) {
#pragma unroll
    for (int q = 0; q < kVec; ++q) {
      const int idx = threadIdx.x + THREADS * q;
      int r, kk;
      if constexpr (KC) {
        r = idx >> 3, kk = (idx & 7) * 4;  // 8 float4 per 32-wide row
      } else {
        kk = idx / (ROWS / 4), r = (idx % (ROWS / 4)) * 4;
      }
      const int gr = row0 + r, gk = k0 + kk;
      const bool ok = gr < n_rows && gk < n_k;  // dims are multiples of 4: all-or-nothing
      const size_t off = KC ? static_cast<size_t>(gr) * ld + gk : static_cast<size_t>(gk) * ld + gr;
      reg[q] = ok ? *reinterpret_cast<const v4f*>(p + off) : v4f{0.f, 0.f, 0.f, 0.f};
    }
  }

  __device__ __forceinline__ static void store(const v4f (&reg)[kVec], float* lds) {
#pragma unroll
    for (int q = 0; q < kVec; ++q) {
      const int idx = threadIdx.x + THREADS * q;
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

// Epilogue shared by the GEMM kernels.  C/D layout of the 32x32 MFMA: col = lane & 31,
// row = (r&3) + 8*(r>>2) + 4*(lane>>5).  `lds` is the (now dead) operand area.
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void write_tile(const GemmArgs& p, float* lds, float* c,
                                           v16f (&acc)[BM / WM / 32][BN / WN / 32], int m0, int n0) {
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int i = lane & 31, h = lane >> 5;
  if ((p.rb & 3) == 0 && (p.ldc & 3) == 0) {
    // Wide path: each 32x32 accumulator tile goes through a per-wave [32][36] LDS patch (the
    // operand images are dead after the loop's last barrier) and leaves as 16-byte-per-lane row
    // segments: 4x fewer store instructions, and bias / ReLU mask arrive as float4 too.
    float* stage = lds + wave * (32 * kKcLd);
    const int srow = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = n0 + wn * WTN + tn * 32 + c4;
      const bool col_ok = col < p.rb;
      v4f bias = {0.f, 0.f, 0.f, 0.f};
      if (p.bias != nullptr && col_ok) bias = *reinterpret_cast<const v4f*>(p.bias + col);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        // the ReLU mask of this tile is requested first: its latency hides behind the LDS staging
        v4f mk[4];
        if (p.mask != nullptr) {
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int row = m0 + wm * WTM + tm * 32 + it * 8 + srow;
            mk[it] = (row < p.ra && col_ok)
                         ? *reinterpret_cast<const v4f*>(p.mask + static_cast<size_t>(row) * p.ldc + col)
                         : v4f{0.f, 0.f, 0.f, 0.f};
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * h) * kKcLd + i] = acc[tm][tn][r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int lrow = it * 8 + srow;
          v4f val = *reinterpret_cast<const v4f*>(stage + lrow * kKcLd + c4) + bias;
          const int row = m0 + wm * WTM + tm * 32 + lrow;
          if (row < p.ra && col_ok) {
            const size_t off = static_cast<size_t>(row) * p.ldc + col;
            if (p.relu) {
#pragma unroll
              for (int e = 0; e < 4; ++e) val[e] = fmaxf(val[e], 0.f);
            }
            if (p.mask != nullptr) {
#pragma unroll
              for (int e = 0; e < 4; ++e) val[e] = mk[it][e] > 0.f ? val[e] : 0.f;
            }
            *reinterpret_cast<v4f*>(c + off) = val;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  } else {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = n0 + wn * WTN + tn * 32 + i;
      const float bias = (p.bias != nullptr && col < p.rb) ? p.bias[col] : 0.f;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * WTM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (row < p.ra && col < p.rb) {
            const size_t off = static_cast<size_t>(row) * p.ldc + col;
            float val = acc[tm][tn][r] + bias;
            if (p.relu) val = fmaxf(val, 0.f);
            if (p.mask != nullptr) val = p.mask[off] > 0.f ? val : 0.f;
            c[off] = val;
          }
        }
      }
    }
  }
}

// waves per SIMD to plan registers for: two co-resident workgroups when the accumulators allow
constexpr int min_waves_per_simd(int wm, int wn, int tm, int tn) {
  const int per_block = wm * wn / 4;                       // waves per SIMD of one workgroup
  return per_block * ((tm * tn * 16 <= 64 || per_block == 1) ? 2 : 1);
}

// DB = false: one LDS image per operand, two barriers per reduction tile; meant for two
//   co-resident workgroups per CU that fill each other's bubbles.
// DB = true:  two images and ONE barrier per tile: while the waves multiply tile t out of image
//   t&1, tile t+1 (already in registers) is written to the other image and tile t+2 is requested
//   from memory, so a workgroup that is alone on its CU (256 x 256 tiles, 16 waves: one round
//   over the 60 000-row matrices) keeps its matrix cores fed without a partner.
template <int BM, int BN, int WM, int WN, bool AKC, bool BKC, bool DB = false, class Probe = NoProbe>
__global__ __launch_bounds__(64 * WM * WN, DB ? WM * WN / 4 : min_waves_per_simd(WM, WN, BM / WM / 32, BN / WN / 32))
void gemm_kernel(const GemmArgs p) {
  constexpr int THREADS = 64 * WM * WN;
  using TA = OperandTile<BM, AKC, THREADS>;
  using TB = OperandTile<BN, BKC, THREADS>;
  constexpr int WTM = BM / WM, WTN = BN / WN;    // wave tile
  constexpr int TM = WTM / 32, TN = WTN / 32;    // MFMA tiles per wave
  static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one 32x32 MFMA tile");
  constexpr int kImage = TA::kFloats + TB::kFloats;
  constexpr int kOperandFloats = (DB ? 2 : 1) * kImage;
  constexpr int kStageFloats = WM * WN * 32 * kKcLd;  // epilogue patches, one per wave
  __shared__ float lds[kOperandFloats > kStageFloats ? kOperandFloats : kStageFloats];
  const float* lds_a = lds;
  const float* lds_b = lds + TA::kFloats;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int problem = p.n_problems ? blockIdx.y / p.tiles_n : 0;
  const int tile_n = p.n_problems ? blockIdx.y % p.tiles_n : blockIdx.y;
  const int m0 = blockIdx.x * BM, n0 = tile_n * BN;
  const float* a_first = p.a[0];
  const float* b_first = p.b[0];
  if (p.n_problems) {
    a_first = kernarg_entry<const float*>(offsetof(GemmArgs, pa), problem);
    b_first = kernarg_entry<const float*>(offsetof(GemmArgs, pb), problem);
  }

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
  // the wave that sums the A columns of row block wm: one per SIMD where the wave grid is square (wave w runs on SIMD w % 4,
  // so `wn == 0` would put the four of them — and their vector adds, which the f32 matrix pipe does not overlap — on SIMD 0)
  const bool want_colsum = !AKC && p.colsum != nullptr && tile_n == 0 && wn == (WM == WN ? wm : 0);

  v4f ra[TA::kVec], rb[TB::kVec];
  auto fetch_a = [&](v4f (&dst)[TA::kVec], int t) {
    const bool second = t >= nt0;
    TA::load(dst, second ? p.a[1] : a_first, second ? p.lda[1] : p.lda[0], m0,
             (second ? t - nt0 : t) * kBK, p.ra, second ? p.kseg[1] : p.kseg[0]);
  };
  auto fetch_b = [&](int t) {
    const bool second = t >= nt0;
    TB::load(rb, second ? p.b[1] : b_first, second ? p.ldb[1] : p.ldb[0], n0,
             (second ? t - nt0 : t) * kBK, p.rb, second ? p.kseg[1] : p.kseg[0]);
  };
  auto compute = [&]() {
#pragma unroll
    for (int g = 0; g < kBK / 8; ++g) {
      if (DB && (p.sched & 1)) {   // the builtin wants a literal
        if (g == 0) __builtin_amdgcn_s_setprio(3);
        else if (g == 1) __builtin_amdgcn_s_setprio(2);
        else if (g == 2) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
      }
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
  };
  Probe::mark(0);
  auto stash = [&](int image) {
    TA::store(ra, lds + image * kImage);
    TB::store(rb, lds + image * kImage + TA::kFloats);
  };
  if (t_beg < t_end) {
    fetch_a(ra, t_beg);
    fetch_b(t_beg);
    stash(0);
    if (DB && t_beg + 1 < t_end) {
      fetch_a(ra, t_beg + 1);
      fetch_b(t_beg + 1);
    }
    __syncthreads();
  }
  Probe::mark(1);
  if constexpr (DB) {
    for (int t = t_beg; t < t_end; ++t) {
      const int cur = (t - t_beg) & 1;
      if (t + 1 < t_end) stash(cur ^ 1);  // tile t+1: requested one iteration ago
      if (t + 2 < t_end) {                // lands under the MFMAs below
        fetch_a(ra, t + 2);
        fetch_b(t + 2);
      }
      lds_a = lds + cur * kImage;
      lds_b = lds_a + TA::kFloats;
      compute();
      __syncthreads();  // image cur^1 complete for the next tile; everyone is done reading image cur
    }
  } else {
    for (int t = t_beg; t < t_end; ++t) {
      const bool more = t + 1 < t_end;
      if (more) {  // in flight under the MFMAs below
        fetch_a(ra, t + 1);
        fetch_b(t + 1);
      }
      compute();
      __syncthreads();  // every wave is done reading this tile
      if (more) {
        stash(0);
        __syncthreads();
      }
    }
  }

  Probe::mark(2);
  const size_t slab = p.n_problems ? static_cast<size_t>(problem) * p.n_splits + blockIdx.z : 0;
  write_tile<BM, BN, WM, WN>(p, lds, p.c + slab * p.ra * p.ldc, acc, m0, n0);
  Probe::mark(3);
  if (want_colsum) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const float total = csum[tm] + __shfl_xor(csum[tm], 32, kWave);  // the two kk halves
      const int row = m0 + wm * WTM + tm * 32 + (lane & 31);
      if ((lane >> 5) == 0 && row < p.ra) p.colsum[slab * p.ra + row] = total;
    }
  }
}

// ---- 240-row panels on v_mfma_f32_16x16x4_f32 -------------------------------------------------
// The layer GEMMs of the path have 60 000 (or 120 000) rows and 256 columns: with 256-row tiles
// that is 235 workgroups for 256 CUs — 21 CUs idle and every busy CU carrying 256 rows where
// 234.4 would do.  Row panels of 240 = 15 x 16 rows make it 250 workgroups of 240 rows (60 000 =
// 250 x 240 exactly): -6 % rows per CU.  240 is not a multiple of 32, so this kernel is built
// on the 16x16x4 MFMA (same flops per cycle as 32x32x2, exact fp32): 12 waves (3 x 4), one
// wave = 80 x 64 outputs = 5 x 4 tiles, 3 waves per SIMD, 5 x 4 x 4 = 80 accumulator registers.
// Forward form only (both operands reduction-contiguous; input gradients reach it through
// transposed weights), two LDS images per operand and one barrier per reduction tile like the
// double-buffered 256 x 256 tile.  Reduction index consumed by MFMA step (g, j) on lane
// quarter q: 16 g + 4 q + j for both operands (one ds_read_b128 per operand tile and 4 MFMAs).
typedef float v4acc __attribute__((ext_vector_type(4)));

constexpr int kR240 = 240, kC240 = 256, kWm240 = 3, kWn240 = 4, kThreads240 = 64 * kWm240 * kWn240;
constexpr int kTm240 = kR240 / kWm240 / 16, kTn240 = kC240 / kWn240 / 16;   // 5 x 4 tiles per wave
constexpr int kStage240 = 16 * (kC240 / kWn240 + 4);                       // per-wave epilogue patch [16][68]

// ---- 240-row panels, operands straight into MFMA fragments (no LDS staging, no barriers) --------
// 240 x 256 outputs per workgroup on the 16x16x4 MFMA (a form that staged the panels through LDS is in tools/diag); a
// lane fetches its own fragments from global memory: lane (i, q) of a wave needs
// A[row i][16 g + 4 q .. + 3] — one 16-byte buffer load — and the 16 lanes of a quarter cover 16
// rows x 64 contiguous bytes.  No LDS images, no stash, no barrier: the waves of a workgroup are
// independent instruction streams, one wave's wait for memory is another wave's MFMA time, and
// the store burst of the epilogue spreads out the same way.  DEPTH + 1 register sets of fragments:
// the loads of reduction groups g+1 .. g+DEPTH are in flight under the MFMAs of group g.
//   WM x WN = 3 x 4: twelve waves of 80 x 64 outputs (3 per SIMD, 80 accumulator registers); every
//             A row is fetched by four waves and every weight row by three (L1 / L2 hits, but
//             42 B/clk of L1 traffic per CU);
//   WM x WN = 1 x 4: four waves of 240 x 64 outputs, ONE per SIMD with 240 accumulator registers and
//             the whole 512-register file: every A element is fetched exactly once per workgroup,
//             10 B/clk of L1 traffic, and a group of 240 MFMAs (3.2 us) covers the next loads.
typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr unsigned kOutOfRange = 0x7FFFFFF0u;   // byte offset no operand panel reaches: the load returns 0

// TM + TN 16-byte buffer loads: address = panel base (SGPR resource) + lane offset (one VGPR per
// fragment row block) + reduction offset (SGPR); offsets past the panel's bytes read as 0
// (`group` = reduction group of 16; the activation panel advances 64 bytes per group, the weights `b_step` bytes: 64 as
// stored by torch, 1024 in fragment order)
template <int TM, int TN, bool B_FIRST = false, int B_STEP = 64>
__device__ __forceinline__ void load_fragments(v4f (&af)[TM], v4f (&bf)[TN], __amdgpu_buffer_rsrc_t ra,
                                               __amdgpu_buffer_rsrc_t rb, const unsigned (&off_a)[TM],
                                               const unsigned (&off_b)[TN], int group) {
  const int k_bytes = 64 * group, kb_bytes = B_STEP * group;
  if constexpr (B_FIRST) {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
      bf[tn] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rb, off_b[tn], kb_bytes, 0));
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
      af[tm] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(ra, off_a[tm], k_bytes, 0));
    return;
  }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
    af[tm] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(ra, off_a[tm], k_bytes, 0));
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
    bf[tn] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rb, off_b[tn], kb_bytes, 0));
}

template <int TM, int TN>
__device__ __forceinline__ void mfma_group(v4acc (&acc)[TM][TN], const v4f (&af)[TM], const v4f (&bf)[TN]) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[tm][j], bf[tn][j], acc[tm][tn], 0, 0, 0);
}

struct PanelStage {   // one GEMM of the panel kernel: c[rows of the panel, rb] = act(a0 b0^T + a1 b1^T + bias) (. mask)
  const float* a[2];
  const float* b[2];
  const float* bp[2];   // b[seg] in fragment order, or null (GemmArgs::bp)
  int lda[2], ldb[2], kseg[2];
  int ra, rb, ldc, relu;
  float* c;
  const float* bias;
  const float* mask;
  const float* sc_l;   // optional score vectors / partial-score outputs (see GemmArgs)
  const float* sc_r;
  float* sc_el;
  float* sc_er;
  unsigned long long* bits_out;        // optional: sign bits of c (see GemmArgs)
  const unsigned long long* bits_in;   // optional: `mask` as bits
  int mask_kind;                       // 0: ReLU mask, 1: ELU derivative through `mask` (see GemmArgs)
  float* col_partial;                  // optional: column sums per wave row block
};

// What the epilogue of a stage does, as template bits: with kEpiRuntime every switch is read from the arguments
// (any shape); without it the switches are compile-time facts and the output is whole 256-column blocks
// (host-checked) — the epilogue of the layer-stack launches loses its ~50 uniform branches per 16 rows and
// most of its code (the generic kernel is ~100 KB of instructions, more than the instruction cache).
enum : int { kEpiBias = 1, kEpiRelu = 2, kEpiMaskBits = 4, kEpiScores = 8, kEpiBitsOut = 16, kEpiEluSums = 32, kEpiRuntime = 256, kEpiAbsent = -1 };

template <int WM, int WN, int DEPTH, int F = kEpiRuntime, int ROWS = kR240, int ILV = 0, bool PK = false>
__device__ __forceinline__ void panel_stage(const PanelStage s, float* lds, int sched, int m0, int n0, int row_end) {
  constexpr bool G = (F & kEpiRuntime) != 0;
  constexpr int WTM = ROWS / WM, WTN = kC240 / WN, TM = WTM / 16, TN = WTN / 16;
  constexpr int R = DEPTH + 1;                     // register sets of fragments
  constexpr int kLd = WTN + 4, kStage = 16 * kLd;  // per-wave epilogue patch [16][WTN + 4]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave index in an SGPR
  const int wm = wave / WN, wn = wave % WN;
  const int i16 = lane & 15, q = lane >> 4;

  v4acc acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = v4acc{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int seg = 0; seg < 2; ++seg) {
    const int kseg = s.kseg[seg];
    if (kseg == 0) continue;
    const int lda = s.lda[seg], ldb = s.ldb[seg];
    // buffer resources over this workgroup's operand panels: rows [m0, row_end) of A, weight rows
    // [n0, n0 + 256) — anything past their last byte reads as 0 (no row clamps, no branches)
    const int cols = min(s.rb - n0, kC240);
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(s.a[seg] + static_cast<size_t>(m0) * lda), 0, (row_end - m0) * lda * 4, 0x00020000);
    // PK: the weights come in fragment order (s.bp): [16-row tile of B][reduction group of 16][lane][4 floats], zero-padded
    // to whole tiles and groups — a fragment is 1 KiB of consecutive bytes, lane l takes bytes 16 l .. 16 l + 15.  A
    // compile-time fact of the instantiation (a run-time choice between the two address forms cost 240 spilled registers)
    constexpr int b_step = PK ? 1024 : 64;
    const int groups = (kseg + 15) >> 4;
    __amdgpu_buffer_rsrc_t rb;
    unsigned off_a[TM], off_b[TN];
    if constexpr (PK) {
      rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s.bp[seg]), 0, ((s.rb + 15) >> 4) * groups * 1024, 0x00020000);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) off_b[tn] = (static_cast<unsigned>((n0 + wn * WTN) / 16 + tn) * groups * 64 + lane) * 16;
    } else {
      rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s.b[seg] + static_cast<size_t>(n0) * ldb), 0, cols * ldb * 4,
                                             0x00020000);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) off_b[tn] = (static_cast<unsigned>(wn * WTN + tn * 16 + i16) * ldb + 4 * q) * 4;
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) off_a[tm] = (static_cast<unsigned>(wm * WTM + tm * 16 + i16) * lda + 4 * q) * 4;
    const int n_full = kseg / 16, tail = kseg % 16;
    v4f af[R][TM], bf[R][TN];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u)     // groups 0 .. DEPTH-1 (clamped: re-reads are harmless)
      if (n_full > 0) load_fragments<TM, TN, false, b_step>(af[u], bf[u], ra, rb, off_a, off_b, min(u, n_full - 1));
    int g = 0;
    for (; g + R <= n_full; g += R) {
#pragma unroll
      for (int u = 0; u < R; ++u) {
        load_fragments<TM, TN, ILV >= 0, b_step>(af[(u + DEPTH) % R], bf[(u + DEPTH) % R], ra, rb, off_a, off_b,
                                                 min(g + u + DEPTH, n_full - 1));
        mfma_group(acc, af[u], bf[u]);
      }
      // pin the software pipeline: the loads of a group are issued before the MFMAs of the group
      // DEPTH in front of it (left alone, the scheduler sinks them to save registers and the wave
      // then waits for each load right after issuing it)
#pragma unroll
      for (int u = 0; u < R; ++u) {
        if constexpr (ILV < 0) {   // the loads of a group in one burst in front of its MFMAs (rounds 1 - 2; kept for A/B runs)
          __builtin_amdgcn_sched_group_barrier(0x020, TM + TN, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM * TN, 0);
        } else {   // one load, then kPer MFMAs, ...: a burst of nine loads holds up the wave's own MFMA issue (round 3)
          constexpr int kPer = ILV > 0 ? ILV : 4 * TM * TN / (TM + TN);
          constexpr int kRest = 4 * TM * TN - kPer * (TM + TN);
          static_assert(kPer >= 1 && kRest >= 0, "MFMAs per load do not fit the group");
#pragma unroll
          for (int l = 0; l < TM + TN; ++l) {
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, kPer, 0);
          }
          if (kRest > 0) __builtin_amdgcn_sched_group_barrier(0x008, kRest, 0);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < R - 1; ++u) {   // the last n_full % R groups: their fragments are already on the way
      if (g + u < n_full) {
        if (g + u + DEPTH < n_full)
          load_fragments<TM, TN, false, b_step>(af[(u + DEPTH) % R], bf[(u + DEPTH) % R], ra, rb, off_a, off_b, g + u + DEPTH);
        mfma_group(acc, af[u], bf[u]);
      }
    }
    if (tail != 0) {   // kseg is a multiple of 4: quarter q lies inside the tail or past the row's end
      if (4 * q >= tail) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) off_a[tm] = kOutOfRange;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) off_b[tn] = kOutOfRange;
      }
      load_fragments<TM, TN, false, b_step>(af[0], bf[0], ra, rb, off_a, off_b, n_full);
      mfma_group(acc, af[0], bf[0]);
    }
  }

  // Epilogue: a row of TN tiles (16 x WTN outputs) through the wave's LDS
  // patch, out as 16-byte row segments with bias / ReLU / mask applied as float4.
  float* stage = lds + wave * kStage;
  const bool wide = G ? (s.rb & 3) == 0 && (s.ldc & 3) == 0 : true;
  constexpr int kC4 = WTN / 4;                // float4 per patch row
  constexpr int kRowsPerIt = 64 / kC4;        // patch rows one pass of the wave covers
  const int c4 = (lane % kC4) * 4, rsub = lane / kC4;
  const int col = n0 + wn * WTN + c4;
  const bool col_ok = G ? col < s.rb : true;
  const bool has_bias = G ? s.bias != nullptr : (F & kEpiBias) != 0;
  const bool relu = G ? s.relu != 0 : (F & kEpiRelu) != 0;
  const bool scores = G ? s.sc_l != nullptr : (F & kEpiScores) != 0;
  v4f bias = {0.f, 0.f, 0.f, 0.f};
  if (wide && has_bias && col_ok) bias = *reinterpret_cast<const v4f*>(s.bias + col);
  v4f sc_wl = {0.f, 0.f, 0.f, 0.f}, sc_wr = sc_wl;
  if (wide && scores && col_ok) {
    sc_wl = *reinterpret_cast<const v4f*>(s.sc_l + col);
    sc_wr = *reinterpret_cast<const v4f*>(s.sc_r + col);
  }
  // ReLU masks as bits (GemmArgs::bits_out / bits_in): this wave's TM * 4 row groups of its 64-column block are
  // TM * 16 consecutive words of the [column block][row group][4] layout
  constexpr int kBitWords = TM * 16;
  static_assert(G || WTN == 64, "the compile-time epilogues keep mask bits: 64-column wave tiles");
  const bool bits_here = G ? WTN == 64 && wide && n0 + wn * WTN < s.rb : true;
  const bool bit_mask = G ? bits_here && s.bits_in != nullptr && s.mask != nullptr : (F & kEpiMaskBits) != 0;
  const bool bits_wanted = G ? bits_here && s.bits_out != nullptr : (F & kEpiBitsOut) != 0;
  const size_t bits_at = (static_cast<size_t>((n0 + wn * WTN) >> 6) * ((s.ra + 3) >> 2) + ((m0 + wm * WTM) >> 2)) * 4;
  unsigned long long* bit_words = reinterpret_cast<unsigned long long*>(lds + WM * WN * kStage) + wave * kBitWords;
  const bool elu_mask = G ? s.mask_kind == 1 : (F & kEpiEluSums) != 0;
  const bool col_sums = G ? s.col_partial != nullptr : (F & kEpiEluSums) != 0;
  v4f csum = {0.f, 0.f, 0.f, 0.f};   // this lane's four columns over the rows it stores (rsub, rsub + 4, ...: fixed order)
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int row_base = m0 + wm * WTM + tm * 16;
    if (wide) {
      v4f mk[16 / kRowsPerIt];
      const bool float_mask = G ? s.mask != nullptr && !bit_mask : (F & kEpiEluSums) != 0;
      if (float_mask) {
#pragma unroll
        for (int it = 0; it < 16 / kRowsPerIt; ++it) {
          const int row = row_base + it * kRowsPerIt + rsub;
          mk[it] = (row < row_end && col_ok)
                       ? *reinterpret_cast<const v4f*>(s.mask + static_cast<size_t>(row) * s.ldc + col)
                       : v4f{0.f, 0.f, 0.f, 0.f};
        }
      }
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) stage[(4 * q + r) * kLd + tn * 16 + i16] = acc[tm][tn][r];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 16 / kRowsPerIt; ++it) {
        const int lrow = it * kRowsPerIt + rsub, row = row_base + lrow;
        v4f val = *reinterpret_cast<const v4f*>(stage + lrow * kLd + c4) + bias;
        // four rows x 64 columns per pass: their four mask words (bit = lane) sit at one wave-uniform
        // address — scalar loads, which do not queue behind this wave's stores as vector loads do
        if (bit_mask && row_base + it * kRowsPerIt < row_end) {
          const unsigned long long* words = s.bits_in + bits_at + (tm * (16 / kRowsPerIt) + it) * 4;
#pragma unroll
          for (int e = 0; e < 4; ++e) val[e] = (words[e] >> lane) & 1ull ? val[e] : 0.f;
        }
        if (row < row_end && col_ok) {
          if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) val[e] = fmaxf(val[e], 0.f);
          }
          if (float_mask) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              val[e] = mk[it][e] > 0.f ? val[e] : elu_mask ? val[e] * (mk[it][e] + 1.0f) : 0.f;
          }
          if (col_sums) csum += val;
          v4f* dst = reinterpret_cast<v4f*>(s.c + static_cast<size_t>(row) * s.ldc + col);
          if (G && (sched & 2)) __builtin_nontemporal_store(val, dst);
          else *dst = val;
        }
        if (bits_wanted) {   // one wave-wide comparison per element slot = one word; collected in LDS, stored once
          const unsigned long long w0 = __ballot(val[0] > 0.f), w1 = __ballot(val[1] > 0.f);
          const unsigned long long w2 = __ballot(val[2] > 0.f), w3 = __ballot(val[3] > 0.f);
          if (lane < 4) bit_words[(tm * (16 / kRowsPerIt) + it) * 4 + lane] = lane == 0 ? w0 : lane == 1 ? w1 : lane == 2 ? w2 : w3;
        }
        if (scores) {   // rb is a multiple of WTN here: every lane's columns are real
          float pl = (val[0] * sc_wl[0] + val[1] * sc_wl[1]) + (val[2] * sc_wl[2] + val[3] * sc_wl[3]);
          float pr = (val[0] * sc_wr[0] + val[1] * sc_wr[1]) + (val[2] * sc_wr[2] + val[3] * sc_wr[3]);
#pragma unroll
          for (int o = 1; o < kC4; o <<= 1) pl += __shfl_xor(pl, o, kWave), pr += __shfl_xor(pr, o, kWave);
          if (lane % kC4 == 0 && row < row_end) {
            const size_t at = static_cast<size_t>(row) * (s.rb / WTN) + (n0 / WTN + wn);
            s.sc_el[at] = pl, s.sc_er[at] = pr;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    } else {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int c = n0 + wn * WTN + tn * 16 + i16;
        const float bs = (s.bias != nullptr && c < s.rb) ? s.bias[c] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = row_base + 4 * q + r;
          if (row < row_end && c < s.rb) {
            const size_t off = static_cast<size_t>(row) * s.ldc + c;
            float val = acc[tm][tn][r] + bs;
            if (s.relu) val = fmaxf(val, 0.f);
            if (s.mask != nullptr) val = s.mask[off] > 0.f ? val : 0.f;
            s.c[off] = val;
          }
        }
      }
    }
  }
  static_assert(kC4 == 16 || !(F & kEpiEluSums), "column sums: 64-column wave tiles");
  if (col_sums && wide && kC4 == 16) {   // the four row residues of a column group sit 16 lanes apart: two fixed-order exchanges
#pragma unroll
    for (int e = 0; e < 4; ++e) csum[e] += __shfl_xor(csum[e], 16, kWave);
#pragma unroll
    for (int e = 0; e < 4; ++e) csum[e] += __shfl_xor(csum[e], 32, kWave);
    if (lane < kC4 && col_ok)
      *reinterpret_cast<v4f*>(s.col_partial + static_cast<size_t>((m0 / ROWS) * WM + wm) * s.rb + col) = csum;
  }
  if (bits_wanted) {
    __builtin_amdgcn_wave_barrier();
    const int valid = min(kBitWords, ((row_end - (m0 + wm * WTM) + 3) >> 2) * 4);   // words of rows that exist
#pragma unroll
    for (int base = 0; base < kBitWords; base += 64)
      if (base + lane < valid) s.bits_out[bits_at + base + lane] = bit_words[base + lane];
    __builtin_amdgcn_wave_barrier();
  }
}

template <int WM, int WN, int DEPTH, class Probe = NoProbe, int F1 = kEpiRuntime, int F2 = kEpiRuntime, int ROWS = kR240, int ILV = 0,
          bool PK = false>
__global__ __launch_bounds__(64 * WM * WN, WM * WN / 4) void gemm_panel_direct_kernel(const GemmArgs p) {
  constexpr int WTN = kC240 / WN;
  static_assert((ROWS / WM) % 16 == 0 && WTN % 16 == 0 && WTN % 4 == 0 && (WM * WN) % 4 == 0, "wave tiles are whole 16x16 tiles");
  // epilogue patches [16][WTN + 4] per wave, then the waves' mask words (ROWS / WM / 16 * 16 of 8 bytes each)
  __shared__ __attribute__((aligned(16))) float lds[WM * WN * 16 * (WTN + 4) + WM * WN * (ROWS / WM) * 2];
  const int m0 = blockIdx.x * ROWS, n0 = blockIdx.y * kC240;
  const int row_end = min(p.ra, m0 + ROWS);
  Probe::mark(0);
  Probe::mark(1);
  PanelStage s0{};
  s0.a[0] = p.a[0], s0.a[1] = p.a[1], s0.b[0] = p.b[0], s0.b[1] = p.b[1], s0.bp[0] = p.bp[0], s0.bp[1] = p.bp[1];
  s0.lda[0] = p.lda[0], s0.lda[1] = p.lda[1], s0.ldb[0] = p.ldb[0], s0.ldb[1] = p.ldb[1];
  s0.kseg[0] = p.kseg[0], s0.kseg[1] = p.kseg[1];
  s0.ra = p.ra, s0.rb = p.rb, s0.ldc = p.ldc, s0.relu = p.relu, s0.c = p.c, s0.bias = p.bias, s0.mask = p.mask;
  s0.sc_l = p.sc_l, s0.sc_r = p.sc_r, s0.sc_el = p.sc_el, s0.sc_er = p.sc_er;
  s0.bits_out = p.bits_out, s0.bits_in = p.bits_in, s0.mask_kind = p.mask_kind, s0.col_partial = p.col_partial;
  panel_stage<WM, WN, DEPTH, F1, ROWS, ILV, PK>(s0, lds, p.sched, m0, n0, row_end);
  Probe::mark(2);
  // With c2 set a second GEMM follows in the same launch: the rows this workgroup has just stored are
  // its A operand (the next layer's fc_pool behind fc_self + fc_neigh; the next input gradient behind
  // this one) — one launch, one cold start and one output burst less per layer, and the operand
  // comes back out of this CU's own L2 slice.
  if (F2 != kEpiAbsent && ((F2 & kEpiRuntime) == 0 || p.c2 != nullptr)) {
    // Stage 2 re-reads rows of c that OTHER waves of this workgroup stored.  Ordering: every wave's vmcnt(0), then the
    // workgroup barrier.  That is enough because the workgroup runs on one CU whose vector L1 all its waves share
    // (the default, non-tgsplit execution mode) and the stores are ordinary ones (the launcher never combines
    // non-temporal stores, sched bit 2, with a second stage).
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's rows of c are in memory ...
    __threadfence_block();
    __syncthreads();                       // ... and so are every other wave's, before any is read back
    PanelStage s1{};
    s1.a[0] = p.c, s1.a[1] = p.c, s1.b[0] = p.b2, s1.b[1] = p.b2, s1.bp[0] = p.bp2, s1.bp[1] = nullptr;
    s1.lda[0] = s1.lda[1] = p.ldc, s1.ldb[0] = s1.ldb[1] = p.ldb2;
    s1.kseg[0] = p.rb, s1.kseg[1] = 0;
    s1.ra = p.ra, s1.rb = p.rb2, s1.ldc = p.ldc2, s1.relu = p.relu2, s1.c = p.c2, s1.bias = p.bias2, s1.mask = nullptr;
    panel_stage<WM, WN, DEPTH, F2 == kEpiAbsent ? kEpiRuntime : F2, ROWS, ILV, PK>(s1, lds, p.sched, m0, 0, row_end);
  }
  Probe::mark(3);
}

// the instantiation that reads its weights in fragment order when the call brought a copy of EVERY weight operand
// (GemmArgs::bp / bp2), else the one that reads them as stored
#define GTS_PANEL_LAUNCH(F1_, F2_, ILV_)                                                                   \
  do {                                                                                                    \
    if (all_packed) gemm_panel_direct_kernel<3, 4, 1, NoProbe, F1_, F2_, ROWS, ILV_, true><<<grid, 768, 0, st>>>(q);   \
    else gemm_panel_direct_kernel<3, 4, 1, NoProbe, F1_, F2_, ROWS, ILV_, false><<<grid, 768, 0, st>>>(q);             \
    return launch_status();                                                                               \
  } while (0)

template <int WM, int WN, int DEPTH, class Probe = NoProbe, int ROWS = kR240>
int launch_panel_direct(const GemmArgs& p, hipStream_t st) {
  dim3 grid((p.ra + ROWS - 1) / ROWS, (p.rb + kC240 - 1) / kC240, 1);
  GemmArgs q = p;
  q.sched = g_gemm_sched;
  if (p.c2 != nullptr) q.sched &= ~2;   // chained launches read their own output back through L1 / L2: ordinary stores only
  const bool all_packed = p.bp[0] != nullptr && (p.kseg[1] == 0 || p.bp[1] != nullptr) && (p.c2 == nullptr || p.bp2 != nullptr) &&
                          !(q.sched & 16);   // GTS_OPT_GEMM_SCHED bit 16: ignore the copies (A/B runs)
  if constexpr (WM == 3 && WN == 4 && DEPTH == 1 && std::is_same<Probe, NoProbe>::value) {
    // the launches of the SAGE-pool layer stack at its 256-wide layers: compile-time epilogues
    const bool whole_cols = p.rb % kC240 == 0 && p.ldc % 4 == 0 && (p.mask == nullptr || p.bits_in != nullptr) &&
                            (p.c2 == nullptr || (p.rb2 % kC240 == 0 && p.ldc2 % 4 == 0)) && !(q.sched & 2) && !(q.sched & 4);
    const bool whole = whole_cols && p.sc_l == nullptr && p.mask_kind == 0 && p.col_partial == nullptr;
    if (p.rb % kC240 == 0 && p.ldc % 4 == 0 && p.mask_kind == 1 && p.mask != nullptr && p.col_partial != nullptr &&
        p.bits_in == nullptr && p.bits_out == nullptr && p.sc_l == nullptr && p.c2 == nullptr && p.bias == nullptr &&
        !p.relu && !(q.sched & 6)) {   // an input gradient through the ELU of the layer below, with that layer's bias gradient
      GTS_PANEL_LAUNCH(kEpiEluSums, kEpiAbsent, 0);
    }
    const int f1 = (p.bias ? kEpiBias : 0) | (p.relu ? kEpiRelu : 0) | (p.mask ? kEpiMaskBits : 0) | (p.bits_out ? kEpiBitsOut : 0);
    const int f2 = p.c2 == nullptr ? kEpiAbsent : (p.bias2 ? kEpiBias : 0) | (p.relu2 ? kEpiRelu : 0);
    constexpr int kFwd = kEpiBias | kEpiRelu;
    if (whole && f1 == (kFwd | kEpiBitsOut) && f2 == kFwd) {          // fc_self + fc_neigh, then the next fc_pool (training)
      if constexpr (ROWS == kR240) {
        if (q.sched & 8) {   // A/B: the grouped load order of rounds 1 - 2
          gemm_panel_direct_kernel<3, 4, 1, NoProbe, kFwd | kEpiBitsOut, kFwd, ROWS, -1><<<grid, 768, 0, st>>>(q);
          return launch_status();
        }
      }
      GTS_PANEL_LAUNCH(kFwd | kEpiBitsOut, kFwd, 0);
    }
    if (whole && f1 == kEpiMaskBits && f2 == 0) {                     // a layer's input gradient, then g @ W_neigh below
      if constexpr (ROWS == kR240) {
        if (q.sched & 8) {   // A/B: the grouped load order of rounds 1 - 2
          gemm_panel_direct_kernel<3, 4, 1, NoProbe, kEpiMaskBits, 0, ROWS, -1><<<grid, 768, 0, st>>>(q);
          return launch_status();
        }
      }
      GTS_PANEL_LAUNCH(kEpiMaskBits, 0, 0);
    }
    if (whole && f1 == kFwd && f2 == kFwd) {                          // the same pair without mask bits (no-grad forward: inference, evaluate)
      GTS_PANEL_LAUNCH(kFwd, kFwd, 0);
    }
    if (whole && f1 == kFwd && f2 == kEpiAbsent) {                    // one biased ReLU layer on its own (fc_pool of the first wide layer)
      GTS_PANEL_LAUNCH(kFwd, kEpiAbsent, 0);
    }
    if (whole_cols && p.sc_l != nullptr && f1 == 0 && f2 == kEpiAbsent) {   // GATConv's fc with the attention scores in its epilogue
      GTS_PANEL_LAUNCH(kEpiScores, kEpiAbsent, 0);
    }
    if (whole && f1 == 0 && f2 == kEpiAbsent) {                       // a plain product (g @ W_neigh of the top layer)
      GTS_PANEL_LAUNCH(0, kEpiAbsent, 0);
    }
    GTS_PANEL_LAUNCH(kEpiRuntime, kEpiRuntime, 0);
  }
  gemm_panel_direct_kernel<WM, WN, DEPTH, Probe, kEpiRuntime, kEpiRuntime, ROWS><<<grid, 64 * WM * WN, 0, st>>>(q);
  return launch_status();
}
#undef GTS_PANEL_LAUNCH

// ---- weight gradients, a main loop of MFMAs, LDS reads and nothing else (round 3) -----------------------------------
// The f32 MFMA and the vector ALU share their arithmetic on gfx950: a vector instruction between two MFMAs costs the
// wave ~17 cycles of matrix-pipe time, each further one ~4 (profiles/r03_mfma_valu_coissue.log).  The two kernels above
// carry 20 - 30 of them per reduction tile (fragment addresses, DMA offsets, the bias column sums in every wave): that,
// not the issue of the LDS reads, is what held them at 87.9 % of the matrix pipe.  Same 256 x 256 tile, slabs, MFMA
// order and column-sum order as gemm_kernel<256, 256, 4, 4, false, false, true> (bit-identical results), same
// LDS-DMA tile copies (as the rejected wgrad_dma_kernel, tools/diag/gemm_rejected_forms.inc), but:
//   * LDS holds [A image 0 | A image 1 | B image 0 | B image 1] (32 KiB each), so ONE address register per 32-row
//     block reaches every fragment element of BOTH images through the immediate offsets of ds_read2st64_b32 (units of
//     256 B: reduction row r of image I sits 4 r + 128 I units up; a read fetches steps j, j + 1);
//   * the reduction tiles are walked two at a time, the image a compile-time fact;
//   * a tile's DMA descriptor is built by the scalar unit (base and length of the tile's 32 rows: rows past the split
//     read as zeros), the lane offsets never change;
//   * the bias column sums run in one wave per SIMD (wn == wm) and only for the problems that have a bias, in a
//     copy of the loop of their own, so the other waves' loop has no vector instruction at all.
// Fragment reads are issued one pair of steps ahead by inline assembly (the compiler would pair tm = 0 / 1 into
// ds_read2_b32 and add up a new address per step); each wait carries the fragment registers as operands so the MFMAs
// that consume them cannot be scheduled above it.
typedef float v2f __attribute__((ext_vector_type(2)));

template <int O0, int O1>
__device__ __forceinline__ v2f lds_read2st64(unsigned addr) {
  static_assert(O0 >= 0 && O1 <= 255, "ds_read2st64_b32 offsets are 8 bits");
  v2f r;
  asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(r) : "v"(addr), "n"(O0), "n"(O1));
  return r;
}

template <class Probe = NoProbe>
__global__ __launch_bounds__(1024, 4) void wgrad_stream_kernel(const GemmArgs p) {
  constexpr int BM = 256, BN = 256, WM = 4, WN = 4, WTM = 64, WTN = 64, TM = 2, TN = 2;
  constexpr int kPlane = kBK * BM;   // floats of one operand image (32 KiB)
  static_assert(BM == BN && kPlane * 4 == 128 * 256, "image I of an operand sits 128 offset units above image 0");
  __shared__ float lds[4 * kPlane];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int i = lane & 31, h = lane >> 5;
  const int problem = blockIdx.y / p.tiles_n, tile_n = blockIdx.y % p.tiles_n;
  const int m0 = blockIdx.x * BM, n0 = tile_n * BN;   // output rows (columns of g) / output columns (columns of act)
  const float* g = kernarg_entry<const float*>(offsetof(GemmArgs, pa), problem);
  const float* act = kernarg_entry<const float*>(offsetof(GemmArgs, pb), problem);
  const int ldg = p.lda[0], lda = p.ldb[0];
  const int n_tiles_all = (p.kseg[0] + kBK - 1) / kBK;
  const int t_beg = min(n_tiles_all, static_cast<int>(blockIdx.z) * p.tiles_per_split);
  const int t_end = min(n_tiles_all, t_beg + p.tiles_per_split);
  const int row_beg = t_beg * kBK, rows = min(p.kseg[0], t_end * kBK) - row_beg;   // reduction rows of this split
  const int n_tiles = t_end - t_beg;
  // wave w copies rows w and w + 16 of both tiles; a lane whose four columns lie past the operand is parked outside
  // the descriptor (zeros land in LDS)
  const bool g_ok = m0 + 4 * lane < p.ra, a_ok = n0 + 4 * lane < p.rb;
  const unsigned vg0 = g_ok ? static_cast<unsigned>(wave * ldg + m0 + 4 * lane) * 4 : kOutOfRange;
  const unsigned vg1 = g_ok ? static_cast<unsigned>((wave + 16) * ldg + m0 + 4 * lane) * 4 : kOutOfRange;
  const unsigned va0 = a_ok ? static_cast<unsigned>(wave * lda + n0 + 4 * lane) * 4 : kOutOfRange;
  const unsigned va1 = a_ok ? static_cast<unsigned>((wave + 16) * lda + n0 + 4 * lane) * 4 : kOutOfRange;
  auto dma = [&](int t, int image) __attribute__((always_inline)) {   // tile t (counted from t_beg) -> image
    const int r0 = row_beg + t * kBK, nr = min(kBK, rows - t * kBK);
    __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g + static_cast<size_t>(r0) * ldg), 0,
                                                                  nr * ldg * 4, 0x00020000);
    __amdgpu_buffer_rsrc_t ract = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(act + static_cast<size_t>(r0) * lda), 0,
                                                                    nr * lda * 4, 0x00020000);
    float* ia = lds + image * kPlane + wave * BM;
    float* ib = lds + (2 + image) * kPlane + wave * BN;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, ia, 16, vg0, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, ia + 16 * BM, 16, vg1, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ract, ib, 16, va0, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ract, ib + 16 * BN, 16, va1, 0, 0, 0);
  };

  v16f acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
  float csum[TM] = {0.f, 0.f};
  const bool want_colsum = p.colsum != nullptr && tile_n == 0 && wn == wm && ((p.colsum_mask >> problem) & 1u) != 0;

  // fragment addresses (LDS bytes) of image 0: element (reduction row 4 h, column of this lane's 32-row block)
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) float*)lds));
  const unsigned a_addr0 = lds0 + static_cast<unsigned>(4 * h * BM + wm * WTM + i) * 4, a_addr1 = a_addr0 + 128;
  const unsigned b_addr0 = lds0 + static_cast<unsigned>(2 * kPlane + 4 * h * BN + wn * WTN + i) * 4, b_addr1 = b_addr0 + 128;
  v2f fa[2][TM], fb[2][TN];   // two pairs of steps in flight: pair q in slot q & 1
  fa[0][0] = fa[0][1] = fa[1][0] = fa[1][1] = v2f{0.f, 0.f};
  fb[0][0] = fb[0][1] = fb[1][0] = fb[1][1] = v2f{0.f, 0.f};

  // the four reads of pair Q (steps 2 Q, 2 Q + 1: reduction rows 8 (Q / 2) + 2 (Q % 2) + {0, 1} (+ 4 h)) of image IMG
  auto issue = [&](auto img_c, auto pair_c) __attribute__((always_inline)) {
    constexpr int IMG = decltype(img_c)::value, Q = decltype(pair_c)::value;
    constexpr int O = 4 * (8 * (Q / 2) + 2 * (Q % 2)) + 128 * IMG;
    fa[Q & 1][0] = lds_read2st64<O, O + 4>(a_addr0);
    fa[Q & 1][1] = lds_read2st64<O, O + 4>(a_addr1);
    fb[Q & 1][0] = lds_read2st64<O, O + 4>(b_addr0);
    fb[Q & 1][1] = lds_read2st64<O, O + 4>(b_addr1);
  };
  auto multiply = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[slot][tm][e], fb[slot][tn][e], acc[tm][tn], 0, 0, 0);
  };
  float half_sum[TM] = {0.f, 0.f};
  // one reduction tile out of image IMG; on entry the reads of its pair 0 are in flight (slot 0)
  auto tile = [&](auto img_c, auto cs_c, int t) __attribute__((always_inline)) {
    constexpr int IMG = decltype(img_c)::value;
    constexpr bool CS = decltype(cs_c)::value;
    auto pair = [&](auto pair_c) __attribute__((always_inline)) {
      constexpr int Q = decltype(pair_c)::value, S = Q & 1;
      if constexpr (Q < 7) {
        issue(img_c, std::integral_constant<int, Q + 1>{});
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fa[S][0]), "+v"(fa[S][1]), "+v"(fb[S][0]), "+v"(fb[S][1]));
      } else {
        // every read of this image has been issued: wait for them and for this wave's pieces of the next tile, meet
        // the other waves, hand the image to the DMA of the tile after next and start on the next image
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier"
                     : "+v"(fa[S][0]), "+v"(fa[S][1]), "+v"(fb[S][0]), "+v"(fb[S][1]) : : "memory");
        if (t + 2 < n_tiles) dma(t + 2, IMG);
        if (t + 1 < n_tiles) issue(std::integral_constant<int, IMG ^ 1>{}, std::integral_constant<int, 0>{});
      }
      if constexpr (Q % 2 == 0) {   // waves further into a tile yield MFMA issue (the builtin wants a literal)
        if constexpr (Q == 0) __builtin_amdgcn_s_setprio(3);
        else if constexpr (Q == 2) __builtin_amdgcn_s_setprio(2);
        else if constexpr (Q == 4) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
      }
      if constexpr (CS) {   // csum += (a0 + a1) + (a2 + a3) per group of four steps, as gemm_kernel sums them; the adds are
                            // pinned here (left to the compiler they drift away from the fragments, which it then spills)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          if constexpr (Q % 2 == 0) {
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(half_sum[tm]) : "v"(fa[S][tm][0]), "v"(fa[S][tm][1]));
          } else {
            float upper;
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(upper) : "v"(fa[S][tm][0]), "v"(fa[S][tm][1]));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(upper) : "v"(half_sum[tm]), "v"(upper));
            asm volatile("v_add_f32 %0, %1, %2" : "+v"(csum[tm]) : "v"(csum[tm]), "v"(upper));
          }
        }
      }
      multiply(S);
    };
    pair(std::integral_constant<int, 0>{}), pair(std::integral_constant<int, 1>{}), pair(std::integral_constant<int, 2>{}),
        pair(std::integral_constant<int, 3>{}), pair(std::integral_constant<int, 4>{}), pair(std::integral_constant<int, 5>{}),
        pair(std::integral_constant<int, 6>{}), pair(std::integral_constant<int, 7>{});
  };
  auto reduce = [&](auto cs_c) __attribute__((always_inline)) {
    int t = 0;
    for (; t + 1 < n_tiles; t += 2) {
      tile(std::integral_constant<int, 0>{}, cs_c, t);
      tile(std::integral_constant<int, 1>{}, cs_c, t + 1);
    }
    if (t < n_tiles) tile(std::integral_constant<int, 0>{}, cs_c, t);
  };

  Probe::mark(0);
  if (n_tiles > 0) {
    dma(0, 0);
    if (n_tiles > 1) dma(1, 1);
    // every piece has landed, everybody's.  vmcnt(0), not (4): should the compiler ever spill around here, its scratch
    // stores would count in vmcnt too and need not retire in order with the loads
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    issue(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  }
  Probe::mark(1);
  if (want_colsum) reduce(std::true_type{});
  else reduce(std::false_type{});
  __builtin_amdgcn_s_setprio(0);
  Probe::mark(2);
  const size_t slab = static_cast<size_t>(problem) * p.n_splits + blockIdx.z;
  write_tile<BM, BN, WM, WN>(p, lds, p.c + slab * p.ra * p.ldc, acc, m0, n0);
  Probe::mark(3);
  if (want_colsum) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const float total = csum[tm] + __shfl_xor(csum[tm], 32, kWave);   // the two kk halves
      const int row = m0 + wm * WTM + tm * 32 + i;
      if (h == 0 && row < p.ra) p.colsum[slab * p.ra + row] = total;
    }
  }
}

struct ReduceArgs {   // weight-slab + bias-slab jobs of every problem in one launch (blockIdx.y = job)
  const float* slabs[2 * kMaxProblems];
  float* out[2 * kMaxProblems];
  int n4[2 * kMaxProblems];  // float4 per job
  int splits;
};

__global__ __launch_bounds__(kBlock) void reduce_slabs_kernel(const ReduceArgs p) {
  __shared__ v4f part[4][64];
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int q = blockIdx.y;
  const float* slabs = kernarg_entry<const float*>(offsetof(ReduceArgs, slabs), q);
  float* out = kernarg_entry<float*>(offsetof(ReduceArgs, out), q);
  const int n4 = kernarg_entry<int>(offsetof(ReduceArgs, n4), q);
  const int i = blockIdx.x * 64 + col;
  if (blockIdx.x * 64 >= n4) return;   // whole workgroup past this job's end
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  if (i < n4) {
    const v4f* src = reinterpret_cast<const v4f*>(slabs) + i;
#pragma unroll 8   // independent loads, issued back to back (the adds keep their order)
    for (int s = grp; s < p.splits; s += 4) acc += src[static_cast<size_t>(s) * n4];
  }
  part[grp][col] = acc;
  __syncthreads();
  if (grp == 0 && i < n4) {
    const v4f total = (part[0][col] + part[1][col]) + (part[2][col] + part[3][col]);
    reinterpret_cast<v4f*>(out)[i] = total;
  }
}

// ---- skinny weight gradients -------------------------------------------------------------------
// When one side of gw[N,K] is tiny (the 4 input channels of the first layer, the 4 classes of the
// last), a 128-row MFMA tile would be >95 % padding and the problem is a stream anyway:
//   P[i, j] = sum_m skinny[m, i] * wide[m, j],   i < S <= 9,  j < w (multiple of 4)
// one 16-byte column group of `wide` per thread, S float4 accumulators, row chunks -> per-chunk
// partials -> fixed-order chunk sum (same determinism as the slab path).  With ones_row the last
// P row is sum_m wide[m, :] (the bias gradient when `wide` is g).
constexpr int kSkinnyMax = 9;
constexpr int kSkinnyChunks = 512;

template <int S>
__global__ __launch_bounds__(kBlock) void skinny_wgrad_kernel(const float* __restrict__ skinny, int s_cols,
                                                             const float* __restrict__ wide,
                                                             float* __restrict__ partial, int64_t m,
                                                             int w, int64_t rows_per_chunk) {
  // 256 threads = (w/4 column groups) x (row lanes): with w = 256 four lanes walk interleaved rows
  // of the chunk and are combined through LDS in lane order before the partial is written.
  __shared__ v4f red[kBlock];
  const int cols4 = w >> 2;
  const int lanes = cols4 >= kBlock ? 1 : kBlock / cols4;
  const int rl = cols4 >= kBlock ? 0 : threadIdx.x / cols4;
  const int64_t row0 = blockIdx.x * rows_per_chunk, row1 = min(m, row0 + rows_per_chunk);
  for (int q0 = 0; q0 < cols4; q0 += kBlock) {
    const int q = q0 + (cols4 >= kBlock ? threadIdx.x : threadIdx.x % cols4);
    const bool live = q < cols4 && rl < lanes;
    v4f acc[S];
#pragma unroll
    for (int i = 0; i < S; ++i) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
    if (live) {
#pragma unroll 4
      for (int64_t row = row0 + rl; row < row1; row += lanes) {
        const v4f f = *reinterpret_cast<const v4f*>(wide + static_cast<size_t>(row) * w + 4 * q);
#pragma unroll
        for (int i = 0; i < S; ++i) acc[i] += (i < s_cols ? skinny[row * s_cols + i] : 1.0f) * f;
      }
    }
#pragma unroll
    for (int i = 0; i < S; ++i) {
      v4f total = acc[i];
      if (lanes > 1) {
        __syncthreads();
        red[threadIdx.x] = acc[i];
        __syncthreads();
        if (rl == 0 && live) {
          for (int l = 1; l < lanes; ++l) total += red[l * cols4 + q];
        }
      }
      if (rl == 0 && live)
        *reinterpret_cast<v4f*>(partial + (static_cast<size_t>(blockIdx.x) * S + i) * w + 4 * q) = total;
    }
  }
}

// out = sum over chunks of partial[chunk][rows*w]; element (i, j) goes to dst0[i*w + j] (direct) or
// dst0[j*s_cols + i] (transposed) for i < s_cols, and row s_cols (the ones row) to dst1[j].
// 16 outputs x 16 chunk lanes per workgroup; lane totals are added in lane order (deterministic).
__global__ __launch_bounds__(kBlock) void skinny_sum_kernel(const float* __restrict__ partial,
                                                           float* __restrict__ dst0,
                                                           float* __restrict__ dst1, int rows, int s_cols,
                                                           int w, int chunks, int transposed) {
  __shared__ float part[16][16];
  const int o = threadIdx.x & 15, lane = threadIdx.x >> 4;
  const int idx = blockIdx.x * 16 + o;
  const int total_out = rows * w;
  float acc = 0.f;
  if (idx < total_out) {
#pragma unroll 8   // independent loads, issued back to back (the adds keep their order)
    for (int c = lane; c < chunks; c += 16) acc += partial[static_cast<size_t>(c) * total_out + idx];
  }
  part[lane][o] = acc;
  __syncthreads();
  if (lane != 0 || idx >= total_out) return;
  float sum = part[0][o];
#pragma unroll
  for (int l = 1; l < 16; ++l) sum += part[l][o];
  const int i = idx / w, j = idx - i * w;
  if (i < s_cols)
    dst0[transposed ? j * s_cols + i : i * w + j] = sum;
  else if (dst1 != nullptr)
    dst1[j] = sum;
}

inline int64_t skinny_workspace_floats(int64_t m, int64_t n, int64_t k) {
  const int64_t small = n < k ? n : k, big = n < k ? k : n;
  if (small + 1 > kSkinnyMax || big < 64) return 0;
  return static_cast<int64_t>(kSkinnyChunks) * (small + 1) * big;
}

template <int S>
void launch_skinny(const float* skinny, int s_cols, const float* wide, float* partial, int64_t m, int w,
                   int chunks, int64_t rpc, hipStream_t st) {
  skinny_wgrad_kernel<S><<<chunks, kBlock, 0, st>>>(skinny, s_cols, wide, partial, m, w, rpc);
}

// gw[n,k] (+ gb[n]) of ONE problem through the skinny path; returns false when it does not apply
bool skinny_wgrad(const float* g, const float* a, float* gw, float* gb, float* workspace, int64_t m,
                  int64_t n, int64_t k, hipStream_t st) {
  if (skinny_workspace_floats(m, n, k) == 0) return false;
  const bool g_is_skinny = n < k;          // gw[n,k] = skinny^T wide directly; else transposed
  const float* skinny = g_is_skinny ? g : a;
  const float* wide = g_is_skinny ? a : g;
  const int s_cols = static_cast<int>(g_is_skinny ? n : k), w = static_cast<int>(g_is_skinny ? k : n);
  const bool ones_row = !g_is_skinny && gb != nullptr;   // colsum(wide = g) is the bias gradient
  const int rows = s_cols + (ones_row ? 1 : 0);
  const int64_t rpc = (m + kSkinnyChunks - 1) / kSkinnyChunks;
  const int chunks = static_cast<int>((m + rpc - 1) / rpc);
  switch (rows) {
    case 1: launch_skinny<1>(skinny, s_cols, wide, workspace, m, w, chunks, rpc, st); break;
    case 2: launch_skinny<2>(skinny, s_cols, wide, workspace, m, w, chunks, rpc, st); break;
    case 3: launch_skinny<3>(skinny, s_cols, wide, workspace, m, w, chunks, rpc, st); break;
    case 4: launch_skinny<4>(skinny, s_cols, wide, workspace, m, w, chunks, rpc, st); break;
    case 5: launch_skinny<5>(skinny, s_cols, wide, workspace, m, w, chunks, rpc, st); break;
    case 6: launch_skinny<6>(skinny, s_cols, wide, workspace, m, w, chunks, rpc, st); break;
    case 7: launch_skinny<7>(skinny, s_cols, wide, workspace, m, w, chunks, rpc, st); break;
    case 8: launch_skinny<8>(skinny, s_cols, wide, workspace, m, w, chunks, rpc, st); break;
    default: launch_skinny<9>(skinny, s_cols, wide, workspace, m, w, chunks, rpc, st); break;
  }
  skinny_sum_kernel<<<(rows * w + 15) / 16, kBlock, 0, st>>>(
      workspace, gw, ones_row ? gb : nullptr, rows, s_cols, w, chunks, g_is_skinny ? 0 : 1);
  if (g_is_skinny && gb != nullptr) {
    // bias gradient = column sums of the skinny g [m, n]: the same kernel with a 1-column "ones"
    // operand (s_cols = 0 -> the single P row is sum_m wide) over wide = g needs n % 4 == 0
    float* part = workspace + static_cast<size_t>(chunks) * rows * w;
    skinny_wgrad_kernel<1><<<chunks, kBlock, 0, st>>>(nullptr, 0, g, part, m, static_cast<int>(n), rpc);
    skinny_sum_kernel<<<(static_cast<int>(n) + 15) / 16, kBlock, 0, st>>>(
        part, nullptr, gb, 1, 0, static_cast<int>(n), chunks, 0);
  }
  return true;
}

inline bool aligned4(int64_t x) { return (x & 3) == 0; }

// Tile configurations (runtime-selectable for tuning through gts_set_option).
// Defaults from tools/tune_gemm.py at M = 60 000, 256-wide (profiles/r01_tune_gemm.log).
int g_fwd_variant = -1;    // forward kernels (both operands kk-contiguous); -1 = a one-round tile (10 = 240-row
                           // panels on the 16x16x4 MFMA with direct-to-fragment loads when panels leave fewer rows
                           // per CU, else 8 = double-buffered 256x256) when that fills >= 3/4 of the CUs, else 3
                           // (64x256, two per CU).  profiles/r02_tune_gemm.log: at M = 60 000, 256-wide, K = 256 / 512:
                           // 8: 74.7 / 137.5 us, 9 (same panels through LDS): 73.1 / 134.3, 10: 70.4 / 129.8,
                           // 11 / 12 (one 240 x 64 wave per SIMD, depth 1 / 2): 74.2 / 129.7, 76.0 / 134.1
int g_igrad_variant = 1;   // input-gradient kernels (B kk-strided)
int g_wgrad_variant = -1;  // split-reduction kernel; -1 = chosen per launch by wgrad_plan()

template <int BM, int BN, int WM, int WN, bool AKC, bool BKC, bool DB = false, class Probe = NoProbe>
int launch_tiles(const GemmArgs& p, int grid_y_mult, int splits, hipStream_t st) {
  GemmArgs q = p;
  q.sched = g_gemm_sched;
  q.tiles_n = (p.rb + BN - 1) / BN;
  dim3 grid((p.ra + BM - 1) / BM, q.tiles_n * grid_y_mult, splits);
  gemm_kernel<BM, BN, WM, WN, AKC, BKC, DB, Probe><<<grid, 64 * WM * WN, 0, st>>>(q);
  return launch_status();
}

// Height of the row panels the direct-to-fragment kernel (variant 10) cuts `rows` into.  One workgroup per CU per
// round, so a CU walks rounds x height rows: 240 rows suit 60 000 (250 panels) and 120 000 rows (500 = two rounds),
// but 35 000 rows (the reference's real batches: 6 graphs of ~6k nodes) are 146 panels of 240 on 256 CUs; 144-row
// panels (243 of them) fill the chip.  Candidates 240 / 192 / 144 (5 / 4 / 3 MFMA row blocks per wave); the taller
// panel wins ties (fewer loads per MFMA).  The result of a row does not depend on the height (same reduction order).
int g_panel_rows = 0;   // 0 = automatic; 240 / 192 / 144 force one (tools/tune_gemm.py)
inline int panel_rows_for(int64_t rows, int64_t col_blocks) {
  if (g_panel_rows == 240 || g_panel_rows == 192 || g_panel_rows == 144) return g_panel_rows;
  int best = kR240;
  int64_t best_cost = -1;
  for (int h : {240, 192, 144}) {
    const int64_t panels = (rows + h - 1) / h * col_blocks;
    const int64_t cost = (panels + 255) / 256 * h;
    if (best_cost < 0 || cost < best_cost) best = h, best_cost = cost;
  }
  return best;
}

template <bool AKC, bool BKC>
int pick_plain_variant(const GemmArgs& p) {
  int variant = BKC ? g_fwd_variant : g_igrad_variant;
  if (p.rb <= 128 && variant < 9) return 0;   // narrow outputs: the 128 x 64 / 128 x 128 tiles (unless a panel kernel is forced)
  const bool row_count_invariant = variant == -2;   // automatic, but only 32x32x2 tiles: one reduction order
  if (variant < 0) {
    const int64_t cols = (p.rb + 255) / 256;
    const int64_t big_tiles = static_cast<int64_t>((p.ra + 255) / 256) * cols;
    variant = big_tiles >= 192 ? 8 : 3;
    if constexpr (AKC && BKC) {
      // one workgroup per CU, in rounds of 256: rows a CU walks with 256-row tiles vs 240-row panels
      const int h = panel_rows_for(p.ra, cols);
      const int64_t panels = static_cast<int64_t>((p.ra + h - 1) / h) * cols;
      const int64_t rows256 = (big_tiles + 255) / 256 * 256, rows_panel = (panels + 255) / 256 * h;
      if (variant == 8 && rows_panel < rows256 && !row_count_invariant) variant = 10;
      // shorter operands: panels that fill at least 3/4 of the CUs in their one round beat two half-empty rounds of 64 x 256 tiles
      if (variant == 3 && panels >= 192 && panels <= 256 && !row_count_invariant) variant = 10;
    }
  }
  return variant;
}

// bits[((col / 64) * ceil(rows / 4) + row / 4) * 4 + e], bit 16 (row % 4) + (col % 64) / 4  <=>  c[row][64 (col / 64) + 4 ((col % 64) / 4) + e] > 0:
// the layout the panel kernels write from their epilogue, here for the outputs of the other tile variants
__global__ __launch_bounds__(256) void relu_bits_kernel(const float* __restrict__ c, unsigned long long* __restrict__ bits,
                                                        int rows, int cols) {
  const int lane = threadIdx.x & 63;
  const int blocks = cols >> 6;
  const long long id = static_cast<long long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  const long long group = id / blocks;
  const int cb = static_cast<int>(id % blocks);
  if (group * 4 >= rows) return;
  const long long row = group * 4 + (lane >> 4);
  v4f v = {0.f, 0.f, 0.f, 0.f};
  if (row < rows) v = *reinterpret_cast<const v4f*>(c + row * cols + 64 * cb + 4 * (lane & 15));
  const unsigned long long w0 = __ballot(v[0] > 0.f), w1 = __ballot(v[1] > 0.f);
  const unsigned long long w2 = __ballot(v[2] > 0.f), w3 = __ballot(v[3] > 0.f);
  const long long groups = (rows + 3) >> 2;
  if (lane < 4) bits[(cb * groups + group) * 4 + lane] = lane == 0 ? w0 : lane == 1 ? w1 : lane == 2 ? w2 : w3;
}

// A few inputs to a few outputs per node (fc_pool of the first layer: 4 -> 4 on 60 000 rows): one lane per row,
// the row's <= 16 inputs as 16-byte loads (consecutive lanes = consecutive rows: contiguous), the weights at
// wave-uniform addresses, fused multiply-adds in reduction order.  A 128 x 64 MFMA tile would multiply zeros for
// 23 us here; this is 0.5 MB of traffic.
template <int N>
__global__ __launch_bounds__(256) void tiny_fwd_kernel(const GemmArgs p) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= p.ra) return;
  float acc[N];
#pragma unroll
  for (int n = 0; n < N; ++n) acc[n] = p.bias != nullptr ? p.bias[n] : 0.f;
#pragma unroll
  for (int seg = 0; seg < 2; ++seg) {
    const float* a = p.a[seg] + static_cast<size_t>(row) * p.lda[seg];
    for (int k = 0; k < p.kseg[seg]; k += 4) {
      const v4f x = *reinterpret_cast<const v4f*>(a + k);
#pragma unroll
      for (int n = 0; n < N; ++n) {
        const v4f w = *reinterpret_cast<const v4f*>(p.b[seg] + n * p.ldb[seg] + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[n] = __builtin_fmaf(x[e], w[e], acc[n]);
      }
    }
  }
  float* out = p.c + static_cast<size_t>(row) * p.ldc;
#pragma unroll
  for (int n = 0; n < N; n += 4) {
    v4f o = {acc[n], acc[n + 1], acc[n + 2], acc[n + 3]};
    if (p.relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
    }
    *reinterpret_cast<v4f*>(out + n) = o;
  }
}

// Wide inputs to 4 or 8 outputs per node (the classifier layer: 256 + 256 -> 4; g @ W_neigh of the first layer:
// 256 -> 4): HBM-bound row streaming.  A wave takes four rows at a time; lane l holds reduction indices
// 4 l .. 4 l + 3 (+ 256 per further chunk) of each row and of every weight row, accumulates its partial dot
// products in a fixed order and the 64 partials meet in an xor butterfly — the same sum for a row whatever
// the batch holds.  (The 128 x 64 MFMA tile reads the same bytes at 3.1 TB/s: 40 us for the pair at C2.)
template <int N, bool BKC>
__global__ __launch_bounds__(256) void skinny_fwd_kernel(const GemmArgs p) {
  constexpr int R = 4;
  const int lane = threadIdx.x & 63;
  const long long wave = static_cast<long long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  const long long row0 = wave * R;
  if (row0 >= p.ra) return;
  float acc[R][N];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int n = 0; n < N; ++n) acc[r][n] = 0.f;
#pragma unroll
  for (int seg = 0; seg < 2; ++seg) {
    const int kseg = p.kseg[seg];
    for (int k = 4 * lane; k < kseg; k += 256) {
      v4f w[N];   // w[n][e] = B(n, k + e)
      if constexpr (BKC) {
#pragma unroll
        for (int n = 0; n < N; ++n) w[n] = *reinterpret_cast<const v4f*>(p.b[seg] + static_cast<size_t>(n) * p.ldb[seg] + k);
      } else {      // weights [K, N] with ldb == N: the four reduction rows of this lane are 16 N contiguous bytes
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int n4 = 0; n4 < N; n4 += 4) {
            const v4f t = *reinterpret_cast<const v4f*>(p.b[seg] + static_cast<size_t>(k + e) * p.ldb[seg] + n4);
#pragma unroll
            for (int j = 0; j < 4; ++j) w[n4 + j][e] = t[j];
          }
      }
      v4f x[R];
#pragma unroll
      for (int r = 0; r < R; ++r)
        x[r] = row0 + r < p.ra ? *reinterpret_cast<const v4f*>(p.a[seg] + static_cast<size_t>(row0 + r) * p.lda[seg] + k)
                               : v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int n = 0; n < N; ++n)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[r][n] = __builtin_fmaf(x[r][e], w[n][e], acc[r][n]);
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc[r][n] += __shfl_xor(acc[r][n], o, kWave);
  // lane r * N / 4 + n / 4 stores the float4 (r, n..n+3): every lane holds every total
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int n = 0; n < N; n += 4) {
      if (lane == r * (N / 4) + n / 4 && row0 + r < p.ra) {
        v4f o = {acc[r][n], acc[r][n + 1], acc[r][n + 2], acc[r][n + 3]};
        if (p.bias != nullptr) o += *reinterpret_cast<const v4f*>(p.bias + n);
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
        }
        *reinterpret_cast<v4f*>(p.c + static_cast<size_t>(row0 + r) * p.ldc + n) = o;
      }
    }
}

template <bool BKC>
inline bool skinny_forward(const GemmArgs& p) {
  const bool automatic = BKC ? g_fwd_variant < 0 : g_igrad_variant <= 1;   // no tile variant forced (the input gradient's default is 1)
  return automatic && (p.rb == 4 || p.rb == 8) && p.ldc == p.rb && p.kseg[0] + p.kseg[1] >= 64 && p.mask == nullptr &&
         p.c2 == nullptr && p.sc_l == nullptr && p.bits_out == nullptr &&
         (BKC || (p.ldb[0] == p.rb && (p.kseg[1] == 0 || p.ldb[1] == p.rb)));
}

inline bool tiny_forward(const GemmArgs& p) {
  return (p.rb == 4 || p.rb == 8) && p.ldc == p.rb && p.kseg[0] + p.kseg[1] <= 16 && p.mask == nullptr && p.c2 == nullptr &&
         p.sc_l == nullptr && p.bits_out == nullptr && g_fwd_variant < 0;
}

template <bool AKC, bool BKC>
int launch_plain_tiles(const GemmArgs& p, int variant, hipStream_t st) {
  if constexpr (AKC && BKC) {
    if (tiny_forward(p)) {
      const unsigned blocks = static_cast<unsigned>((p.ra + 255) / 256);
      if (p.rb == 4) tiny_fwd_kernel<4><<<blocks, 256, 0, st>>>(p);
      else tiny_fwd_kernel<8><<<blocks, 256, 0, st>>>(p);
      return launch_status();
    }
  }
  if constexpr (AKC) {
    if (skinny_forward<BKC>(p)) {
      const unsigned blocks = static_cast<unsigned>((p.ra + 15) / 16);   // four waves of four rows
      if (p.rb == 4) skinny_fwd_kernel<4, BKC><<<blocks, 256, 0, st>>>(p);
      else skinny_fwd_kernel<8, BKC><<<blocks, 256, 0, st>>>(p);
      return launch_status();
    }
  }
  if (p.rb <= 64) return launch_tiles<128, 64, 2, 2, AKC, BKC>(p, 1, 1, st);
  if (p.rb <= 128) return launch_tiles<128, 128, 2, 2, AKC, BKC>(p, 1, 1, st);
  switch (variant) {
    // the variants that survived the sweeps in profiles/r01_tune_gemm.log (numbers kept from there)
    case 3: return launch_tiles<64, 256, 2, 4, AKC, BKC>(p, 1, 1, st);
    case 5: return launch_tiles<256, 128, 4, 2, AKC, BKC>(p, 1, 1, st);
    case 8: return launch_tiles<256, 256, 4, 4, AKC, BKC, true>(p, 1, 1, st);
    default: return launch_tiles<128, 256, 2, 4, AKC, BKC>(p, 1, 1, st);   // 1
  }
}

template <bool AKC, bool BKC>
int launch_plain(const GemmArgs& p, hipStream_t st) {
  const int variant = pick_plain_variant<AKC, BKC>(p);
  if constexpr (AKC && BKC) {
    if (p.c2 != nullptr && !(variant == 10 && p.rb <= kC240 && p.rb2 <= kC240 && p.rb2 > 128)) {
      // not a one-panel-per-row-block launch (or a narrow second output, which a launch of its own
      // would give to the 32x32x2 tiles): the chained GEMM runs as a launch of its own — same bits either way
      GemmArgs first = p, next{};
      first.c2 = nullptr;
      int rc = launch_plain<true, true>(first, st);
      if (rc != GTS_OK) return rc;
      next.a[0] = next.a[1] = p.c, next.b[0] = next.b[1] = p.b2, next.bp[0] = p.bp2;
      first.bp2 = nullptr;
      next.lda[0] = next.lda[1] = p.ldc, next.ldb[0] = next.ldb[1] = p.ldb2;
      next.kseg[0] = p.rb, next.kseg[1] = 0;
      next.ra = p.ra, next.rb = p.rb2, next.c = p.c2, next.ldc = p.ldc2, next.bias = p.bias2, next.relu = p.relu2;
      next.tiles_per_split = (p.rb + kBK - 1) / kBK;
      return launch_plain<true, true>(next, st);
    }
    // the panel kernels keep the mask bits themselves (written / read in their epilogue)
    if (variant == 10) {
      switch (panel_rows_for(p.ra, (p.rb + kC240 - 1) / kC240)) {
        case 144: return launch_panel_direct<3, 4, 1, NoProbe, 144>(p, st);
        case 192: return launch_panel_direct<3, 4, 1, NoProbe, 192>(p, st);
        default: return launch_panel_direct<3, 4, 1>(p, st);
      }
    }
  }
  // every other tile: the float mask is read as before, and the bits of the output come from a pass of their own
  int rc = launch_plain_tiles<AKC, BKC>(p, variant, st);
  if (rc != GTS_OK || p.bits_out == nullptr) return rc;
  const long long words = static_cast<long long>((p.ra + 3) / 4) * (p.rb >> 6);
  relu_bits_kernel<<<static_cast<unsigned>((words + 3) / 4), 256, 0, st>>>(p.c, p.bits_out, p.ra, p.rb);
  return launch_status();
}

// Split-reduction plan of a weight-gradient launch: tile variant, its edge lengths, and how many
// ways the reduction over the M nodes is split (also sizes the workspace).
struct WgradPlan {
  int variant, bm, bn, splits;
};

inline void wgrad_candidate(int variant, int64_t k, int* bm, int* bn, int64_t* slots) {
  *bm = 128, *bn = k <= 64 ? 64 : 128, *slots = 512;   // 2 workgroups per CU
  if (k <= 64) return;
  if (variant == 2) *bn = 256;
  if (variant == 4 || variant == 6) *bm = 256, *bn = 256, *slots = 256;  // one workgroup per CU
}

inline WgradPlan wgrad_plan(int64_t m, int64_t n, int64_t k, int n_problems) {
  const int64_t tiles = (m + kBK - 1) / kBK;
  // automatic (in-bench sweeps, profiles/r01_tune_gemm.log): up to 4 problems (one layer) ->
  // 128x128 tiles (1): 42 splits of the three 256x256 problems fill 504 of 512 slots with half the
  // slab traffic of the wider tiles.  More problems (a whole layer stack at once) -> of the
  // double-buffered 256x256 tile (4) and the 128x256 tile (2), the one whose workgroup count
  // (output tiles x splits, never more than the slots: no lone tail round) fills the chip best;
  // 19 problems: variant 4, 13 splits, 247 of 256 slots, 144 reduction tiles per workgroup.
  // Few problems with LARGE outputs (GAT: one or two 1024 x 1024 gradients per layer, 16 output
  // tiles of 256 x 256 x 16 splits = 256 workgroups): the double-buffered 256x256 tile again
  // (C3 308 -> 312 graphs/s, profiles/r02_ab_c3_wgrad.log).
  // Round 3: 6 (wgrad_stream_kernel: the same tile and bits as 4 with a main loop free of vector instructions) takes
  // the place of 4 — 1 094 against 1 193 us for the 19 problems of C2 (profiles/r03_tune_wgrad.log).
  static const int kMany[2] = {6, 2};
  static const int kFewLarge[2] = {6, 1};
  const bool few = n_problems <= 4;
  const bool large = n * k * n_problems >= (1 << 20);
  const int n_candidates = g_wgrad_variant >= 0 || (few && !large) ? 1 : 2;
  WgradPlan best{};
  double best_fill = -1.0;
  for (int c = 0; c < n_candidates; ++c) {
    WgradPlan plan{};
    plan.variant = g_wgrad_variant >= 0 ? g_wgrad_variant : (few ? (large ? kFewLarge[c] : 1) : kMany[c]);
    int64_t slots;
    wgrad_candidate(plan.variant, k, &plan.bm, &plan.bn, &slots);
    const int64_t out_tiles = ((n + plan.bm - 1) / plan.bm) * ((k + plan.bn - 1) / plan.bn) * n_problems;
    int64_t splits = out_tiles >= slots ? 1 : slots / out_tiles;
    if (splits > tiles) splits = tiles;
    plan.splits = static_cast<int>(splits < 1 ? 1 : splits);
    const int64_t groups = out_tiles * plan.splits;
    const int64_t rounds = (groups + slots - 1) / slots;
    const double fill = static_cast<double>(groups) / static_cast<double>(rounds * slots);
    if (fill > best_fill + 1e-9) best = plan, best_fill = fill;
  }
  return best;
}

int launch_wgrad(const GemmArgs& p, const WgradPlan& plan, hipStream_t st) {
  const int np = p.n_problems, splits = plan.splits;
  if (p.rb <= 64) return launch_tiles<128, 64, 2, 2, false, false>(p, np, splits, st);
  switch (plan.variant) {
    case 2: return launch_tiles<128, 256, 2, 4, false, false>(p, np, splits, st);
    case 4: return launch_tiles<256, 256, 4, 4, false, false, true>(p, np, splits, st);
    case 6: {
      GemmArgs q = p;
      q.tiles_n = (p.rb + 255) / 256;
      dim3 grid((p.ra + 255) / 256, q.tiles_n * np, splits);
      wgrad_stream_kernel<><<<grid, 1024, 0, st>>>(q);
      return launch_status();
    }
    default: return launch_tiles<128, 128, 2, 4, false, false>(p, np, splits, st);   // 1
  }
}

}  // namespace
}  // namespace gts

extern "C" int32_t gts_set_option(int32_t option, int32_t value) {
  switch (option) {
    case GTS_OPT_GEMM_TILE:      // the forms the library carries (the rejected ones live in tools/diag/gemm_rejected_forms.inc)
      if (value != -1 && value != -2 && value != 1 && value != 3 && value != 5 && value != 8 && value != 10) return GTS_ERR_ARGKIND;
      gts::g_fwd_variant = value;
      return GTS_OK;
    case GTS_OPT_IGRAD_TILE:
      if (value != -1 && value != 1 && value != 3 && value != 5 && value != 8 && value != 10) return GTS_ERR_ARGKIND;
      gts::g_igrad_variant = value;
      return GTS_OK;
    case GTS_OPT_SPMM_ROWS_PER_WAVE: gts::g_spmm_seq = value; return GTS_OK;
    case GTS_OPT_SPMM_STREAMING: gts::g_spmm_nt = value; return GTS_OK;
    case GTS_OPT_PROJECT_STREAMING: gts::g_project_nt = value; return GTS_OK;
    case GTS_OPT_WGRAD_TILE:
      if (value != -1 && value != 1 && value != 2 && value != 4 && value != 6) return GTS_ERR_ARGKIND;
      gts::g_wgrad_variant = value;
      return GTS_OK;
    case GTS_OPT_GEMM_SCHED: gts::g_gemm_sched = value; return GTS_OK;
    case GTS_OPT_CLUSTER_STREAMING: gts::g_cluster_nt = value; return GTS_OK;
    case GTS_OPT_PANEL_ROWS: gts::g_panel_rows = value; return GTS_OK;
    case GTS_OPT_GAT_WALK: gts::g_gat_walk = value; return GTS_OK;
    case GTS_OPT_GAT_CLUSTER_WAVES: gts::g_gat_cluster_waves = value; return GTS_OK;
    case GTS_OPT_GAT_CLUSTER_GROUP: gts::g_gat_cluster_group = value; return GTS_OK;
    case GTS_OPT_CLUSTER_DEALING:
      if (value < 0 || value > 2) return GTS_ERR_ARGKIND;
      gts::g_cluster_dealing = value;
      return GTS_OK;
    case GTS_OPT_GAT_CLUSTER_DEALING:
      if (value != 0 && value != 1) return GTS_ERR_ARGKIND;
      gts::g_gat_cluster_dealing = value;
      return GTS_OK;
    case GTS_OPT_CLUSTER_RING: gts::g_cluster_ring = value; return GTS_OK;
    case GTS_OPT_CLUSTER_PER_CU: gts::g_cluster_per_cu = value; return GTS_OK;
    case GTS_OPT_CLUSTER_CONSUMERS: gts::g_cluster_consumers = value; return GTS_OK;
    default: return GTS_ERR_ARGKIND;
  }
}

extern "C" int32_t gts_get_option(int32_t option) {
  switch (option) {
    case GTS_OPT_GEMM_TILE: return gts::g_fwd_variant;
    case GTS_OPT_IGRAD_TILE: return gts::g_igrad_variant;
    case GTS_OPT_SPMM_ROWS_PER_WAVE: return gts::g_spmm_seq;
    case GTS_OPT_SPMM_STREAMING: return gts::g_spmm_nt;
    case GTS_OPT_PROJECT_STREAMING: return gts::g_project_nt;
    case GTS_OPT_WGRAD_TILE: return gts::g_wgrad_variant;
    case GTS_OPT_GEMM_SCHED: return gts::g_gemm_sched;
    case GTS_OPT_CLUSTER_STREAMING: return gts::g_cluster_nt;
    case GTS_OPT_PANEL_ROWS: return gts::g_panel_rows;
    case GTS_OPT_GAT_WALK: return gts::g_gat_walk;
    case GTS_OPT_GAT_CLUSTER_WAVES: return gts::g_gat_cluster_waves;
    case GTS_OPT_GAT_CLUSTER_GROUP: return gts::g_gat_cluster_group;
    case GTS_OPT_GAT_CLUSTER_DEALING: return gts::g_gat_cluster_dealing;
    case GTS_OPT_CLUSTER_DEALING: return gts::g_cluster_dealing;
    case GTS_OPT_CLUSTER_RING: return gts::g_cluster_ring;
    case GTS_OPT_CLUSTER_PER_CU: return gts::g_cluster_per_cu;
    case GTS_OPT_CLUSTER_CONSUMERS: return gts::g_cluster_consumers;
    default: return INT32_MIN;
  }
}

extern "C" int32_t gts_linear_fwd_f32(const float* a0, const float* w0, const float* a1,
                                      const float* w1, const float* bias, float* out, int64_t m,
                                      int64_t n, int64_t k0, int64_t k1, int32_t relu,
                                      uint64_t* relu_bits, const float* const* packed, void* stream) {
  using namespace gts;
  if (!a0 || !w0 || !out || ((a1 == nullptr) != (w1 == nullptr))) return GTS_ERR_NULL;
  if (m < 0 || n <= 0 || k0 <= 0 || k1 < 0 || m >= (1LL << 31) || n >= (1 << 20) ||
      k0 >= (1 << 20) || k1 >= (1 << 20) || !aligned4(k0) || !aligned4(k1) || (a1 && k1 == 0) ||
      (relu_bits && n % 64 != 0))
    return GTS_ERR_SHAPE;
  if (m == 0) return GTS_OK;
  GemmArgs p{};
  p.a[0] = a0, p.b[0] = w0, p.lda[0] = static_cast<int>(k0), p.ldb[0] = static_cast<int>(k0);
  p.kseg[0] = static_cast<int>(k0);
  p.a[1] = a1 ? a1 : a0, p.b[1] = w1 ? w1 : w0;
  p.lda[1] = p.ldb[1] = static_cast<int>(k1), p.kseg[1] = a1 ? static_cast<int>(k1) : 0;
  p.ra = static_cast<int>(m), p.rb = static_cast<int>(n), p.c = out, p.ldc = static_cast<int>(n);
  p.bias = bias, p.relu = relu;
  p.bits_out = reinterpret_cast<unsigned long long*>(relu_bits);
  p.tiles_per_split = (p.kseg[0] + kBK - 1) / kBK + (p.kseg[1] + kBK - 1) / kBK;
  if (packed != nullptr) p.bp[0] = packed[0], p.bp[1] = a1 ? packed[1] : nullptr;
  return launch_plain<true, true>(p, static_cast<hipStream_t>(stream));
}

extern "C" int32_t gts_relu_bits_pay(int64_t m, int64_t n) {
  using namespace gts;
  if (m <= 0 || n <= 0 || n % 64 != 0 || m >= (1LL << 31) || n >= (1 << 20)) return 0;
  GemmArgs p{};
  p.ra = static_cast<int>(m), p.rb = static_cast<int>(n);
  const int variant = pick_plain_variant<true, true>(p);
  return variant >= 10 && variant <= 12 ? 1 : 0;
}

extern "C" int64_t gts_relu_bits_bytes(int64_t m, int64_t n) {
  if (m < 0 || n <= 0 || n % 64 != 0) return 0;
  return (m + 3) / 4 * (n / 64) * 4 * static_cast<int64_t>(sizeof(uint64_t));
}

namespace gts {
namespace {
// out[i] = sum_p in[i * parts + p] (fixed order), two arrays in one launch
__global__ __launch_bounds__(kBlock) void sum_parts_kernel(const float* __restrict__ in_l, const float* __restrict__ in_r,
                                                           float* __restrict__ out_l, float* __restrict__ out_r,
                                                           int64_t n, int parts) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  float sl = 0.f, sr = 0.f;
  for (int q = 0; q < parts; ++q) sl += in_l[i * parts + q], sr += in_r[i * parts + q];
  out_l[i] = sl, out_r[i] = sr;
}
}  // namespace
}  // namespace gts

extern "C" int32_t gts_gat_scores_f32(const float* ft, const float* attn_l, const float* attn_r, float* el, float* er,
                                      int64_t n, int64_t heads, int64_t dim, void* stream);

extern "C" int64_t gts_gat_fc_scores_workspace(int64_t m, int64_t heads, int64_t dim) {
  if (m <= 0 || heads <= 0 || dim <= 0 || dim % 64 != 0) return 0;
  return 2 * m * heads * (dim / 64) * static_cast<int64_t>(sizeof(float));
}

extern "C" int32_t gts_gat_fc_scores_f32(const float* h, const float* w_fc, const float* attn_l, const float* attn_r,
                                         float* ft, float* el, float* er, float* workspace, int64_t workspace_bytes,
                                         int64_t m, int64_t heads, int64_t dim, int64_t k, const float* w_fc_packed,
                                         void* stream) {
  using namespace gts;
  if (!h || !w_fc || !attn_l || !attn_r || !ft || !el || !er) return GTS_ERR_NULL;
  const int64_t n = heads * dim;
  if (m < 0 || heads <= 0 || dim <= 0 || k <= 0 || m >= (1LL << 31) || n >= (1 << 20) || k >= (1 << 20) || !aligned4(k))
    return GTS_ERR_SHAPE;
  if (m == 0) return GTS_OK;
  GemmArgs p{};
  p.a[0] = p.a[1] = h, p.b[0] = p.b[1] = w_fc, p.lda[0] = p.ldb[0] = p.lda[1] = p.ldb[1] = static_cast<int>(k);
  p.kseg[0] = static_cast<int>(k), p.kseg[1] = 0;
  p.ra = static_cast<int>(m), p.rb = static_cast<int>(n), p.c = ft, p.ldc = static_cast<int>(n);
  p.tiles_per_split = (p.kseg[0] + kBK - 1) / kBK;
  p.bp[0] = w_fc_packed;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int parts = static_cast<int>(dim / 64);
  const bool fuse = dim % 64 == 0 && pick_plain_variant<true, true>(p) == 10 &&
                    (parts == 1 || (workspace && workspace_bytes >= gts_gat_fc_scores_workspace(m, heads, dim)));
  if (!fuse) {   // small problems / odd head widths: the GEMM, then the scores in a pass of their own
    const int rc = launch_plain<true, true>(p, st);
    return rc != GTS_OK ? rc : gts_gat_scores_f32(ft, attn_l, attn_r, el, er, m, heads, dim, stream);
  }
  p.sc_l = attn_l, p.sc_r = attn_r;
  p.sc_el = parts == 1 ? el : workspace, p.sc_er = parts == 1 ? er : workspace + m * heads * parts;
  int rc = launch_plain<true, true>(p, st);
  if (rc != GTS_OK || parts == 1) return rc;
  const int64_t rows = m * heads;
  sum_parts_kernel<<<static_cast<unsigned>((rows + kBlock - 1) / kBlock), kBlock, 0, st>>>(p.sc_el, p.sc_er, el, er, rows, parts);
  return launch_status();
}

extern "C" int32_t gts_linear_fwd_chain_f32(const float* a0, const float* w0, const float* a1,
                                            const float* w1, const float* bias, float* out,
                                            const float* w2, const float* bias2, float* out2, int64_t m,
                                            int64_t n, int64_t k0, int64_t k1, int32_t relu, int64_t n2,
                                            int32_t relu2, uint64_t* relu_bits, const float* const* packed,
                                            void* stream) {
  using namespace gts;
  if (!a0 || !w0 || !out || !w2 || !out2 || ((a1 == nullptr) != (w1 == nullptr))) return GTS_ERR_NULL;
  if (m < 0 || n <= 0 || n2 <= 0 || k0 <= 0 || k1 < 0 || m >= (1LL << 31) || n >= (1 << 20) || n2 >= (1 << 20) ||
      k0 >= (1 << 20) || k1 >= (1 << 20) || !aligned4(k0) || !aligned4(k1) || !aligned4(n) || (a1 && k1 == 0) ||
      (relu_bits && n % 64 != 0))
    return GTS_ERR_SHAPE;
  if (m == 0) return GTS_OK;
  GemmArgs p{};
  p.a[0] = a0, p.b[0] = w0, p.lda[0] = static_cast<int>(k0), p.ldb[0] = static_cast<int>(k0);
  p.kseg[0] = static_cast<int>(k0);
  p.a[1] = a1 ? a1 : a0, p.b[1] = w1 ? w1 : w0;
  p.lda[1] = p.ldb[1] = static_cast<int>(k1), p.kseg[1] = a1 ? static_cast<int>(k1) : 0;
  p.ra = static_cast<int>(m), p.rb = static_cast<int>(n), p.c = out, p.ldc = static_cast<int>(n);
  p.bias = bias, p.relu = relu;
  p.bits_out = reinterpret_cast<unsigned long long*>(relu_bits);
  p.tiles_per_split = (p.kseg[0] + kBK - 1) / kBK + (p.kseg[1] + kBK - 1) / kBK;
  p.b2 = w2, p.bias2 = bias2, p.c2 = out2, p.rb2 = static_cast<int>(n2), p.ldb2 = static_cast<int>(n);
  p.ldc2 = static_cast<int>(n2), p.relu2 = relu2;
  if (packed != nullptr) p.bp[0] = packed[0], p.bp[1] = a1 ? packed[1] : nullptr, p.bp2 = packed[2];
  return launch_plain<true, true>(p, static_cast<hipStream_t>(stream));
}

extern "C" int32_t gts_linear_bwd_input_chain_t_f32(const float* g0, const float* w0t, const float* g1,
                                                    const float* w1t, const float* relu_mask,
                                                    const uint64_t* relu_bits, float* gin,
                                                    const float* w2t, float* gin2, int64_t m, int64_t k,
                                                    int64_t n0, int64_t n1, int64_t k2,
                                                    const float* const* packed, void* stream) {
  using namespace gts;
  if (!g0 || !w0t || !gin || !w2t || !gin2 || ((g1 == nullptr) != (w1t == nullptr)) || (relu_bits && !relu_mask))
    return GTS_ERR_NULL;
  if (m < 0 || k <= 0 || k2 <= 0 || n0 <= 0 || n1 < 0 || m >= (1LL << 31) || k >= (1 << 20) || k2 >= (1 << 20) ||
      n0 >= (1 << 20) || n1 >= (1 << 20) || !aligned4(k) || !aligned4(n0) || !aligned4(n1) || (g1 && n1 == 0) ||
      (relu_bits && k % 64 != 0))
    return GTS_ERR_SHAPE;
  if (m == 0) return GTS_OK;
  GemmArgs p{};
  p.a[0] = g0, p.b[0] = w0t, p.lda[0] = p.ldb[0] = static_cast<int>(n0), p.kseg[0] = static_cast<int>(n0);
  p.a[1] = g1 ? g1 : g0, p.b[1] = w1t ? w1t : w0t;
  p.lda[1] = p.ldb[1] = static_cast<int>(n1), p.kseg[1] = g1 ? static_cast<int>(n1) : 0;
  p.ra = static_cast<int>(m), p.rb = static_cast<int>(k), p.c = gin, p.ldc = static_cast<int>(k);
  p.mask = relu_mask, p.bits_in = reinterpret_cast<const unsigned long long*>(relu_bits);
  p.tiles_per_split = (p.kseg[0] + kBK - 1) / kBK + (p.kseg[1] + kBK - 1) / kBK;
  p.b2 = w2t, p.bias2 = nullptr, p.c2 = gin2, p.rb2 = static_cast<int>(k2), p.ldb2 = static_cast<int>(k);
  p.ldc2 = static_cast<int>(k2), p.relu2 = 0;
  if (packed != nullptr) p.bp[0] = packed[0], p.bp[1] = g1 ? packed[1] : nullptr, p.bp2 = packed[2];
  return launch_plain<true, true>(p, static_cast<hipStream_t>(stream));
}

extern "C" int32_t gts_linear_bwd_input_f32(const float* g0, const float* w0, const float* g1,
                                            const float* w1, const float* relu_mask, float* gin,
                                            int64_t m, int64_t k, int64_t n0, int64_t n1,
                                            void* stream) {
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
  p.mask = relu_mask;
  p.tiles_per_split = (p.kseg[0] + kBK - 1) / kBK + (p.kseg[1] + kBK - 1) / kBK;
  return launch_plain<true, false>(p, static_cast<hipStream_t>(stream));
}

extern "C" int32_t gts_linear_bwd_input_t_f32(const float* g0, const float* w0t, const float* g1,
                                              const float* w1t, const float* relu_mask,
                                              const uint64_t* relu_bits, float* gin,
                                              int64_t m, int64_t k, int64_t n0, int64_t n1,
                                              const float* const* packed, void* stream) {
  using namespace gts;
  if (!g0 || !w0t || !gin || ((g1 == nullptr) != (w1t == nullptr)) || (relu_bits && !relu_mask)) return GTS_ERR_NULL;
  if (m < 0 || k <= 0 || n0 <= 0 || n1 < 0 || m >= (1LL << 31) || k >= (1 << 20) ||
      n0 >= (1 << 20) || n1 >= (1 << 20) || !aligned4(k) || !aligned4(n0) || !aligned4(n1) ||
      (g1 && n1 == 0) || (relu_bits && k % 64 != 0))
    return GTS_ERR_SHAPE;
  if (m == 0) return GTS_OK;
  GemmArgs p{};
  // C[m, k] = sum_n g[m, n] * Wt[k, n]: the forward form (both operands reduction-contiguous)
  p.a[0] = g0, p.b[0] = w0t, p.lda[0] = p.ldb[0] = static_cast<int>(n0), p.kseg[0] = static_cast<int>(n0);
  p.a[1] = g1 ? g1 : g0, p.b[1] = w1t ? w1t : w0t;
  p.lda[1] = p.ldb[1] = static_cast<int>(n1), p.kseg[1] = g1 ? static_cast<int>(n1) : 0;
  p.ra = static_cast<int>(m), p.rb = static_cast<int>(k), p.c = gin, p.ldc = static_cast<int>(k);
  p.mask = relu_mask, p.bits_in = reinterpret_cast<const unsigned long long*>(relu_bits);
  p.tiles_per_split = (p.kseg[0] + kBK - 1) / kBK + (p.kseg[1] + kBK - 1) / kBK;
  if (packed != nullptr) p.bp[0] = packed[0], p.bp[1] = g1 ? packed[1] : nullptr;
  return launch_plain<true, true>(p, static_cast<hipStream_t>(stream));
}

// The panel launch that can carry an activation backward + column sums in its epilogue: tall, whole 256-column blocks.
static bool act_fold_in_epilogue(const gts::GemmArgs& p) {
  return gts::pick_plain_variant<true, true>(p) == 10 && p.rb % gts::kC240 == 0;
}

extern "C" int64_t gts_linear_bwd_input_t_act_workspace(int64_t m, int64_t k) {
  if (m <= 0 || k <= 0) return 0;
  const int64_t row_blocks = (m + 143) / 144 * 3;   // the shortest panel: most row blocks
  const int64_t pass = gts_gat_reduce_workspace(m, k) / 2;
  const int64_t fold = row_blocks * k * static_cast<int64_t>(sizeof(float));
  return fold > pass ? fold : pass;
}

extern "C" int32_t gts_linear_bwd_input_t_act_f32(const float* g0, const float* w0t, const float* g1, const float* w1t,
                                                  const float* act_out, int32_t activation, float* gin, float* g_bias,
                                                  float* workspace, int64_t workspace_bytes, int64_t m, int64_t k,
                                                  int64_t n0, int64_t n1, const float* const* packed, void* stream) {
  using namespace gts;
  if (!g0 || !w0t || !gin || !act_out || ((g1 == nullptr) != (w1t == nullptr)) || (g_bias && !workspace)) return GTS_ERR_NULL;
  if (activation != 1 && activation != 2) return GTS_ERR_ARGKIND;
  if (m < 0 || k <= 0 || n0 <= 0 || n1 < 0 || m >= (1LL << 31) || k >= (1 << 20) || n0 >= (1 << 20) || n1 >= (1 << 20) ||
      !aligned4(k) || !aligned4(n0) || !aligned4(n1) || (g1 && n1 == 0))
    return GTS_ERR_SHAPE;
  if (g_bias && workspace_bytes < gts_linear_bwd_input_t_act_workspace(m, k)) return GTS_ERR_SHAPE;
  if (m == 0) return GTS_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  GemmArgs p{};
  p.a[0] = g0, p.b[0] = w0t, p.lda[0] = p.ldb[0] = static_cast<int>(n0), p.kseg[0] = static_cast<int>(n0);
  p.a[1] = g1 ? g1 : g0, p.b[1] = w1t ? w1t : w0t;
  p.lda[1] = p.ldb[1] = static_cast<int>(n1), p.kseg[1] = g1 ? static_cast<int>(n1) : 0;
  p.ra = static_cast<int>(m), p.rb = static_cast<int>(k), p.c = gin, p.ldc = static_cast<int>(k);
  p.tiles_per_split = (p.kseg[0] + kBK - 1) / kBK + (p.kseg[1] + kBK - 1) / kBK;
  if (packed != nullptr) p.bp[0] = packed[0], p.bp[1] = g1 ? packed[1] : nullptr;
  if (activation == 1 && g_bias && act_fold_in_epilogue(p)) {
    const int rows = panel_rows_for(p.ra, p.rb / kC240);
    const int row_blocks = (p.ra + rows - 1) / rows * 3;
    p.mask = act_out, p.mask_kind = 1, p.col_partial = workspace;
    const int rc = launch_plain<true, true>(p, st);
    if (rc != GTS_OK) return rc;
    return sum_chunks(workspace, g_bias, p.rb, row_blocks, st);
  }
  // every other shape: the product, then the activation backward as the pass of its own (in place)
  const int rc = launch_plain<true, true>(p, st);
  if (rc != GTS_OK) return rc;
  return gts_gat_act_bwd_f32(gin, act_out, activation, gin, g_bias, workspace, workspace_bytes, m, k, stream);
}

extern "C" int64_t gts_linear_bwd_weight_workspace(int64_t m, int64_t n, int64_t k,
                                                   int32_t n_problems) {
  using namespace gts;
  if (m <= 0 || n <= 0 || k <= 0 || n_problems < 1 || n_problems > kMaxProblems) return 0;
  const int64_t splits = wgrad_plan(m, n, k, n_problems).splits;
  const int64_t slabs = n_problems * splits * (n * k + n);
  // the skinny path handles one problem at a time: [chunks][small+1][big] (+ [chunks][small] bias)
  const int64_t skinny = skinny_workspace_floats(m, n, k) + kSkinnyChunks * (n < k ? n : 0);
  return (slabs > skinny ? slabs : skinny) * static_cast<int64_t>(sizeof(float));
}

extern "C" int32_t gts_linear_bwd_weight_f32(const float* const* g, const float* const* a,
                                             float* const* gw, float* const* gb,
                                             int32_t n_problems, float* workspace,
                                             int64_t workspace_bytes, int64_t m, int64_t n,
                                             int64_t k, void* stream) {
  using namespace gts;
  if (!g || !a || !gw || !workspace) return GTS_ERR_NULL;
  if (n_problems < 1 || n_problems > kMaxProblems) return GTS_ERR_ARGKIND;
  if (m <= 0 || n <= 0 || k <= 0 || m >= (1LL << 31) || n >= (1 << 20) || k >= (1 << 20) ||
      !aligned4(n) || !aligned4(k))
    return GTS_ERR_SHAPE;
  if (workspace_bytes < gts_linear_bwd_weight_workspace(m, n, k, n_problems)) return GTS_ERR_SHAPE;
  bool any_bias = false;
  for (int q = 0; q < n_problems; ++q) {
    if (!g[q] || !a[q] || !gw[q]) return GTS_ERR_NULL;
    any_bias = any_bias || (gb && gb[q]);
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (skinny_workspace_floats(m, n, k) > 0) {   // tiny N or K: streaming kernel, problem by problem
    for (int q = 0; q < n_problems; ++q)
      skinny_wgrad(g[q], a[q], gw[q], gb ? gb[q] : nullptr, workspace, m, n, k, st);
    return launch_status();
  }
  const WgradPlan plan = wgrad_plan(m, n, k, n_problems);
  const int splits = plan.splits;
  const int tiles = static_cast<int>((m + kBK - 1) / kBK);
  GemmArgs p{};
  // C[n, k] = sum_m g[m, n] * act[m, k]: both operands reduction-strided
  for (int q = 0; q < n_problems; ++q) p.pa[q] = g[q], p.pb[q] = a[q];
  p.a[0] = p.a[1] = g[0], p.b[0] = p.b[1] = a[0];
  p.lda[0] = p.lda[1] = static_cast<int>(n), p.ldb[0] = p.ldb[1] = static_cast<int>(k);
  p.kseg[0] = static_cast<int>(m), p.kseg[1] = 0;
  p.ra = static_cast<int>(n), p.rb = static_cast<int>(k);
  p.c = workspace, p.ldc = static_cast<int>(k);
  p.colsum = any_bias ? workspace + static_cast<size_t>(n_problems) * splits * n * k : nullptr;
  for (int q = 0; q < n_problems && any_bias; ++q)
    if (gb[q]) p.colsum_mask |= 1u << q;
  p.n_problems = n_problems, p.n_splits = splits;
  p.tiles_per_split = (tiles + splits - 1) / splits;
  int rc = launch_wgrad(p, plan, st);
  if (rc != GTS_OK) return rc;
  // one reduction launch: weight slabs of every problem, then the bias slabs that were asked for
  ReduceArgs r{};
  r.splits = splits;
  int jobs = 0;
  for (int q = 0; q < n_problems; ++q, ++jobs) {
    r.slabs[jobs] = workspace + static_cast<size_t>(q) * splits * n * k;
    r.out[jobs] = gw[q], r.n4[jobs] = static_cast<int>(n * k / 4);
  }
  for (int q = 0; q < n_problems && any_bias; ++q) {
    if (!gb[q]) continue;
    r.slabs[jobs] = p.colsum + static_cast<size_t>(q) * splits * n;
    r.out[jobs] = gb[q], r.n4[jobs] = static_cast<int>(n / 4);
    ++jobs;
  }
  reduce_slabs_kernel<<<dim3(static_cast<unsigned>((n * k / 4 + 63) / 64), jobs), kBlock, 0, st>>>(r);
  return launch_status();
}
