// Batched transpose of small weight matrices (one launch for a whole layer stack).
//
// The input-gradient GEMM gin = g . W reduces over W's ROW index, i.e. it reads W with the
// reduction index strided.  Handing the GEMM kernel W^T instead makes both of its operands
// reduction-contiguous — the forward form, whose LDS fragment reads are one ds_read_b128 per four
// MFMAs instead of four ds_read_b32 — at the price of copying <= 21 matrices of 256 KB once per
// backward pass (5.4 MB, a few microseconds).  HBM-trivial; classic 32x32 LDS tile with a padded
// row so that the transposed read is conflict-free.
#include <cstddef>

#include "gts_common.h"

namespace gts {
namespace {

constexpr int kMaxMats = 32;
constexpr int kTile = 32;

struct TransposeArgs {
  const float* src[kMaxMats];  // [rows, cols] each
  float* dst[kMaxMats];        // [cols, rows] each
  int rows, cols;
};

__global__ __launch_bounds__(kBlock) void transpose_batch_kernel(const TransposeArgs p) {
  __shared__ float tile[kTile][kTile + 1];
  const float* src = kernarg_entry<const float*>(offsetof(TransposeArgs, src), blockIdx.z);
  float* dst = kernarg_entry<float*>(offsetof(TransposeArgs, dst), blockIdx.z);
  const int tx = threadIdx.x & (kTile - 1), ty = threadIdx.x / kTile;  // 32 x 8 threads
  const int c0 = blockIdx.x * kTile, r0 = blockIdx.y * kTile;
#pragma unroll
  for (int i = ty; i < kTile; i += kBlock / kTile) {
    const int r = r0 + i, c = c0 + tx;
    if (r < p.rows && c < p.cols) tile[i][tx] = src[static_cast<size_t>(r) * p.cols + c];
  }
  __syncthreads();
#pragma unroll
  for (int i = ty; i < kTile; i += kBlock / kTile) {
    const int c = c0 + i, r = r0 + tx;  // output row = input column
    if (r < p.rows && c < p.cols) dst[static_cast<size_t>(c) * p.rows + r] = tile[tx][i];
  }
}

// Weights in FRAGMENT ORDER for the panel GEMMs (csrc/gts_gemm.hip): B [n, k] (the operand whose rows are output columns
// and whose columns are the reduction) is cut into 16-row tiles and reduction groups of 16; the 16 x 16 block of tile T,
// group g is stored as the 64 x 4 floats the 64 lanes of a wave feed to four v_mfma_f32_16x16x4_f32:
//     packed[((T * G + g) * 64 + lane) * 4 + e] = B[16 T + (lane & 15)][16 g + 4 (lane >> 4) + e]      (0 past the edges)
// with G = ceil(k / 16).  A wave's fragment load is then ONE run of 1 KiB (sixteen 64-byte accesses of the vector L1)
// where rows as torch stores them cost one 16-byte access per lane, 1 KiB apart (64 accesses): the panel kernels are bound
// by that access rate (profiles/r04/panel144_l1_bound.log).  transposed: B = src^T; plain_t (optional): src^T row-major too.
struct PackArgs {
  const float* src[kMaxMats];  // [rows, cols] each
  float* dst[kMaxMats];        // fragment order
  float* plain_t[kMaxMats];    // [cols, rows] each, or null
  int rows, cols, transposed;
};

__global__ __launch_bounds__(kBlock) void pack_weights_kernel(const PackArgs p) {
  const float* src = kernarg_entry<const float*>(offsetof(PackArgs, src), blockIdx.y);
  float* dst = kernarg_entry<float*>(offsetof(PackArgs, dst), blockIdx.y);
  float* plain_t = kernarg_entry<float*>(offsetof(PackArgs, plain_t), blockIdx.y);
  const int n = p.transposed ? p.cols : p.rows, k = p.transposed ? p.rows : p.cols;   // B is [n, k]
  const int groups = (k + 15) >> 4, tiles = (n + 15) >> 4;
  const int idx = blockIdx.x * kBlock + threadIdx.x;      // one float4 of the packed buffer
  if (idx >= tiles * groups * 64) return;
  const int lane = idx & 63, block = idx >> 6;
  const int g = block % groups, t = block / groups;
  const int row = 16 * t + (lane & 15), k0 = 16 * g + 4 * (lane >> 4);
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int kk = k0 + e;
    const bool ok = row < n && kk < k;
    v[e] = !ok ? 0.f : p.transposed ? src[static_cast<size_t>(kk) * p.cols + row] : src[static_cast<size_t>(row) * p.cols + kk];
    if (ok && p.transposed && plain_t != nullptr) plain_t[static_cast<size_t>(row) * p.rows + kk] = v[e];
  }
  *reinterpret_cast<float4*>(dst + static_cast<size_t>(idx) * 4) = make_float4(v[0], v[1], v[2], v[3]);
}

}  // namespace
}  // namespace gts

extern "C" int64_t gts_packed_weight_floats(int64_t n, int64_t k) {
  if (n <= 0 || k <= 0 || n >= (1 << 20) || k >= (1 << 20)) return -1;
  return ((n + 15) / 16) * ((k + 15) / 16) * 256;
}

extern "C" int32_t gts_pack_weights_f32(const float* const* src, float* const* dst, float* const* plain_t, int32_t n_mats,
                                        int64_t rows, int64_t cols, int32_t transposed, void* stream) {
  using namespace gts;
  if (!src || !dst) return GTS_ERR_NULL;
  if (n_mats < 0 || rows <= 0 || cols <= 0 || rows >= (1 << 20) || cols >= (1 << 20)) return GTS_ERR_SHAPE;
  if (plain_t != nullptr && !transposed) return GTS_ERR_ARGKIND;
  for (int q = 0; q < n_mats; ++q)
    if (!src[q] || !dst[q]) return GTS_ERR_NULL;
  const int64_t total4 = gts_packed_weight_floats(transposed ? cols : rows, transposed ? rows : cols) / 4;
  if (total4 >= (1LL << 31)) return GTS_ERR_SHAPE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int first = 0; first < n_mats; first += kMaxMats) {
    const int count = n_mats - first < kMaxMats ? n_mats - first : kMaxMats;
    PackArgs p{};
    for (int q = 0; q < count; ++q)
      p.src[q] = src[first + q], p.dst[q] = dst[first + q], p.plain_t[q] = plain_t ? plain_t[first + q] : nullptr;
    p.rows = static_cast<int>(rows), p.cols = static_cast<int>(cols), p.transposed = transposed ? 1 : 0;
    dim3 grid(static_cast<unsigned>((total4 + kBlock - 1) / kBlock), static_cast<unsigned>(count), 1);
    pack_weights_kernel<<<grid, kBlock, 0, st>>>(p);
  }
  return launch_status();
}

extern "C" int32_t gts_transpose_batch_f32(const float* const* src, float* const* dst,
                                           int32_t n_mats, int64_t rows, int64_t cols,
                                           void* stream) {
  using namespace gts;
  if (!src || !dst) return GTS_ERR_NULL;
  if (n_mats < 0 || rows <= 0 || cols <= 0 || rows >= (1 << 20) || cols >= (1 << 20)) return GTS_ERR_SHAPE;
  for (int q = 0; q < n_mats; ++q)
    if (!src[q] || !dst[q]) return GTS_ERR_NULL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int first = 0; first < n_mats; first += kMaxMats) {
    const int count = n_mats - first < kMaxMats ? n_mats - first : kMaxMats;
    TransposeArgs p{};
    for (int q = 0; q < count; ++q) p.src[q] = src[first + q], p.dst[q] = dst[first + q];
    p.rows = static_cast<int>(rows), p.cols = static_cast<int>(cols);
    dim3 grid(static_cast<unsigned>((cols + kTile - 1) / kTile), static_cast<unsigned>((rows + kTile - 1) / kTile),
              static_cast<unsigned>(count));
    transpose_batch_kernel<<<grid, kBlock, 0, st>>>(p);
  }
  return launch_status();
}
