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

}  // namespace
}  // namespace gts

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
