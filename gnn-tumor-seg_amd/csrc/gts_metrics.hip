// K15: joint label histogram ("confusion counts") of two int16 label arrays — the device half of
// GNN.evaluate's Dice metrics (model/gnn_model.py:76-87 -> model/evaluation.py:24-46,64-79,98-106
// in the reference count label coincidences over every node / voxel with numpy on the host).
//
// HBM-bound streaming reduction: 2 + 2 bytes per element in, 25 integers out.  Class index of a
// label v is v for 0..3 and 4 ("anything else") otherwise, so the 5x5 table carries exactly what
// the reference's region masks (!= 0, isin [2,3], == 3) can distinguish.  Every thread keeps a
// private column of the table in LDS (bin-major, so a wave's updates hit 64 distinct addresses),
// the columns are summed per workgroup and added to the caller's int64 table with integer
// atomics — the result does not depend on scheduling.
#include "gts_common.h"

namespace gts {
namespace {

constexpr int kClasses = 5;
constexpr int kBins = kClasses * kClasses;

__device__ __forceinline__ int class_of(int v) { return static_cast<unsigned>(v) < 4u ? v : 4; }

__global__ __launch_bounds__(kBlock) void label_confusion_kernel(
    const int16_t* __restrict__ pred, const int16_t* __restrict__ truth,
    unsigned long long* __restrict__ counts, int64_t n, int vector_ok) {
  __shared__ unsigned hist[kBins][kBlock];
  const int tid = threadIdx.x;
#pragma unroll
  for (int b = 0; b < kBins; ++b) hist[b][tid] = 0;  // own column only: no barrier needed yet

  const int64_t n_groups = vector_ok ? n / 8 : 0;  // 16-byte groups of 8 labels
  for (int64_t g = static_cast<int64_t>(blockIdx.x) * kBlock + tid; g < n_groups;
       g += static_cast<int64_t>(gridDim.x) * kBlock) {
    const uint4 p = reinterpret_cast<const uint4*>(pred)[g];
    const uint4 t = reinterpret_cast<const uint4*>(truth)[g];
    const unsigned pw[4] = {p.x, p.y, p.z, p.w}, tw[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int p0 = static_cast<int16_t>(pw[w] & 0xffffu), p1 = static_cast<int16_t>(pw[w] >> 16);
      const int t0 = static_cast<int16_t>(tw[w] & 0xffffu), t1 = static_cast<int16_t>(tw[w] >> 16);
      hist[class_of(p0) * kClasses + class_of(t0)][tid] += 1;
      hist[class_of(p1) * kClasses + class_of(t1)][tid] += 1;
    }
  }
  // what the 16-byte groups do not cover (everything when a pointer is not 16-byte aligned)
  for (int64_t i = n_groups * 8 + static_cast<int64_t>(blockIdx.x) * kBlock + tid; i < n;
       i += static_cast<int64_t>(gridDim.x) * kBlock)
    hist[class_of(pred[i]) * kClasses + class_of(truth[i])][tid] += 1;
  __syncthreads();

  // wave w sums bins w, w+4, ...: lane l adds columns l, l+64, l+128, l+192, then a butterfly
  const int lane = tid & (kWave - 1), wave = tid / kWave;
  for (int b = wave; b < kBins; b += kWavesPerBlock) {
    unsigned long long s = 0;
#pragma unroll
    for (int c = lane; c < kBlock; c += kWave) s += hist[b][c];
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, kWave);
    if (lane == 0 && s != 0) atomicAdd(counts + b, s);
  }
}

}  // namespace
}  // namespace gts

extern "C" int32_t gts_label_confusion_i16(const int16_t* pred, const int16_t* truth,
                                           int64_t* counts, int64_t n, void* stream) {
  using namespace gts;
  if (!counts || (n > 0 && (!pred || !truth))) return GTS_ERR_NULL;
  if (n < 0) return GTS_ERR_SHAPE;
  if (n == 0) return GTS_OK;
  const int vector_ok = ((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(truth)) & 15) == 0;
  const int64_t per_block = static_cast<int64_t>(kBlock) * 8 * 4;  // >= 4 groups per thread
  int64_t blocks = (n + per_block - 1) / per_block;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  label_confusion_kernel<<<static_cast<unsigned>(blocks), kBlock, 0, static_cast<hipStream_t>(stream)>>>(
      pred, truth, reinterpret_cast<unsigned long long*>(counts), n, vector_ok);
  return launch_status();
}
