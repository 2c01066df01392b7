// K15: joint label histogram ("confusion counts") of two int16 label arrays — the device half of
// GNN.evaluate's Dice metrics (model/gnn_model.py:76-87 -> model/evaluation.py:24-46,64-79,98-106
// in the reference count label coincidences over every node / voxel with numpy on the host).
//
// HBM-bound streaming reduction: 2 + 2 bytes per element in, 25 integers out.  Class index of a
// label v is v for 0..3 and 4 ("anything else") otherwise, so the 5x5 table carries exactly what
// the reference's region masks (!= 0, isin [2,3], == 3) can distinguish.  Every thread keeps a
// private column of the table in LDS (bin-major, so a wave's updates hit 64 distinct addresses),
// the columns are summed per workgroup into one row of caller-owned scratch, and a one-workgroup
// second kernel adds the rows to the caller's int64 table (thousands of same-address atomics
// from the first kernel would cost more than the whole streaming pass).  Integer sums: exact.
#include "gts_common.h"

namespace gts {
namespace {

constexpr int kClasses = 5;
constexpr int kBins = kClasses * kClasses;

__device__ __forceinline__ int class_of(int v) { return static_cast<unsigned>(v) < 4u ? v : 4; }

__global__ __launch_bounds__(kBlock) void label_confusion_kernel(
    const int16_t* __restrict__ pred, const int16_t* __restrict__ truth,
    unsigned long long* __restrict__ partial, int64_t n, int vector_ok) {
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
    if (lane == 0) partial[static_cast<size_t>(blockIdx.x) * kBins + b] = s;
  }
}

// counts[b] += sum over workgroups of partial[g][b]; one workgroup, thread t owns bin t % 32 and
// every 8th row, then the 8 row groups are added through LDS.
__global__ __launch_bounds__(kBlock) void label_confusion_finish_kernel(
    const unsigned long long* __restrict__ partial, long long* __restrict__ counts, int n_groups) {
  __shared__ unsigned long long part[8][32];
  const int b = threadIdx.x & 31, slice = threadIdx.x >> 5;
  unsigned long long s = 0;
  if (b < kBins) {
#pragma unroll 16   // independent loads, issued back to back
    for (int g = slice; g < n_groups; g += 8) s += partial[static_cast<size_t>(g) * kBins + b];
  }
  part[slice][b] = s;
  __syncthreads();
  if (slice == 0 && b < kBins) {
#pragma unroll
    for (int k = 1; k < 8; ++k) s += part[k][b];
    counts[b] += static_cast<long long>(s);
  }
}

constexpr int kMaxGroups = 512;  // two workgroups per CU

inline int confusion_groups(int64_t n) {
  const int64_t per_block = static_cast<int64_t>(kBlock) * 8 * 4;  // >= 4 16-byte groups per thread
  const int64_t blocks = (n + per_block - 1) / per_block;
  return static_cast<int>(blocks < 1 ? 1 : (blocks > kMaxGroups ? kMaxGroups : blocks));
}

}  // namespace
}  // namespace gts

extern "C" int64_t gts_label_confusion_workspace(int64_t n) {
  using namespace gts;
  return n <= 0 ? 0 : static_cast<int64_t>(confusion_groups(n)) * kBins * sizeof(unsigned long long);
}

extern "C" int32_t gts_label_confusion_i16(const int16_t* pred, const int16_t* truth,
                                           int64_t* counts, void* workspace,
                                           int64_t workspace_bytes, int64_t n, void* stream) {
  using namespace gts;
  if (!counts || (n > 0 && (!pred || !truth || !workspace))) return GTS_ERR_NULL;
  if (n < 0 || workspace_bytes < gts_label_confusion_workspace(n)) return GTS_ERR_SHAPE;
  if (n == 0) return GTS_OK;
  const int vector_ok = ((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(truth)) & 15) == 0;
  const int groups = confusion_groups(n);
  hipStream_t st = static_cast<hipStream_t>(stream);
  unsigned long long* partial = static_cast<unsigned long long*>(workspace);
  label_confusion_kernel<<<groups, kBlock, 0, st>>>(pred, truth, partial, n, vector_ok);
  label_confusion_finish_kernel<<<1, kBlock, 0, st>>>(partial, reinterpret_cast<long long*>(counts), groups);
  return launch_status();
}
