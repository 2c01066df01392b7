// Class-weighted cross-entropy over node logits (model/gnn_model.py:30,42 of the reference:
// torch.nn.CrossEntropyLoss(weight=class_weights)(logits, labels)), forward + the unscaled
// gradient in one pass.  [N, C] with C <= 32 (4 classes here): one lane per node, tiny rows,
// HBM-bound and small; the point is ONE launch with a deterministic two-level reduction
// instead of a log_softmax + nll_loss chain whose reduction runs in a single workgroup.
//
//   lse_i  = log sum_c exp(x_ic - max_c x_ic) + max_c x_ic
//   num    = sum_i w[y_i] (lse_i - x_i,y_i)        den = sum_i w[y_i]
//   loss   = num / den
//   gu_ic  = w[y_i] (softmax_ic - [c == y_i])      (d loss / d x_ic = gu_ic / den)
#include "gts_common.h"

namespace gts {
namespace {

constexpr int kMaxClasses = 32;
constexpr int kRowsPerThread = 4;

// NC > 0: class count known at compile time (row lives in registers); NC == 0: runtime count
template <int NC>
__global__ __launch_bounds__(kBlock) void weighted_ce_kernel(
    const float* __restrict__ logits, const int64_t* __restrict__ labels,
    const float* __restrict__ class_w, float* __restrict__ grad_unscaled,
    float* __restrict__ partials, int64_t n, int n_classes_rt) {
  const int n_classes = NC ? NC : n_classes_rt;
  __shared__ float red[2][kWavesPerBlock];
  float num = 0.f, den = 0.f;
  const int64_t base = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) * kRowsPerThread;
  for (int r = 0; r < kRowsPerThread; ++r) {
    const int64_t i = base + r;
    if (i >= n) break;
    const float* x = logits + i * n_classes;
    float v[NC ? NC : kMaxClasses];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < n_classes; ++c) v[c] = x[c], mx = fmaxf(mx, v[c]);
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < n_classes; ++c) v[c] = expf(v[c] - mx), sum += v[c];
    // a label outside [0, C): torch's ignore_index (-100) contributes nothing; anything else is an
    // error that torch reports with a device assert — here it poisons the loss with NaN instead of
    // reading class_w / the logits row out of bounds
    const long long label = labels[i];
    const bool valid = label >= 0 && label < n_classes;
    const int y = valid ? static_cast<int>(label) : 0;
    const float w = !valid ? 0.0f : (class_w != nullptr ? class_w[y] : 1.0f);
    num += valid ? w * (logf(sum) + mx - x[y]) : (label == -100 ? 0.0f : NAN);
    den += w;
    if (grad_unscaled != nullptr) {
      float* g = grad_unscaled + i * n_classes;
#pragma unroll
      for (int c = 0; c < n_classes; ++c) g[c] = w * (v[c] / sum - (c == y ? 1.0f : 0.0f));
    }
  }
  // wave butterfly, then the four wave totals in wave order: a fixed association
  for (int m = kWave / 2; m >= 1; m >>= 1) {
    num += __shfl_xor(num, m, kWave);
    den += __shfl_xor(den, m, kWave);
  }
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  if (lane == 0) red[0][wave] = num, red[1][wave] = den;
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[2 * blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    partials[2 * blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

// one workgroup adds the per-block partials in block order -> out = {num, den, num / den}
__global__ __launch_bounds__(kBlock) void weighted_ce_finish_kernel(const float* __restrict__ partials,
                                                                   float* __restrict__ out,
                                                                   int n_blocks) {
  __shared__ float red[2][kBlock];
  float num = 0.f, den = 0.f;
  for (int b = threadIdx.x; b < n_blocks; b += kBlock) num += partials[2 * b], den += partials[2 * b + 1];
  red[0][threadIdx.x] = num, red[1][threadIdx.x] = den;
  __syncthreads();
  for (int s = kBlock / 2; s >= 1; s >>= 1) {
    if (threadIdx.x < s) {
      red[0][threadIdx.x] += red[0][threadIdx.x + s];
      red[1][threadIdx.x] += red[1][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0][0], out[1] = red[1][0], out[2] = red[0][0] / red[1][0];
}

}  // namespace
}  // namespace gts

extern "C" int64_t gts_weighted_ce_workspace(int64_t n) {
  using namespace gts;
  if (n <= 0) return 0;
  const int64_t blocks = (n + kBlock * kRowsPerThread - 1) / (kBlock * kRowsPerThread);
  return blocks * 2 * static_cast<int64_t>(sizeof(float));
}

extern "C" int32_t gts_weighted_ce_f32(const float* logits, const int64_t* labels,
                                       const float* class_w, float* grad_unscaled,
                                       float* workspace, int64_t workspace_bytes, float* out3,
                                       int64_t n, int64_t n_classes, void* stream) {
  using namespace gts;
  if (!logits || !labels || !workspace || !out3) return GTS_ERR_NULL;
  if (n <= 0 || n >= (1LL << 40) || n_classes < 1 || n_classes > kMaxClasses) return GTS_ERR_SHAPE;
  if (workspace_bytes < gts_weighted_ce_workspace(n)) return GTS_ERR_SHAPE;
  const int64_t blocks = (n + kBlock * kRowsPerThread - 1) / (kBlock * kRowsPerThread);
  if (blocks >= (1LL << 31)) return GTS_ERR_SHAPE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n_classes == 4)
    weighted_ce_kernel<4><<<static_cast<unsigned>(blocks), kBlock, 0, st>>>(
        logits, labels, class_w, grad_unscaled, workspace, n, 4);
  else
    weighted_ce_kernel<0><<<static_cast<unsigned>(blocks), kBlock, 0, st>>>(
        logits, labels, class_w, grad_unscaled, workspace, n, static_cast<int>(n_classes));
  weighted_ce_finish_kernel<<<1, kBlock, 0, st>>>(workspace, out3, static_cast<int>(blocks));
  return launch_status();
}
