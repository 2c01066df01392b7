// AdamW over ONE flat fp32 parameter buffer (torch.optim.AdamW(net.parameters(), lr, weight_decay)
// at model/gnn_model.py:28 of the reference, its update rule unchanged):
//   p   <- p * (1 - lr * wd)
//   m   <- m + (1 - beta1) * (g - m)
//   v   <- beta2 * v + (1 - beta2) * g * g
//   p   <- p - (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
// HBM-bound elementwise pass: 16 B in per element (p, g, m, v), 12 B out.  The network has
// 1.25 M parameters in 32 tensors; a multi-tensor launch that hands each workgroup a 64 Ki-element
// chunk keeps ~50 of 256 CUs busy (43 us per launch, two launches), one flat pass takes a few us.
#include "gts_common.h"

namespace gts {
namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

struct AdamWConsts {
  float decay;      // 1 - lr * weight_decay
  float w1;         // 1 - beta1
  float beta2, w2;  // beta2, 1 - beta2
  float step_size;  // lr / bias_correction1
  float inv_bc2_sqrt_den;  // sqrt(bias_correction2): v_hat denominator divisor
  float eps;
};

__device__ __forceinline__ void adamw_one(float& p, float g, float& m, float& v, const AdamWConsts& c) {
  p = p * c.decay;
  m = m + c.w1 * (g - m);
  v = c.beta2 * v + c.w2 * (g * g);
  const float denom = sqrtf(v) / c.inv_bc2_sqrt_den + c.eps;
  p = p - c.step_size * (m / denom);
}

__global__ __launch_bounds__(kBlock) void adamw_kernel(float* __restrict__ param,
                                                      const float* __restrict__ grad,
                                                      float* __restrict__ exp_avg,
                                                      float* __restrict__ exp_avg_sq, int64_t n,
                                                      AdamWConsts c, int vector_ok) {
  const int64_t n4 = vector_ok ? n / 4 : 0;
  for (int64_t q = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; q < n4;
       q += static_cast<int64_t>(gridDim.x) * kBlock) {
    v4f p = reinterpret_cast<v4f*>(param)[q];
    const v4f g = reinterpret_cast<const v4f*>(grad)[q];
    v4f m = reinterpret_cast<v4f*>(exp_avg)[q];
    v4f v = reinterpret_cast<v4f*>(exp_avg_sq)[q];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float pe = p[e], me = m[e], ve = v[e];
      adamw_one(pe, g[e], me, ve, c);
      p[e] = pe, m[e] = me, v[e] = ve;
    }
    reinterpret_cast<v4f*>(param)[q] = p;
    reinterpret_cast<v4f*>(exp_avg)[q] = m;
    reinterpret_cast<v4f*>(exp_avg_sq)[q] = v;
  }
  for (int64_t i = 4 * n4 + static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kBlock)
    adamw_one(param[i], grad[i], exp_avg[i], exp_avg_sq[i], c);
}

}  // namespace
}  // namespace gts

extern "C" int32_t gts_adamw_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                                 int64_t n, double lr, double beta1, double beta2, double eps,
                                 double weight_decay, int64_t step, void* stream) {
  using namespace gts;
  if (n < 0 || step < 1 || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0))
    return GTS_ERR_SHAPE;
  if (n == 0) return GTS_OK;
  if (!param || !grad || !exp_avg || !exp_avg_sq) return GTS_ERR_NULL;
  // scalar constants in double, rounded once (torch forms them as Python floats)
  const double bc1 = 1.0 - pow(beta1, static_cast<double>(step));
  const double bc2 = 1.0 - pow(beta2, static_cast<double>(step));
  AdamWConsts c;
  c.decay = static_cast<float>(1.0 - lr * weight_decay);
  c.w1 = static_cast<float>(1.0 - beta1);
  c.beta2 = static_cast<float>(beta2);
  c.w2 = static_cast<float>(1.0 - beta2);
  c.step_size = static_cast<float>(lr / bc1);
  c.inv_bc2_sqrt_den = static_cast<float>(sqrt(bc2));
  c.eps = static_cast<float>(eps);
  const int vector_ok = ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) |
                          reinterpret_cast<uintptr_t>(exp_avg) | reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15) == 0;
  int64_t blocks = (n / 4 + kBlock - 1) / kBlock;
  blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
  adamw_kernel<<<static_cast<unsigned>(blocks), kBlock, 0, static_cast<hipStream_t>(stream)>>>(
      param, grad, exp_avg, exp_avg_sq, n, c, vector_ok);
  return launch_status();
}
