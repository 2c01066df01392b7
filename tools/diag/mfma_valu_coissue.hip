// Diagnostic (not part of the library): does the f32 vector ALU run beside the f32 matrix pipe?
// Each wave issues groups of v_mfma_f32_16x16x4_f32 (independent accumulators) with NF v_fma_f32 (VGPR x SGPR,
// independent chains) behind every MFMA; all CUs busy, 3 waves per SIMD as in the K11 panel kernel.  Prints
// the MFMA rate, the VALU rate and the in-kernel clock for NF = 0 .. 8, and the VALU-only rate.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/diag/mfma_valu_coissue.hip -o /tmp/coissue && /tmp/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int NM, int NF, bool MFMA>
__global__ __launch_bounds__(768, 3) void loop_kernel(const float* in, const float* __restrict__ sc, float* out,
                                                      unsigned long long* stamps, int iters) {
  const int lane = threadIdx.x;
  float a = in[lane], b = in[lane + 768];
  v4f acc[NM];
  for (int k = 0; k < NM; ++k) acc[k] = v4f{0.f, 0.f, 0.f, 0.f};
  constexpr int NV = NF > 0 ? 16 : 1;
  float vacc[NV];
  for (int c = 0; c < NV; ++c) vacc[c] = 0.f;
  float va = in[lane + 1536];
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    float s[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) s[c] = sc[(i & 15) * 16 + c];   // uniform: scalar loads
    int f = 0;
#pragma unroll
    for (int k = 0; k < NM; ++k) {
      if (MFMA) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NF; ++j, ++f) vacc[f % NV] = __builtin_fmaf(va, s[f % 16], vacc[f % NV]);
    }
#pragma unroll
    for (int k = 0; k < NM; ++k) {
      if (MFMA) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (NF > 0) __builtin_amdgcn_sched_group_barrier(0x002, NF, 0);
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float sum = 0.f;
  for (int k = 0; k < NM; ++k)
    for (int r = 0; r < 4; ++r) sum += acc[k][r];
  for (int c = 0; c < NV; ++c) sum += vacc[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = c1 - c0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}

template <int NM, int NF, bool MFMA>
void run(int iters) {
  const int blocks = 256;
  float *in, *out, *sc;
  unsigned long long* stamps;
  hipMalloc(&in, 2304 * 4);
  hipMalloc(&sc, 256 * 4);
  hipMalloc(&out, blocks * 768 * 4);
  hipMalloc(&stamps, blocks * 16);
  std::vector<float> h(2304), hs(256);
  for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
  for (auto& v : hs) v = (float)rand() / RAND_MAX - 0.5f;
  hipMemcpy(in, h.data(), 2304 * 4, hipMemcpyHostToDevice);
  hipMemcpy(sc, hs.data(), 256 * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) loop_kernel<NM, NF, MFMA><<<blocks, 768>>>(in, sc, out, stamps, iters);
  hipEventRecord(e0);
  const int reps = 4;
  for (int rep = 0; rep < reps; ++rep) loop_kernel<NM, NF, MFMA><<<blocks, 768>>>(in, sc, out, stamps, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)reps * blocks * 12 * iters;
  const double mf = MFMA ? waves * NM * 2048.0 : 0.0, vf = waves * NM * NF * 128.0;
  std::vector<unsigned long long> st(2 * blocks);
  hipMemcpy(st.data(), stamps, blocks * 16, hipMemcpyDeviceToHost);
  std::vector<double> ghz;
  for (int b = 0; b < blocks; ++b) ghz.push_back((double)st[2 * b] / (double)st[2 * b + 1] * 0.1);
  std::sort(ghz.begin(), ghz.end());
  const double cyc = (double)st[0] / iters / NM;
  printf("mfma=%d fma/mfma=%d: MFMA %.1f TF + VALU %.1f TF = %.1f TF; %.1f cycles per MFMA slot per wave; clock %.3f GHz\n",
         (int)MFMA, NF, mf / (ms * 1e-3) / 1e12, vf / (ms * 1e-3) / 1e12, (mf + vf) / (ms * 1e-3) / 1e12, cyc,
         ghz[ghz.size() / 2]);
  hipFree(in); hipFree(out); hipFree(stamps); hipFree(sc);
}

int main() {
  run<16, 0, true>(4000);
  run<16, 1, true>(4000);
  run<16, 2, true>(4000);
  run<16, 3, true>(4000);
  run<16, 4, true>(4000);
  run<16, 6, true>(4000);
  run<16, 8, true>(4000);
  run<16, 12, true>(4000);
  run<16, 4, false>(4000);
  run<16, 8, false>(4000);
  return 0;
}
