// Diagnostic (not part of the library): sustained rate and in-kernel clock of a bare
// v_mfma_f32_32x32x2_f32 loop on random operands, all CUs busy — the practical ceiling the
// K11 GEMMs can be compared with.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/diag/mfma_f32_ceiling.hip -o /tmp/mfma_ceiling && /tmp/mfma_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float v16f __attribute__((ext_vector_type(16)));

template <int ACCS>
__global__ __launch_bounds__(256) void mfma_loop(const float* in, float* out, unsigned long long* stamps, int iters) {
  const int lane = threadIdx.x;
  float a = in[lane], b = in[lane + 256];
  v16f acc[ACCS];
  for (int k = 0; k < ACCS; ++k)
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < ACCS; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int k = 0; k < ACCS; ++k)
    for (int r = 0; r < 16; ++r) s += acc[k][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = c1 - c0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}

template <int ACCS>
void run(int blocks_per_cu, int iters) {
  const int blocks = 256 * blocks_per_cu;
  float *in, *out;
  unsigned long long* stamps;
  hipMalloc(&in, 512 * 4);
  hipMalloc(&out, blocks * 256 * 4);
  hipMalloc(&stamps, blocks * 16);
  std::vector<float> h(512);
  for (int i = 0; i < 512; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
  hipMemcpy(in, h.data(), 512 * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) mfma_loop<ACCS><<<blocks, 256>>>(in, out, stamps, iters);
  hipEventRecord(e0);
  const int reps = 5;
  for (int rep = 0; rep < reps; ++rep) mfma_loop<ACCS><<<blocks, 256>>>(in, out, stamps, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)reps * blocks * 4 /*waves*/ * iters * ACCS * 4096.0;
  std::vector<unsigned long long> st(2 * blocks);
  hipMemcpy(st.data(), stamps, blocks * 16, hipMemcpyDeviceToHost);
  std::vector<double> ghz;
  for (int b = 0; b < blocks; ++b) ghz.push_back((double)st[2 * b] / (double)st[2 * b + 1] * 0.1);
  std::sort(ghz.begin(), ghz.end());
  printf("accs=%d blocks/CU=%d: %.1f TF/s, in-kernel clock median %.3f GHz (min %.3f max %.3f)\n", ACCS,
         blocks_per_cu, flops / (ms * 1e-3) / 1e12, ghz[ghz.size() / 2], ghz.front(), ghz.back());
  hipFree(in); hipFree(out); hipFree(stamps);
}

int main() {
  run<4>(1, 20000);
  run<4>(2, 20000);
  run<8>(1, 10000);
  run<8>(2, 10000);
  return 0;
}
