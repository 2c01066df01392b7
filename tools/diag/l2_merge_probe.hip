// Does an XCD's L2 merge misses to a line that is already in flight?  (tools/diag; run under rocprofv3 --pmc FETCH_SIZE)
// 256 workgroups, one per CU; the 32 workgroups of an XCD (blockIdx % 8) read the SAME 64 MiB region chunk by chunk (64 KiB per
// step, 16 B per lane) — mode 0: all of them chunk k at step k (every line asked for by 32 CUs at nearly the same moment);
// mode 1: workgroup j reads chunk k + j (simultaneous readers touch different chunks; every chunk is read by all 32 within
// a window of 2 MiB, which the 4 MiB L2 holds);  mode 2: every workgroup its own region (no sharing: the compulsory figure x 32).
// FETCH_SIZE x 2 against 8 x 64 MiB = 512 MiB tells: ~1 x = merged / served by the L2, ~32 x = every request went to the fabric.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void probe(const float4* __restrict__ buf, float* __restrict__ sink, int mode, int chunks,
                                             size_t region_f4) {
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const float4* base = buf + (mode == 2 ? (size_t)blockIdx.x : (size_t)xcd) * region_f4;
  float4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < chunks; ++k) {
    const int c = mode == 1 ? (k + j) % chunks : k;
    const float4* p = base + (size_t)c * 4096 + threadIdx.x;   // 64 KiB = 4096 float4
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float4 v = p[q * 256];
      acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.25f) sink[blockIdx.x] = acc.x;
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const size_t region = mode == 2 ? (2u << 20) : (64u << 20);   // bytes per region
  const int regions = mode == 2 ? 256 : 8;
  const int chunks = (int)(region / 65536);
  float4* buf;
  float* sink;
  hipMalloc(&buf, region * regions);
  hipMalloc(&sink, 4096);
  hipMemset(buf, 0, region * regions);
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    probe<<<256, 256>>>(buf, sink, mode, chunks, region / 16);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("mode %d: %.1f us, distinct bytes %.1f MiB, requested %.1f MiB\n", mode, ms * 1000.f, region * regions / 1048576.0,
           256.0 * region / 1048576.0);
  }
  return 0;
}
