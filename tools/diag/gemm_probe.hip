// Diagnostic translation unit (never part of libgts_hip.so): the K11 kernels of
// gnn-tumor-seg_amd/csrc/gts_gemm.hip instantiated with a probe that records, per workgroup and
// phase (0 start, 1 operands staged, 2 main loop done, 3 tile stored), the constant 100 MHz
// s_memrealtime counter and the shader-clock s_memtime counter (their ratio = in-kernel clock).
// Built and driven by tools/diag/gemm_stamps.py.
#include "../../gnn-tumor-seg_amd/csrc/gts_gemm.hip"
#include "gemm_rejected_forms.inc"   // the kernel forms measured and rejected: built here, never shipped

namespace gts {
namespace {

__device__ unsigned long long* g_probe = nullptr;  // [workgroup][phase][2]

struct StampProbe {
  __device__ __forceinline__ static void mark(int phase) {
    if (phase == 3) __syncthreads();
    if (threadIdx.x == 0 && g_probe != nullptr) {
      const size_t wg = (static_cast<size_t>(blockIdx.z) * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      const size_t at = (wg * 4 + phase) * 2;
      g_probe[at] = __builtin_amdgcn_s_memrealtime();
      // phase 0 also records where the wave runs: HW_ID (wave / SIMD / CU / SE) | XCC_ID << 32
      g_probe[at + 1] = phase == 0 ? (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg(6164)) << 32) |
                                         __builtin_amdgcn_s_getreg(63492)
                                   : __builtin_amdgcn_s_memtime();
    }
  }
};

// per-WAVE stamps (lane 0 of every wave; no barrier): for the kernels whose waves run independently
struct WaveProbe {
  __device__ __forceinline__ static void mark(int phase) {
    if ((threadIdx.x & 63) == 0 && g_probe != nullptr) {
      const size_t wg = static_cast<size_t>(blockIdx.y) * gridDim.x + blockIdx.x;
      const size_t at = (((wg * (blockDim.x >> 6)) + (threadIdx.x >> 6)) * 4 + phase) * 2;
      g_probe[at] = __builtin_amdgcn_s_memrealtime();
      // phase 0 also records where the wave runs: HW_ID (wave / SIMD / CU / SE) | XCC_ID << 32
      g_probe[at + 1] = phase == 0 ? (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg(6164)) << 32) |
                                         __builtin_amdgcn_s_getreg(63492)
                                   : __builtin_amdgcn_s_memtime();
    }
  }
};

}  // namespace
}  // namespace gts

extern "C" int gts_probe_set_buffer(unsigned long long* buf) {
  return static_cast<int>(hipMemcpyToSymbol(HIP_SYMBOL(gts::g_probe), &buf, sizeof(buf)));
}

// forward GEMM of the library with the stamping kernels; variant as GTS_OPT_GEMM_TILE;
// a_hot != 0: every row tile reads the same (cache-hot) A rows (row stride 0)
extern "C" int gts_probe_linear_fwd(const float* a0, const float* w0, const float* a1, const float* w1,
                                    const float* bias, float* out, int64_t m, int64_t n, int64_t k0,
                                    int64_t k1, int32_t relu, int32_t variant, int32_t a_hot,
                                    void* stream) {
  using namespace gts;
  GemmArgs p{};
  p.a[0] = a0, p.b[0] = w0, p.lda[0] = static_cast<int>(k0), p.ldb[0] = static_cast<int>(k0);
  p.kseg[0] = static_cast<int>(k0);
  p.a[1] = a1 ? a1 : a0, p.b[1] = w1 ? w1 : w0;
  p.lda[1] = p.ldb[1] = static_cast<int>(k1), p.kseg[1] = a1 ? static_cast<int>(k1) : 0;
  p.ra = static_cast<int>(m), p.rb = static_cast<int>(n), p.c = out, p.ldc = static_cast<int>(n);
  p.bias = bias, p.relu = relu;
  p.tiles_per_split = (p.kseg[0] + kBK - 1) / kBK + (p.kseg[1] + kBK - 1) / kBK;
  if (a_hot & 1) p.lda[0] = p.lda[1] = 0;
  if (a_hot & 2) p.ldb[0] = p.ldb[1] = 0;   // every weight row of a tile is the same row: a fragment load touches one 64-byte piece
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (variant) {
    case 3: return launch_tiles<64, 256, 2, 4, true, true, false, StampProbe>(p, 1, 1, st);
    case 8: return launch_tiles<256, 256, 4, 4, true, true, true, StampProbe>(p, 1, 1, st);
    case 9: return launch_rows240<StampProbe>(p, st);
    case 10: return launch_panel_direct<3, 4, 1, StampProbe>(p, st);
    case 110: return launch_panel_direct<3, 4, 1, WaveProbe>(p, st);
    // the reference's batch size (35 000 rows): 144-row panels, fragments 1 / 2 / 3 reduction groups ahead
    case 1441: return launch_panel_direct<3, 4, 1, StampProbe, 144>(p, st);
    case 1442: return launch_panel_direct<3, 4, 2, StampProbe, 144>(p, st);
    case 1443: return launch_panel_direct<3, 4, 3, StampProbe, 144>(p, st);
    case 11: return launch_panel_direct<1, 4, 1, StampProbe>(p, st);
    case 12: return launch_panel_direct<1, 4, 2, StampProbe>(p, st);
    default: return launch_tiles<128, 256, 2, 4, true, true, false, StampProbe>(p, 1, 1, st);
  }
}

// the 19-problem weight-gradient launch (variant 4 = double-buffered LDS tile, 7 = LDS-DMA tiles) with stamps
extern "C" int gts_probe_wgrad(const float* const* g, const float* const* a, int32_t n_problems, float* workspace,
                               int64_t m, int64_t n, int64_t k, int32_t variant, int32_t splits, void* stream) {
  using namespace gts;
  GemmArgs p{};
  for (int q = 0; q < n_problems; ++q) p.pa[q] = g[q], p.pb[q] = a[q];
  p.a[0] = p.a[1] = g[0], p.b[0] = p.b[1] = a[0];
  p.lda[0] = p.lda[1] = static_cast<int>(n), p.ldb[0] = p.ldb[1] = static_cast<int>(k);
  p.kseg[0] = static_cast<int>(m), p.kseg[1] = 0;
  p.ra = static_cast<int>(n), p.rb = static_cast<int>(k);
  p.c = workspace, p.ldc = static_cast<int>(k);
  p.colsum = nullptr;
  p.n_problems = n_problems, p.n_splits = splits;
  const int tiles = static_cast<int>((m + kBK - 1) / kBK);
  p.tiles_per_split = (tiles + splits - 1) / splits;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (variant == 7 || variant == 71 || variant == 72) {
    GemmArgs q = p;
    q.tiles_n = (p.rb + 255) / 256;
    dim3 grid((p.ra + 255) / 256, q.tiles_n * n_problems, splits);
    if (variant == 7) wgrad_dma_kernel<StampProbe><<<grid, 1024, 0, st>>>(q);
    if (variant == 71) wgrad_dma_kernel<StampProbe, 1><<<grid, 1024, 0, st>>>(q);   // no LDS fragment reads
    if (variant == 72) wgrad_dma_kernel<StampProbe, 2><<<grid, 1024, 0, st>>>(q);   // no barrier, no DMA after the first two tiles
    return launch_status();
  }
  return launch_tiles<256, 256, 4, 4, false, false, true, StampProbe>(p, n_problems, splits, st);
}
