// GATConv backward: the column reductions over the node axis, fused with the elementwise work
// that produces their inputs (HBM-bound, one pass over each [N, H*D] tensor):
//   gat_act_bwd      g_pre = gout * act'(out)  (ELU or ReLU through its output)  +  bias grad = colsum(g_pre)
//                    (also the ReLU / bias backward of the SAGEConv mean / gcn layers)
//   gat_param_grad   g_attn_l[h,:] = sum_n gel[n,h] ft[n,h,:],  g_attn_r likewise with ger
// Both walk row chunks with one 16-byte column group per thread (256 threads x 16 B = one 4 KiB
// row of the 1024-wide C3 layers per iteration), keep per-chunk partial sums in registers, write
// them to caller-owned scratch and add the chunks in a fixed order (bitwise reproducible).
#include "gts_common.h"

namespace gts {
namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int kMaxChunks = 512;  // two row chunks per CU

inline int64_t rows_per_chunk(int64_t n) { return (n + kMaxChunks - 1) / kMaxChunks; }
inline int n_chunks(int64_t n) {
  const int64_t rpc = rows_per_chunk(n);
  return static_cast<int>((n + rpc - 1) / rpc);
}

__global__ __launch_bounds__(kBlock) void gat_act_bwd_kernel(
    const float* gout, const float* __restrict__ out, int act,
    float* g_pre /* may be gout itself: every element is read, then written, by one thread */, float* __restrict__ partial, int64_t n, int cols, int64_t rpc) {
  const int64_t row0 = blockIdx.x * rpc, row1 = min(n, row0 + rpc);
  const int cols4 = cols >> 2;
  for (int q = threadIdx.x; q < cols4; q += kBlock) {
    v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int64_t row = row0; row < row1; ++row) {
      const size_t off = static_cast<size_t>(row) * cols + 4 * q;
      v4f g = *reinterpret_cast<const v4f*>(gout + off);
      if (act != 0) {  // ELU: d/dx = 1 for x > 0, exp(x) = out + 1 otherwise; ReLU (2): 1 for out > 0, else 0
        const v4f o = *reinterpret_cast<const v4f*>(out + off);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = o[e] > 0.f ? g[e] : (act == 1 ? g[e] * (o[e] + 1.0f) : 0.0f);
        *reinterpret_cast<v4f*>(g_pre + off) = g;
      }
      acc += g;
    }
    if (partial != nullptr)
      *reinterpret_cast<v4f*>(partial + static_cast<size_t>(blockIdx.x) * cols + 4 * q) = acc;
  }
}

__global__ __launch_bounds__(kBlock) void gat_param_grad_kernel(
    const float* __restrict__ ft, const float* __restrict__ gel, const float* __restrict__ ger,
    float* __restrict__ partial_l, float* __restrict__ partial_r, int64_t n, int heads, int dim,
    int64_t rpc) {
  const int64_t row0 = blockIdx.x * rpc, row1 = min(n, row0 + rpc);
  const int cols = heads * dim, cols4 = cols >> 2;
  for (int q = threadIdx.x; q < cols4; q += kBlock) {
    const int h = (4 * q) / dim;
    v4f al = {0.f, 0.f, 0.f, 0.f}, ar = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int64_t row = row0; row < row1; ++row) {
      const v4f f = *reinterpret_cast<const v4f*>(ft + static_cast<size_t>(row) * cols + 4 * q);
      const float wl = gel[row * heads + h], wr = ger[row * heads + h];
      al += wl * f;
      ar += wr * f;
    }
    const size_t off = static_cast<size_t>(blockIdx.x) * cols + 4 * q;
    *reinterpret_cast<v4f*>(partial_l + off) = al;
    *reinterpret_cast<v4f*>(partial_r + off) = ar;
  }
}

// out[c] = sum over chunks of partial[chunk][c].  A workgroup covers 16 column groups x 16 chunk
// lanes: lane j adds chunks j, j+16, ... in order, then the 16 lane totals are added in lane order
// (a fixed association: bitwise reproducible).
__global__ __launch_bounds__(kBlock) void sum_chunks_kernel(const float* __restrict__ partial,
                                                           float* __restrict__ out, int cols,
                                                           int chunks) {
  __shared__ v4f part[16][16];
  const int cg = threadIdx.x & 15, lane = threadIdx.x >> 4;
  const int q = blockIdx.x * 16 + cg;
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  if (q < (cols >> 2)) {
#pragma unroll 8   // independent loads, issued back to back (the adds keep their order)
    for (int b = lane; b < chunks; b += 16)
      acc += *reinterpret_cast<const v4f*>(partial + static_cast<size_t>(b) * cols + 4 * q);
  }
  part[lane][cg] = acc;
  __syncthreads();
  if (lane == 0 && q < (cols >> 2)) {
    v4f total = part[0][cg];
#pragma unroll
    for (int j = 1; j < 16; ++j) total += part[j][cg];
    *reinterpret_cast<v4f*>(out + 4 * q) = total;
  }
}

inline bool bad(int64_t n, int64_t cols) {
  return n <= 0 || cols <= 0 || (cols & 3) != 0 || cols >= (1 << 24) || n >= (1LL << 40);
}

}  // namespace

int sum_chunks(const float* partial, float* out, int cols, int chunks, hipStream_t st) {
  sum_chunks_kernel<<<(cols / 4 + 15) / 16, kBlock, 0, st>>>(partial, out, cols, chunks);
  return launch_status();
}
}  // namespace gts

extern "C" int64_t gts_gat_reduce_workspace(int64_t n, int64_t cols) {
  using namespace gts;
  if (bad(n, cols)) return 0;
  return 2 * static_cast<int64_t>(n_chunks(n)) * cols * static_cast<int64_t>(sizeof(float));
}

extern "C" int32_t gts_gat_act_bwd_f32(const float* gout, const float* out, int32_t activation,
                                       float* g_pre, float* g_bias, float* workspace,
                                       int64_t workspace_bytes, int64_t n, int64_t cols,
                                       void* stream) {
  using namespace gts;
  if (!gout || (activation != 0 && (!out || !g_pre)) || (g_bias && !workspace)) return GTS_ERR_NULL;
  if (bad(n, cols)) return GTS_ERR_SHAPE;
  if (activation < 0 || activation > 2) return GTS_ERR_ARGKIND;
  if (g_bias && workspace_bytes < gts_gat_reduce_workspace(n, cols) / 2) return GTS_ERR_SHAPE;
  if (activation == 0 && !g_bias) return GTS_OK;  // nothing to do
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int chunks = n_chunks(n), nc = static_cast<int>(cols);
  gat_act_bwd_kernel<<<chunks, kBlock, 0, st>>>(gout, out, activation, g_pre,
                                                g_bias ? workspace : nullptr, n, nc, rows_per_chunk(n));
  if (g_bias)
    sum_chunks_kernel<<<(nc / 4 + 15) / 16, kBlock, 0, st>>>(workspace, g_bias, nc, chunks);
  return launch_status();
}

extern "C" int32_t gts_gat_param_grad_f32(const float* ft, const float* gel, const float* ger,
                                          float* g_attn_l, float* g_attn_r, float* workspace,
                                          int64_t workspace_bytes, int64_t n, int64_t heads,
                                          int64_t dim, void* stream) {
  using namespace gts;
  if (!ft || !gel || !ger || !g_attn_l || !g_attn_r || !workspace) return GTS_ERR_NULL;
  const int64_t cols = heads * dim;
  if (heads <= 0 || dim <= 0 || (dim & 3) != 0 || bad(n, cols)) return GTS_ERR_SHAPE;
  if (workspace_bytes < gts_gat_reduce_workspace(n, cols)) return GTS_ERR_SHAPE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int chunks = n_chunks(n), nc = static_cast<int>(cols);
  float* pl = workspace;
  float* pr = workspace + static_cast<size_t>(chunks) * cols;
  gat_param_grad_kernel<<<chunks, kBlock, 0, st>>>(ft, gel, ger, pl, pr, n, static_cast<int>(heads),
                                                   static_cast<int>(dim), rows_per_chunk(n));
  const unsigned grid = (nc / 4 + 15) / 16;
  sum_chunks_kernel<<<grid, kBlock, 0, st>>>(pl, g_attn_l, nc, chunks);
  sum_chunks_kernel<<<grid, kBlock, 0, st>>>(pr, g_attn_r, nc, chunks);
  return launch_status();
}
