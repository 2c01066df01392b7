// A whole stack of SAGEConv('pool') layers behind ONE C-ABI call each way.
//
// Replaces the per-layer Python loop of GraphSage.forward (/root/reference/model/networks.py:32-36: `for layer in
// self.layers: h = layer(graph, h)`) and its autograd for the case every reference configuration of the path has:
// pool aggregator, ReLU on all but the last layer, bias on, no active dropout.  Host code only: it enqueues the
// kernels of this library (K1 / K2, K11 and their chained / transposed / batched forms) in exactly the order and
// with exactly the arguments gts/nn.py::_SagePoolStack issues them one ctypes call at a time, so the results are
// bit-identical (tests/test_gpu_stack.py) while a training step costs the host 4 calls instead of 55.
// The caller owns every byte: activations and winners live in one forward arena, gradients land where the caller's
// pointer table says (e.g. inside a flat gradient buffer), scratch comes in one block.
#include <algorithm>
#include <vector>

#include "gts_common.h"

namespace {

constexpr int kMaxLayers = 64;
constexpr int kMaxWgradProblems = 32;   // kMaxProblems of gts_gemm.hip
constexpr int kFlagChain = 1, kFlagReluBits = 2, kFlagTransposedIgrad = 4;

inline int64_t align256(int64_t bytes) { return (bytes + 255) & ~255LL; }

inline bool bad_stack(int64_t n_rows, const int64_t* widths, int32_t n_layers) {
  if (!widths || n_layers < 1 || n_layers > kMaxLayers || n_rows < 0 || n_rows >= (1LL << 31)) return true;
  for (int i = 0; i <= n_layers; ++i)
    if (widths[i] <= 0 || widths[i] % 4 != 0 || widths[i] >= (1 << 20)) return true;
  return false;
}

struct FwdPlan {
  int64_t m[kMaxLayers], arg[kMaxLayers], out[kMaxLayers], bits[kMaxLayers];   // byte offsets; -1 = absent
  int64_t p[2];
  int64_t wp[kMaxLayers][3];   // w_pool / w_self / w_neigh in fragment order (gts_pack_weights_f32); -1 = read as stored
  int64_t counters;            // unit counters of the clustered K1 launches (zeroed once per call; every launch leaves them zero)
  int64_t total;
};

// weights the panel GEMMs may read (outputs wider than 128 columns): kept in fragment order beside the activations
inline bool packs(int64_t out_cols) { return out_cols > 128; }

inline FwdPlan plan_forward(int64_t n, const int64_t* w, int n_layers, bool training, int arg_bytes, int flags) {
  FwdPlan p{};
  int64_t at = 0, widest = 0;
  auto take = [&](int64_t bytes) { const int64_t o = at; at += align256(bytes); return o; };
  for (int i = 0; i < n_layers; ++i) {
    const bool last = i == n_layers - 1;
    widest = std::max(widest, w[i]);
    p.m[i] = take(4 * n * w[i]);
    p.arg[i] = training ? take(static_cast<int64_t>(arg_bytes) * n * w[i]) : -1;
    p.out[i] = take(4 * n * w[i + 1]);
    const bool bits = (flags & kFlagReluBits) && training && !last && gts_relu_bits_pay(n, w[i + 1]) == 1;
    p.bits[i] = bits ? take(gts_relu_bits_bytes(n, w[i + 1])) : -1;
  }
  p.p[0] = take(4 * n * widest);
  p.p[1] = take(4 * n * widest);
  for (int i = 0; i < n_layers; ++i) {
    p.wp[i][0] = packs(w[i]) ? take(4 * gts_packed_weight_floats(w[i], w[i])) : -1;
    p.wp[i][1] = packs(w[i + 1]) ? take(4 * gts_packed_weight_floats(w[i + 1], w[i])) : -1;
    p.wp[i][2] = packs(w[i + 1]) ? take(4 * gts_packed_weight_floats(w[i + 1], w[i])) : -1;
  }
  p.counters = take(4 * GTS_CLUSTER_COUNTER_WORDS);
  p.total = at;
  return p;
}

// the launch conditions of gts/nn.py::_chainable: out has n columns from operands k0 / k1 wide, then out @ w2^T (n2 columns)
inline bool chainable(int64_t n, int64_t k0, int64_t k1, int64_t n2) {
  return n % 4 == 0 && k0 % 4 == 0 && k1 % 4 == 0 && n <= 256 && n2 <= 256;
}

struct Schedule {
  const int32_t* rec;
  int64_t clusters;
  int32_t rows, srcs, loc_words;
};

// weights the backward reads transposed (gts/nn.py: wide rows, both dimensions multiples of 4)
inline bool turns(int64_t rows, int64_t cols) { return cols >= 128 && rows % 4 == 0 && cols % 4 == 0; }

struct BwdPlan {
  int64_t wt[kMaxLayers][3];               // transposed w_pool / w_self / w_neigh (-1 = read as stored)
  int64_t wtp[kMaxLayers][3];              // the same transposes in fragment order (present whenever wt is)
  int64_t g[kMaxLayers], gp[kMaxLayers];   // g[i] (i >= 1): gradient w.r.t. layer i-1's pre-activation output; gp[i]: w.r.t. fc_pool's
  int64_t gm[2];
  int64_t workspace, workspace_bytes;
  int64_t counters;                        // unit counters of the clustered K2 launches (as FwdPlan::counters)
  int64_t total;
};

struct WgradGroup {
  int64_t n, k;
  std::vector<const float*> g, a;
  std::vector<float*> gw, gb;
};

inline BwdPlan plan_backward(int64_t n, const int64_t* w, int n_layers, int flags) {
  BwdPlan p{};
  int64_t at = 0, widest = 0;
  auto take = [&](int64_t bytes) { const int64_t o = at; at += align256(bytes); return o; };
  const bool t = (flags & kFlagTransposedIgrad) != 0;
  for (int i = 0; i < n_layers; ++i) {
    const int64_t fin = w[i], fout = w[i + 1];
    widest = std::max(widest, fin);
    p.wt[i][0] = t && turns(fin, fin) ? take(4 * fin * fin) : -1;
    p.wt[i][1] = t && turns(fout, fin) ? take(4 * fout * fin) : -1;
    p.wt[i][2] = t && turns(fout, fin) ? take(4 * fout * fin) : -1;
    p.wtp[i][0] = p.wt[i][0] >= 0 ? take(4 * gts_packed_weight_floats(fin, fin)) : -1;       // W^T is [fin, rows of W]
    p.wtp[i][1] = p.wt[i][1] >= 0 ? take(4 * gts_packed_weight_floats(fin, fout)) : -1;
    p.wtp[i][2] = p.wt[i][2] >= 0 ? take(4 * gts_packed_weight_floats(fin, fout)) : -1;
    p.g[i] = i > 0 ? take(4 * n * fin) : -1;
    p.gp[i] = take(4 * n * fin);
  }
  p.gm[0] = take(4 * n * widest);
  p.gm[1] = take(4 * n * widest);
  // split-reduction slabs: the largest request of any weight-gradient group (the grouping of the run below)
  struct Key { int64_t n, k; int count; };
  std::vector<Key> groups;
  auto add = [&](int64_t nn, int64_t kk) {
    for (auto& gk : groups)
      if (gk.n == nn && gk.k == kk) { ++gk.count; return; }
    groups.push_back({nn, kk, 1});
  };
  for (int i = n_layers - 1; i >= 0; --i) add(w[i], w[i]), add(w[i + 1], w[i]), add(w[i + 1], w[i]);
  int64_t ws = 0;
  if (n > 0)
    for (const auto& gk : groups)
      for (int first = 0; first < gk.count; first += kMaxWgradProblems)
        ws = std::max(ws, gts_linear_bwd_weight_workspace(n, gk.n, gk.k, std::min(kMaxWgradProblems, gk.count - first)));
  p.workspace_bytes = ws;
  p.workspace = take(ws);
  p.counters = take(4 * GTS_CLUSTER_COUNTER_WORDS);
  p.total = at;
  return p;
}

#define GTS_TRY(call)                    \
  do {                                   \
    const int32_t code_ = (call);        \
    if (code_ != GTS_OK) return code_;   \
  } while (0)

}  // namespace

extern "C" int64_t gts_sage_pool_stack_fwd_arena(int64_t n_rows, const int64_t* widths, int32_t n_layers, int32_t training,
                                                 int32_t arg_bytes, int32_t flags, int64_t* offsets) {
  if (bad_stack(n_rows, widths, n_layers) || (training && arg_bytes != 1 && arg_bytes != 4)) return -1;
  const FwdPlan p = plan_forward(n_rows, widths, n_layers, training != 0, arg_bytes, flags);
  if (offsets != nullptr) {
    for (int i = 0; i < n_layers; ++i)
      offsets[4 * i] = p.m[i], offsets[4 * i + 1] = p.arg[i], offsets[4 * i + 2] = p.out[i], offsets[4 * i + 3] = p.bits[i];
    offsets[4 * n_layers] = p.p[0], offsets[4 * n_layers + 1] = p.p[1];
  }
  return p.total;
}

extern "C" int32_t gts_sage_pool_stack_fwd_f32(const int32_t* indptr, const int32_t* indices, const int32_t* sched_rec,
                                               int64_t sched_clusters, int32_t sched_rows, int32_t sched_srcs,
                                               int32_t sched_loc_words, const float* x, const float* const* params,
                                               int64_t n_rows, const int64_t* widths, int32_t n_layers, int32_t training,
                                               int32_t arg_bytes, int32_t flags, void* arena, int64_t arena_bytes,
                                               void* stream) {
  if (!indptr || !x || !params || !arena) return GTS_ERR_NULL;
  if (bad_stack(n_rows, widths, n_layers)) return GTS_ERR_SHAPE;
  if (training && arg_bytes != 1 && arg_bytes != 4) return GTS_ERR_ARGKIND;
  for (int i = 0; i < 5 * n_layers; ++i)
    if (!params[i]) return GTS_ERR_NULL;
  const bool train = training != 0;
  const FwdPlan plan = plan_forward(n_rows, widths, n_layers, train, arg_bytes, flags);
  if (arena_bytes < plan.total) return GTS_ERR_SHAPE;
  if (n_rows == 0) return GTS_OK;
  char* base = static_cast<char*>(arena);
  auto f32 = [&](int64_t off) { return reinterpret_cast<float*>(base + off); };
  uint32_t* counters = nullptr;   // unit counters of the clustered K1 launches: zeroed once here, every launch leaves them zero
  if (sched_rec != nullptr && gts_cluster_uses_counters(sched_clusters, sched_rows, sched_srcs, sched_loc_words, 0) != 0) {
    counters = reinterpret_cast<uint32_t*>(base + plan.counters);
    if (hipMemsetAsync(counters, 0, 4 * GTS_CLUSTER_COUNTER_WORDS, static_cast<hipStream_t>(stream)) != hipSuccess) return GTS_ERR_SHAPE;
  }
  // the weights the panel GEMMs will read, in fragment order: one launch per weight shape (the weights change once per
  // optimizer step; 19 matrices of 256 KiB at C2)
  {
    struct Batch { int64_t rows, cols; std::vector<const float*> src; std::vector<float*> dst; };
    std::vector<Batch> batches;
    for (int i = 0; i < n_layers; ++i) {
      const int64_t fin = widths[i], fout = widths[i + 1];
      const int64_t shape[3][2] = {{fin, fin}, {fout, fin}, {fout, fin}};
      const int which[3] = {0, 2, 3};
      for (int q = 0; q < 3; ++q) {
        if (plan.wp[i][q] < 0) continue;
        Batch* b = nullptr;
        for (auto& c : batches)
          if (c.rows == shape[q][0] && c.cols == shape[q][1]) b = &c;
        if (b == nullptr) batches.push_back({shape[q][0], shape[q][1], {}, {}}), b = &batches.back();
        b->src.push_back(params[5 * i + which[q]]);
        b->dst.push_back(f32(plan.wp[i][q]));
      }
    }
    for (const auto& b : batches)
      GTS_TRY(gts_pack_weights_f32(b.src.data(), b.dst.data(), nullptr, static_cast<int32_t>(b.src.size()), b.rows, b.cols, 0,
                                   stream));
  }
  auto wp = [&](int i, int q) -> const float* { return plan.wp[i][q] >= 0 ? f32(plan.wp[i][q]) : nullptr; };
  const float* h = x;
  float* p = nullptr;   // relu(fc_pool(h)) of the layer about to run, when the previous launch already made it
  int cur = 0;
  for (int i = 0; i < n_layers; ++i) {
    const float* w_pool = params[5 * i];
    const float* b_pool = params[5 * i + 1];
    const float* w_self = params[5 * i + 2];
    const float* w_neigh = params[5 * i + 3];
    const float* bias = params[5 * i + 4];
    const int64_t fin = widths[i], fout = widths[i + 1];
    const bool last = i == n_layers - 1;
    if (p == nullptr) {
      p = f32(plan.p[cur]);
      const float* packed[2] = {wp(i, 0), nullptr};
      GTS_TRY(gts_linear_fwd_f32(h, w_pool, nullptr, nullptr, b_pool, p, n_rows, fin, fin, 0, 1, nullptr, packed, stream));
    }
    float* m = f32(plan.m[i]);
    void* arg = train ? static_cast<void*>(base + plan.arg[i]) : nullptr;
    const int ab = train ? arg_bytes : 0;
    if (sched_rec != nullptr && fin == 256 && ab != 4 && n_rows * 1024 < (1LL << 32)) {
      GTS_TRY(gts_spmm_max_fwd_cluster_f32(sched_rec, sched_clusters, sched_rows, sched_srcs, sched_loc_words, p, m, arg, ab,
                                           1, n_rows, fin, counters, stream));
    } else {
      GTS_TRY(gts_spmm_max_fwd_f32(indptr, indices, p, m, arg, ab, 1, n_rows, fin, stream));
    }
    uint64_t* bits = plan.bits[i] >= 0 ? reinterpret_cast<uint64_t*>(base + plan.bits[i]) : nullptr;
    float* out = f32(plan.out[i]);
    if ((flags & kFlagChain) && !last && chainable(fout, fin, fin, fout)) {
      // fc_self + fc_neigh of this layer and fc_pool of the next one in one launch
      float* p_next = f32(plan.p[cur ^ 1]);
      const float* packed[3] = {wp(i, 1), wp(i, 2), wp(i + 1, 0)};
      GTS_TRY(gts_linear_fwd_chain_f32(h, w_self, m, w_neigh, bias, out, params[5 * (i + 1)], params[5 * (i + 1) + 1], p_next,
                                       n_rows, fout, fin, fin, 1, fout, 1, bits, packed, stream));
      p = p_next, cur ^= 1;
    } else {
      const float* packed[2] = {wp(i, 1), wp(i, 2)};
      GTS_TRY(gts_linear_fwd_f32(h, w_self, m, w_neigh, bias, out, n_rows, fout, fin, fin, last ? 0 : 1, bits, packed, stream));
      p = nullptr;
    }
    h = out;
  }
  return GTS_OK;
}

extern "C" int64_t gts_sage_pool_stack_bwd_scratch(int64_t n_rows, const int64_t* widths, int32_t n_layers, int32_t flags) {
  if (bad_stack(n_rows, widths, n_layers)) return -1;
  return plan_backward(n_rows, widths, n_layers, flags).total;
}

extern "C" int32_t gts_sage_pool_stack_bwd_f32(const int32_t* t_indptr, const int32_t* t_indices, const int32_t* t_slot,
                                               const int32_t* sched_rec, int64_t sched_clusters, int32_t sched_rows,
                                               int32_t sched_srcs, int32_t sched_loc_words, const float* gout, const float* x,
                                               const float* const* params, int64_t n_rows, const int64_t* widths,
                                               int32_t n_layers, int32_t arg_bytes, int32_t flags, const void* fwd_arena,
                                               float* const* grads, float* gx, void* scratch, int64_t scratch_bytes,
                                               void* stream) {
  if (!t_indptr || !gout || !x || !params || !fwd_arena || !grads || !scratch) return GTS_ERR_NULL;
  if (bad_stack(n_rows, widths, n_layers)) return GTS_ERR_SHAPE;
  if (arg_bytes != 1 && arg_bytes != 4) return GTS_ERR_ARGKIND;
  for (int i = 0; i < 5 * n_layers; ++i)
    if (!params[i] || !grads[i]) return GTS_ERR_NULL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n_rows == 0) {   // no rows: every gradient is zero
    for (int i = 0; i < n_layers; ++i) {
      const int64_t fin = widths[i], fout = widths[i + 1];
      const int64_t sizes[5] = {fin * fin, fin, fout * fin, fout * fin, fout};
      for (int q = 0; q < 5; ++q)
        if (hipMemsetAsync(grads[5 * i + q], 0, 4 * sizes[q], st) != hipSuccess) return gts::launch_status();
    }
    return GTS_OK;
  }
  const FwdPlan fwd = plan_forward(n_rows, widths, n_layers, true, arg_bytes, flags);
  const BwdPlan plan = plan_backward(n_rows, widths, n_layers, flags);
  if (scratch_bytes < plan.total) return GTS_ERR_SHAPE;
  const char* acts = static_cast<const char*>(fwd_arena);
  char* base = static_cast<char*>(scratch);
  auto f32 = [&](int64_t off) { return reinterpret_cast<float*>(base + off); };
  uint32_t* counters = nullptr;   // unit counters of the clustered K2 launches (as in the forward call)
  if (sched_rec != nullptr && gts_cluster_uses_counters(sched_clusters, sched_rows, sched_srcs, sched_loc_words, 1) != 0) {
    counters = reinterpret_cast<uint32_t*>(base + plan.counters);
    if (hipMemsetAsync(counters, 0, 4 * GTS_CLUSTER_COUNTER_WORDS, static_cast<hipStream_t>(stream)) != hipSuccess) return GTS_ERR_SHAPE;
  }
  auto act = [&](int64_t off) { return reinterpret_cast<const float*>(acts + off); };
  auto input_of = [&](int i) { return i == 0 ? x : act(fwd.out[i - 1]); };

  // transposed weights: one batch per shape, in the order the shapes first appear (pool, self, neigh per layer)
  {
    struct Batch { int64_t rows, cols; std::vector<const float*> src; std::vector<float*> dst, packed; };
    std::vector<Batch> batches;
    for (int i = 0; i < n_layers; ++i) {
      const int64_t fin = widths[i], fout = widths[i + 1];
      const int64_t shape[3][2] = {{fin, fin}, {fout, fin}, {fout, fin}};
      const int which[3] = {0, 2, 3};
      for (int q = 0; q < 3; ++q) {
        if (plan.wt[i][q] < 0) continue;
        Batch* b = nullptr;
        for (auto& c : batches)
          if (c.rows == shape[q][0] && c.cols == shape[q][1]) b = &c;
        if (b == nullptr) batches.push_back({shape[q][0], shape[q][1], {}, {}, {}}), b = &batches.back();
        b->src.push_back(params[5 * i + which[q]]);
        b->dst.push_back(f32(plan.wt[i][q]));
        b->packed.push_back(f32(plan.wtp[i][q]));
      }
    }
    // W^T twice in one launch per shape: row-major (the tiles that stage through LDS) and in fragment order (the panel kernels)
    for (const auto& b : batches)
      GTS_TRY(gts_pack_weights_f32(b.src.data(), b.packed.data(), b.dst.data(), static_cast<int32_t>(b.src.size()), b.rows,
                                   b.cols, 1, stream));
  }
  auto wt = [&](int i, int q) -> const float* { return plan.wt[i][q] >= 0 ? f32(plan.wt[i][q]) : nullptr; };
  auto wtp = [&](int i, int q) -> const float* { return plan.wtp[i][q] >= 0 ? f32(plan.wtp[i][q]) : nullptr; };

  // input gradient g0 @ w0 [+ g1 @ w1] into `gin` [n_rows, k]: transposed weights when every operand has them
  auto igrad = [&](const float* g0, const float* w0, const float* w0t, int64_t n0, const float* g1, const float* w1,
                   const float* w1t, int64_t n1, const float* relu_mask, const uint64_t* relu_bits, float* gin,
                   int64_t k, const float* w0tp, const float* w1tp) -> int32_t {
    if (w0t != nullptr && (g1 == nullptr || w1t != nullptr)) {
      const float* packed[2] = {w0tp, g1 ? w1tp : nullptr};
      return gts_linear_bwd_input_t_f32(g0, w0t, g1, g1 ? w1t : nullptr, relu_mask, relu_mask ? relu_bits : nullptr, gin,
                                        n_rows, k, n0, g1 ? n1 : 0, packed, stream);
    }
    return gts_linear_bwd_input_f32(g0, w0, g1, g1 ? w1 : nullptr, relu_mask, gin, n_rows, k, n0, g1 ? n1 : 0, stream);
  };

  std::vector<WgradGroup> groups;   // weight-gradient problems by shape, in order of first appearance
  auto defer = [&](const float* g, int64_t n, const float* a, int64_t k, float* gw, float* gb) {
    WgradGroup* grp = nullptr;
    for (auto& c : groups)
      if (c.n == n && c.k == k) grp = &c;
    if (grp == nullptr) groups.push_back({n, k, {}, {}, {}, {}}), grp = &groups.back();
    grp->g.push_back(g), grp->a.push_back(a), grp->gw.push_back(gw), grp->gb.push_back(gb);
  };

  const float* g = gout;    // gradient w.r.t. the pre-activation output of layer i
  const float* gm = nullptr;   // g @ W_neigh of the layer about to run, when the previous launch already made it
  int cur = 0;
  for (int i = n_layers - 1; i >= 0; --i) {
    const float* w_pool = params[5 * i];
    const float* w_self = params[5 * i + 2];
    const float* w_neigh = params[5 * i + 3];
    const int64_t fin = widths[i], fout = widths[i + 1];
    const float* h = input_of(i);
    const float* m = act(fwd.m[i]);
    const void* arg = acts + fwd.arg[i];
    if (gm == nullptr) {
      float* buf = f32(plan.gm[cur]);
      GTS_TRY(igrad(g, w_neigh, wt(i, 2), fout, nullptr, nullptr, nullptr, 0, nullptr, nullptr, buf, fin, wtp(i, 2), nullptr));
      gm = buf;
    }
    float* gp = f32(plan.gp[i]);   // ReLU'(p) is already in the winner record
    if (sched_rec != nullptr && fin == 256 && arg_bytes == 1 && n_rows * 1024 < (1LL << 32)) {
      GTS_TRY(gts_spmm_max_bwd_cluster_f32(sched_rec, sched_clusters, sched_rows, sched_srcs, sched_loc_words, gm, arg, 1, gp,
                                           n_rows, fin, counters, stream));
    } else {
      GTS_TRY(gts_spmm_max_bwd_f32(t_indptr, t_indices, t_slot, gm, arg, arg_bytes, nullptr, gp, n_rows, fin, stream));
    }
    gm = nullptr;
    defer(gp, fin, h, fin, grads[5 * i], grads[5 * i + 1]);        // fc_pool.weight, fc_pool.bias
    defer(g, fout, h, fin, grads[5 * i + 2], grads[5 * i + 4]);    // fc_self.weight, bias
    defer(g, fout, m, fin, grads[5 * i + 3], nullptr);             // fc_neigh.weight
    if (i > 0) {   // h is layer i-1's ReLU output: its backward is the mask h > 0
      const int64_t below_in = widths[i - 1];
      const uint64_t* hbits = fwd.bits[i - 1] >= 0 ? reinterpret_cast<const uint64_t*>(acts + fwd.bits[i - 1]) : nullptr;
      float* g_next = f32(plan.g[i]);
      const float* below_t = wt(i - 1, 2);
      if ((flags & kFlagChain) && wt(i, 1) && wt(i, 0) && below_t && chainable(fin, fout, fin, below_in)) {
        // this layer's input gradient and the next one's g @ W_neigh in one launch
        float* gm_next = f32(plan.gm[cur ^ 1]);
        const float* packed[3] = {wtp(i, 1), wtp(i, 0), wtp(i - 1, 2)};
        GTS_TRY(gts_linear_bwd_input_chain_t_f32(g, wt(i, 1), gp, wt(i, 0), h, hbits, g_next, below_t, gm_next, n_rows, fin,
                                                 fout, fin, below_in, packed, stream));
        gm = gm_next, cur ^= 1;
      } else {
        GTS_TRY(igrad(g, w_self, wt(i, 1), fout, gp, w_pool, wt(i, 0), fin, h, hbits, g_next, fin, wtp(i, 1), wtp(i, 0)));
      }
      g = g_next;
    } else if (gx != nullptr) {
      GTS_TRY(igrad(g, w_self, wt(i, 1), fout, gp, w_pool, wt(i, 0), fin, nullptr, nullptr, gx, fin, wtp(i, 1), wtp(i, 0)));
    }
  }
  for (const auto& grp : groups) {
    const int count = static_cast<int>(grp.g.size());
    for (int first = 0; first < count; first += kMaxWgradProblems) {
      const int q = std::min(kMaxWgradProblems, count - first);
      GTS_TRY(gts_linear_bwd_weight_f32(grp.g.data() + first, grp.a.data() + first, grp.gw.data() + first,
                                        grp.gb.data() + first, q, f32(plan.workspace), plan.workspace_bytes, n_rows, grp.n,
                                        grp.k, stream));
    }
  }
  return GTS_OK;
}
