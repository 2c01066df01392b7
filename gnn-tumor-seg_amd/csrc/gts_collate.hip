// K9h: batch collate on the host, straight into the upload layout (include/gts_hip.h: gts_collate_plan,
// gts_collate_batch).
//
// Stands in, on the training loader's hot path, for `minibatch_graphs` (/root/reference/data_processing/
// data_loader.py:165-169: dgl.batch + np.concatenate + FloatTensor / LongTensor) and the three `.to(device)`
// calls of the training loop (/root/reference/model/gnn_model.py:37-40).  Host code only: no GPU call, no
// allocation that outlives the call, no state.  The Python path (gts.batch, ClusterSchedule.concat) stays the
// tested reference: tests/test_collate_host.py compares the two byte for byte.
//
// Why C: the loader thread of GNN.run_epoch shares the interpreter lock with the thread that enqueues the
// training step; 10 - 20 numpy calls per member and array held that lock for milliseconds per batch
// (profiles/r03_epoch_throughput.jsonl).  One ctypes call releases it for the whole assembly.
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "gts_cluster.h"
#include "gts_common.h"

namespace {

constexpr int64_t kI32Max = 2147483647LL;

inline int64_t align256(int64_t bytes) { return (bytes + 255) & ~255LL; }

struct Offsets {   // per member: first node / edge of the member inside the union
  std::vector<int64_t> node, edge;
  std::vector<int64_t> cluster[GTS_COLLATE_MAX_SCHEDULES];
};

int32_t make_plan(const gts_collate_member_t* members, int32_t n_members, int64_t feat_width,
                  const gts_collate_kind_t* kinds, int32_t n_kinds, gts_collate_plan_t* plan, Offsets* offs) {
  if (!members || !plan || (n_kinds > 0 && !kinds)) return GTS_ERR_NULL;
  if (n_members < 1 || feat_width < 0 || n_kinds < 0 || n_kinds > GTS_COLLATE_MAX_SCHEDULES) return GTS_ERR_SHAPE;
  int64_t n = 0, e = 0;
  const bool labelled = members[0].labels != nullptr;
  if (offs) offs->node.assign(n_members + 1, 0), offs->edge.assign(n_members + 1, 0);
  for (int i = 0; i < n_members; ++i) {
    const gts_collate_member_t& m = members[i];
    if (m.n_nodes < 0 || m.n_edges < 0) return GTS_ERR_SHAPE;
    if (!m.indptr || !m.t_indptr) return GTS_ERR_NULL;
    if (m.n_edges > 0 && (!m.indices || !m.t_indices || !m.t_slot || !m.t_pos)) return GTS_ERR_NULL;
    if (m.n_nodes > 0 && feat_width > 0 && !m.features) return GTS_ERR_NULL;
    if ((m.labels != nullptr) != labelled) return GTS_ERR_NULL;      // all members carry labels, or none
    if (feat_width > 0 && m.feat_bytes != 4 && m.feat_bytes != 8) return GTS_ERR_ARGKIND;
    if (labelled && m.label_bytes != 8 && m.label_bytes != 4) return GTS_ERR_ARGKIND;
    if (offs) offs->node[i] = n, offs->edge[i] = e;
    n += m.n_nodes, e += m.n_edges;
    if (n > kI32Max || e > kI32Max) return GTS_ERR_SHAPE;
  }
  if (offs) offs->node[n_members] = n, offs->edge[n_members] = e;
  std::memset(plan, 0, sizeof(*plan));
  plan->n_nodes = n, plan->n_edges = e;
  int64_t at = 0;
  auto take = [&](int64_t bytes) { const int64_t o = at; at += align256(bytes); return o; };
  plan->features = take(4 * n * feat_width);
  plan->labels = labelled ? take(8 * n) : -1;
  const int64_t seg[8] = {n + 1, e, n + 1, e, e, e, n, n};
  for (int s = 0; s < 8; ++s) plan->csr[s] = take(4 * seg[s]);
  for (int k = 0; k < GTS_COLLATE_MAX_SCHEDULES; ++k) plan->sched[k] = -1, plan->sched_clusters[k] = -1;
  for (int k = 0; k < n_kinds; ++k) {
    if (kinds[k].max_rows < 1 || kinds[k].max_srcs < 1 || kinds[k].max_srcs > 256) return GTS_ERR_SHAPE;
    bool all = true;
    int64_t clusters = 0;
    int32_t lw = 0;
    if (offs) offs->cluster[k].assign(n_members + 1, 0);
    for (int i = 0; i < n_members; ++i) {
      const gts_collate_member_t& m = members[i];
      if (!m.sched_rec[k]) { all = false; break; }
      if (m.sched_clusters[k] < 0 || m.sched_loc_words[k] < 0 || (m.sched_loc_words[k] & 3)) return GTS_ERR_SHAPE;
      if (offs) offs->cluster[k][i] = clusters;
      clusters += m.sched_clusters[k];
      lw = std::max(lw, m.sched_loc_words[k]);
    }
    if (!all) continue;      // a member without a (worthwhile) schedule: the union has none (gts.Graph.cluster_schedule)
    if (offs) offs->cluster[k][n_members] = clusters;
    const gts::RecLayout lay = gts::rec_layout(kinds[k].max_rows, kinds[k].max_srcs, lw, kinds[k].tagged != 0);
    plan->sched_clusters[k] = clusters;
    plan->sched_loc_words[k] = lw;
    plan->sched_record_words[k] = lay.words;
    plan->sched[k] = take(4 * clusters * lay.words);
  }
  plan->total_bytes = at;
  return GTS_OK;
}

// out[j] = in[j] + shift over `count` ints (shift may be 0)
inline void copy_shifted(int32_t* out, const int32_t* in, int64_t count, int32_t shift) {
  if (shift == 0) {
    std::memcpy(out, in, 4 * count);
    return;
  }
  for (int64_t j = 0; j < count; ++j) out[j] = in[j] + shift;
}

struct Job {
  const gts_collate_member_t* members;
  const gts_collate_kind_t* kinds;
  const gts_collate_plan_t* plan;
  const Offsets* offs;
  int32_t n_members, n_kinds;
  int64_t feat_width;
  char* dst;
};

void features_and_labels(const Job& j, int i) {
  const gts_collate_member_t& m = j.members[i];
  const int64_t count = m.n_nodes * j.feat_width;
  float* out = reinterpret_cast<float*>(j.dst + j.plan->features) + j.offs->node[i] * j.feat_width;
  if (count > 0) {
    if (m.feat_bytes == 4) {
      std::memcpy(out, m.features, 4 * count);
    } else {
      const double* in = static_cast<const double*>(m.features);
      for (int64_t q = 0; q < count; ++q) out[q] = static_cast<float>(in[q]);   // round to nearest even, as numpy / torch
    }
  }
  if (j.plan->labels >= 0) {
    int64_t* lo = reinterpret_cast<int64_t*>(j.dst + j.plan->labels) + j.offs->node[i];
    if (m.label_bytes == 8) {
      std::memcpy(lo, m.labels, 8 * m.n_nodes);
    } else {
      const int32_t* in = static_cast<const int32_t*>(m.labels);
      for (int64_t q = 0; q < m.n_nodes; ++q) lo[q] = in[q];
    }
  }
}

void csr_in(const Job& j, int i) {   // indptr, indices, the two degree vectors
  const gts_collate_member_t& m = j.members[i];
  const int64_t n0 = j.offs->node[i], e0 = j.offs->edge[i];
  int32_t* indptr = reinterpret_cast<int32_t*>(j.dst + j.plan->csr[0]);
  copy_shifted(indptr + n0, m.indptr, m.n_nodes, static_cast<int32_t>(e0));
  if (i == j.n_members - 1) indptr[j.plan->n_nodes] = static_cast<int32_t>(j.plan->n_edges);
  copy_shifted(reinterpret_cast<int32_t*>(j.dst + j.plan->csr[1]) + e0, m.indices, m.n_edges, static_cast<int32_t>(n0));
  float* clamped = reinterpret_cast<float*>(j.dst + j.plan->csr[6]) + n0;
  float* plus1 = reinterpret_cast<float*>(j.dst + j.plan->csr[7]) + n0;
  for (int64_t v = 0; v < m.n_nodes; ++v) {
    const float deg = static_cast<float>(m.indptr[v + 1] - m.indptr[v]);
    clamped[v] = deg > 1.0f ? deg : 1.0f;
    plus1[v] = deg + 1.0f;
  }
}

void csr_out(const Job& j, int i) {   // t_indptr, t_indices, t_slot, t_pos
  const gts_collate_member_t& m = j.members[i];
  const int64_t n0 = j.offs->node[i], e0 = j.offs->edge[i];
  int32_t* t_indptr = reinterpret_cast<int32_t*>(j.dst + j.plan->csr[2]);
  copy_shifted(t_indptr + n0, m.t_indptr, m.n_nodes, static_cast<int32_t>(e0));
  if (i == j.n_members - 1) t_indptr[j.plan->n_nodes] = static_cast<int32_t>(j.plan->n_edges);
  copy_shifted(reinterpret_cast<int32_t*>(j.dst + j.plan->csr[3]) + e0, m.t_indices, m.n_edges, static_cast<int32_t>(n0));
  copy_shifted(reinterpret_cast<int32_t*>(j.dst + j.plan->csr[4]) + e0, m.t_slot, m.n_edges, 0);
  copy_shifted(reinterpret_cast<int32_t*>(j.dst + j.plan->csr[5]) + e0, m.t_pos, m.n_edges, static_cast<int32_t>(e0));
}

void schedule(const Job& j, int i, int k) {
  if (j.plan->sched[k] < 0) return;
  const gts_collate_member_t& m = j.members[i];
  const gts_collate_kind_t& kind = j.kinds[k];
  const bool tagged = kind.tagged != 0;
  const int32_t lw_in = m.sched_loc_words[k], lw_out = j.plan->sched_loc_words[k];
  const gts::RecLayout in = gts::rec_layout(kind.max_rows, kind.max_srcs, lw_in, tagged);
  const gts::RecLayout out = gts::rec_layout(kind.max_rows, kind.max_srcs, lw_out, tagged);
  const int32_t shift = static_cast<int32_t>(j.offs->node[i]);
  int32_t* dst = reinterpret_cast<int32_t*>(j.dst + j.plan->sched[k]) + j.offs->cluster[k][i] * out.words;
  const int32_t* src = m.sched_rec[k];
  for (int64_t c = 0; c < m.sched_clusters[k]; ++c, src += in.words, dst += out.words) {
    std::memcpy(dst, src, 4 * in.rows);                                        // n_rows, n_srcs, padded edges, 0
    copy_shifted(dst + out.rows, src + in.rows, in.eoff - in.rows, shift);     // row ids and neighbour ids
    std::memcpy(dst + out.eoff, src + in.eoff, 4 * (in.loc - in.eoff));        // first chunk | degree per row
    std::memcpy(dst + out.loc, src + in.loc, 4 * lw_in);
    if (lw_out > lw_in) std::memset(dst + out.loc + lw_in, 0, 4 * (lw_out - lw_in));
    if (tagged) {
      std::memcpy(dst + out.tag, src + in.tag, 4 * lw_in);
      if (lw_out > lw_in) std::memset(dst + out.tag + lw_in, 0, 4 * (lw_out - lw_in));
    }
  }
}

}  // namespace

extern "C" int32_t gts_collate_plan(const gts_collate_member_t* members, int32_t n_members, int64_t feat_width,
                                    const gts_collate_kind_t* kinds, int32_t n_kinds, gts_collate_plan_t* plan) {
  return make_plan(members, n_members, feat_width, kinds, n_kinds, plan, nullptr);
}

extern "C" int32_t gts_collate_batch(const gts_collate_member_t* members, int32_t n_members, int64_t feat_width,
                                     const gts_collate_kind_t* kinds, int32_t n_kinds, void* dst, int64_t dst_bytes,
                                     int32_t n_threads, gts_collate_plan_t* plan) {
  if (!dst) return GTS_ERR_NULL;
  Offsets offs;
  const int32_t code = make_plan(members, n_members, feat_width, kinds, n_kinds, plan, &offs);
  if (code != GTS_OK) return code;
  if (dst_bytes < plan->total_bytes) return GTS_ERR_SHAPE;
  for (int k = 0; k < n_kinds; ++k) {      // a member's records must have been built with the kind's limits
    if (plan->sched[k] < 0) continue;
    for (int i = 0; i < n_members; ++i)
      if (members[i].sched_loc_words[k] > plan->sched_loc_words[k]) return GTS_ERR_SHAPE;
  }
  const Job job{members, kinds, plan, &offs, n_members, n_kinds, feat_width, static_cast<char*>(dst)};
  // tasks: per member features + labels | in-CSR | out-CSR | one per schedule kind; dealt through one counter
  const int per_member = 3 + n_kinds;
  const int n_tasks = n_members * per_member;
  std::atomic<int> next{0};
  auto work = [&]() {
    for (int t = next.fetch_add(1); t < n_tasks; t = next.fetch_add(1)) {
      const int what = t / n_members, i = t % n_members;     // all members' features first, then their CSRs, ...
      if (what == 0) features_and_labels(job, i);
      else if (what == 1) csr_in(job, i);
      else if (what == 2) csr_out(job, i);
      else schedule(job, i, what - 3);
    }
  };
  const int helpers = std::max(0, std::min(n_threads, n_tasks) - 1);
  std::vector<std::thread> pool;
  pool.reserve(helpers);
  for (int t = 0; t < helpers; ++t) {
    try {
      pool.emplace_back(work);
    } catch (...) {      // no more threads to be had: the calling thread does the rest
      break;
    }
  }
  work();
  for (auto& th : pool) th.join();
  return GTS_OK;
}
