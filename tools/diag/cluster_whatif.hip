// Diagnostic translation unit (never part of libgts_hip.so): the persistent clustered K1 / K2 kernel of
// gnn-tumor-seg_amd/csrc/gts_spmm_cluster.hip with parts of its work switched off (timing only, wrong results):
// whatif 1 = records and reduction but no row gathers, 2 = gathers and records but no reduction / stores,
// 3 = everything but the stores (K1), 4 = gathers and stores but no reduction (K1).
// Built and driven by tools/diag/cluster_whatif.py.
#include "../../gnn-tumor-seg_amd/csrc/gts_spmm_cluster.hip"

extern "C" int gts_whatif_cluster(const int32_t* rec, int64_t n_clusters, int32_t max_rows, int32_t max_srcs,
                                  int32_t loc_words, const float* table, const uint8_t* winners, float* out, uint8_t* arg,
                                  int32_t bwd, int32_t whatif, int64_t n_rows, void* stream) {
  using namespace gts;
  ClusterArgs a{};
  a.rec = rec, a.layout = rec_layout(max_rows, max_srcs, loc_words, bwd != 0);
  a.n_clusters = static_cast<int>(n_clusters), a.max_srcs = max_srcs;
  a.table = table, a.winners = winners, a.out = out, a.arg = arg;
  a.table_bytes = static_cast<unsigned>(n_rows * kF * 4), a.winners_bytes = static_cast<unsigned>(n_rows * kF);
  a.relu_input = 1, a.nt = 1;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (bwd) {
    if (whatif == 1) return launch_cluster<true, 1, 1>(a, max_rows, loc_words, st);
    if (whatif == 2) return launch_cluster<true, 1, 2>(a, max_rows, loc_words, st);
    return launch_cluster<true, 1, 0>(a, max_rows, loc_words, st);
  }
  if (whatif == 1) return launch_cluster<false, 1, 1>(a, max_rows, loc_words, st);
  if (whatif == 2) return launch_cluster<false, 1, 2>(a, max_rows, loc_words, st);
  if (whatif == 3) return launch_cluster<false, 1, 3>(a, max_rows, loc_words, st);
  if (whatif == 4) return launch_cluster<false, 1, 4>(a, max_rows, loc_words, st);
  if (whatif == 9) return launch_cluster<false, 0, 9>(a, max_rows, loc_words, st);   // arg = stamp buffer, no winners written
  return launch_cluster<false, 1, 0>(a, max_rows, loc_words, st);
}
