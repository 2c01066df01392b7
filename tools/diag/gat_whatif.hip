// Diagnostic translation unit (never part of libgts_hip.so): the forward streaming kernel of
// gnn-tumor-seg_amd/csrc/gts_gat_cluster.hip alone (the weight blocks in `workspace` come from a regular call before) with parts
// of its work switched off — timing only, wrong results: whatif 0 = everything, 1 = no neighbour gathers (records, weights,
// reduction, stores), 2 = gathers and records but no reduction / stores, 3 = everything but the stores.
// Built and driven by tools/diag/gat_whatif.py.
#include "../../gnn-tumor-seg_amd/csrc/gts_gat_cluster.hip"

static int g_whatif_grid = 0;    // workgroups of the launch (0 = the library's choice); 8 = ONE per XCD: the span's units strictly in order
static int g_whatif_depth = 1;
static int g_whatif_nt = 1;      // non-temporal stores of the output rows (the library's default)   // units the gathers run ahead (the library itself runs 1)

extern "C" int gts_whatif_gat_fwd(const int32_t* rec, int64_t n_clusters, int32_t max_rows, int32_t max_srcs, int32_t loc_words,
                                  const float* ft, const float* bias, int32_t activation, float* out, float* workspace, int64_t n,
                                  int64_t heads, int32_t whatif, void* stream) {
  using namespace gts;
  const GatPlan p = gat_plan(max_rows, max_srcs, loc_words, false, false, false, g_whatif_depth);
  GatClusterArgs a{};
  a.rec = rec, a.layout = rec_layout(max_rows, max_srcs, loc_words, false), a.n_clusters = static_cast<int>(n_clusters), a.max_srcs = max_srcs;
  a.table = ft, a.side = workspace, a.vec = bias, a.out = out;
  a.counters = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(workspace) + weight_block_bytes(n_clusters, heads, p.side_floats));
  if (hipMemsetAsync(a.counters, 0, kCounterBytes, static_cast<hipStream_t>(stream)) != hipSuccess) return -100;
  a.row_bytes = static_cast<unsigned>(heads * kF * 4);
  a.table_bytes = static_cast<unsigned>(n * a.row_bytes);
  a.side_bytes = static_cast<unsigned>(n_clusters * heads * p.side_floats * 4);
  a.vec_bytes = static_cast<unsigned>(heads * kF * 4);
  a.heads = static_cast<int>(heads), a.act = activation, a.nt = g_whatif_nt;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (whatif == 1) return launch_gat_cluster<0, 1, 3>(a, p, st, g_whatif_grid);
  if (whatif == 2) return launch_gat_cluster<0, 2, 3>(a, p, st, g_whatif_grid);
  if (whatif == 3) return launch_gat_cluster<0, 3, 3>(a, p, st, g_whatif_grid);
  if (whatif == 9) return launch_gat_cluster<0, 9, 3>(a, p, st, g_whatif_grid);   // out = stamp buffer
  return launch_gat_cluster<0, 0, 3>(a, p, st, g_whatif_grid);
}

extern "C" void gts_whatif_grid(int32_t grid) { g_whatif_grid = grid; }
extern "C" void gts_whatif_nt(int32_t nt) { g_whatif_nt = nt; }
extern "C" void gts_whatif_dealing(int32_t dealing) { gts::g_gat_cluster_dealing = dealing; }

extern "C" void gts_whatif_knobs(int32_t depth, int32_t per_cu, int32_t waves, int32_t group) {
  g_whatif_depth = depth, gts::g_cluster_per_cu = per_cu, gts::g_gat_cluster_waves = waves, gts::g_gat_cluster_group = group;
}
