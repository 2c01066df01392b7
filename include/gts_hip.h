/*
 * gts_hip.h — C ABI of libgts_hip.so, the MI355X (gfx950) kernels behind the GNN
 * node-classification path of rsinghlab/GNN-Tumor-Seg.
 *
 * The reference has no FFI of its own: its arithmetic is reached through DGL's Python
 * API.  Each entry point below names the reference call site (file:line under
 * /root/reference) and the DGL operator it stands in for.
 *
 * Conventions (all entry points)
 *   - every pointer is a DEVICE pointer owned by the caller; the library allocates
 *     nothing, frees nothing and keeps no state between calls (re-entrant);
 *   - `stream` is a hipStream_t (NULL = default stream); work is only enqueued, never
 *     synchronised; the device is whatever the caller made current;
 *   - return value: 0 on success, GTS_ERR_* (<0) for a rejected argument, otherwise a
 *     positive hipError_t from the launch.  Nothing throws or aborts;
 *   - feature matrices are dense row-major fp32; graphs are int32 CSR
 *     (E < 2^31, N < 2^31).  "in-CSR" = rows are destinations, entries are the
 *     sources of their in-edges in COO order; "out-CSR" = rows are sources.
 */
#ifndef GTS_HIP_H
#define GTS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GTS_OK 0
#define GTS_ERR_NULL (-1)    /* a required pointer is NULL            */
#define GTS_ERR_SHAPE (-2)   /* negative / overflowing / unsupported shape */
#define GTS_ERR_ARGKIND (-3) /* unsupported arg_bytes / mode value    */

/* ABI version, bumped when a signature changes. */
int32_t gts_abi_version(void);
/* Static string for a GTS_ERR_* / hipError_t code returned by this library. */
const char* gts_error_string(int32_t code);

/* ---- K1: copy_u + max reducer (forward) -------------------------------------------
 * Replaces DGL update_all(copy_u('h','m'), max('m','neigh')) inside SAGEConv('pool'),
 * reached from model/networks.py:25,28,30.
 *   out[v,f] = max_{k} x[indices[indptr[v]+k], f]; first maximum wins (strict '<');
 *   +-inf results and empty rows give 0 and select nothing.
 *   arg (optional, arg_bytes = 0 -> not written): the SLOT k of the winner inside
 *   row v, as uint8 (arg_bytes=1, slot 0xFF = none; needs max in-degree <= 254)
 *   or int32 (arg_bytes=4, -1 = none); layout [n_dst, n_feat].
 *   relu_input != 0: x is a ReLU output (SAGEConv-pool: x = relu(fc_pool(h))); a maximum that
 *   is not positive is recorded as "none", because relu'(0) = 0 stops its gradient anyway — the
 *   backward (K2) then gives the gradient w.r.t. the PRE-activation without reading x. */
int32_t gts_spmm_max_fwd_f32(const int32_t* indptr, const int32_t* indices, const float* x,
                             float* out, void* arg, int32_t arg_bytes, int32_t relu_input,
                             int64_t n_dst, int64_t n_feat, void* stream);

/* ---- K2: max reducer (backward), gather form over the out-CSR -----------------------
 * Replaces the autograd of K1 (DGL scatters gout by argmax).
 *   gx[u,f] = sum_{e in out(u)} [arg[t_indices[e], f] == t_slot[e]] * gout[t_indices[e], f]
 *   t_slot[e] = position of edge e inside its destination's in-CSR row.
 *   relu_src (optional): if given, gx[u,f] is zeroed where relu_src[u,f] <= 0
 *   (the ReLU that precedes the pooling, SAGEConv: relu(fc_pool(h))). */
int32_t gts_spmm_max_bwd_f32(const int32_t* t_indptr, const int32_t* t_indices,
                             const int32_t* t_slot, const float* gout, const void* arg,
                             int32_t arg_bytes, const float* relu_src, float* gx,
                             int64_t n_src, int64_t n_feat, void* stream);

/* ---- K1 / K2 over a cluster row schedule (F = 256): LDS-staged neighbour tiles ---------------
 * Same operators as gts_spmm_max_fwd_f32 / gts_spmm_max_bwd_f32 (DGL copy_u + max inside SAGEConv('pool'),
 * model/networks.py:25,28,30, and its autograd), bit-identical results, different work distribution: the rows
 * of the CSR are dealt to workgroups as clusters that share their neighbour rows; persistent workgroups stage
 * each distinct neighbour row of a cluster in LDS once (LDS-DMA, several clusters ahead) instead of fetching
 * one row per edge behind a chain of dependent index loads.
 *
 * gts_cluster_schedule (HOST function: host pointers, no GPU call) builds the schedule of one CSR once per
 * graph: (indptr, indices) is the CSR to schedule, (t_indptr, t_indices) its transpose, edge_tag an optional
 * per-edge payload in 0..255 (K2: t_slot).  A cluster holds <= max_rows rows, <= max_srcs (<= 256) distinct
 * neighbours and <= max_edges (<= 65535) PADDED edges (every row's edge list is padded to whole chunks of 8);
 * rows keep their edges in CSR slot order.  One fixed-size RECORD per cluster is written to rec[c * W ...],
 * W = gts_cluster_record_words(max_rows, max_srcs, LW, with_tag) int32 words with LW = ((max_edges + 3) / 4
 * rounded up to a multiple of 4):
 *   words 0..3: n_rows, n_srcs, padded edges, 0 | row ids [max_rows] | neighbour ids [max_srcs] (entries
 *   past n_srcs repeat the last one) | per row [max_rows]: index of its first 8-edge chunk (low 16 bits),
 *   degree (high 16 bits) | uint8 per padded edge: position of its neighbour in the cluster's list [4 LW]
 *   (pads repeat the row's last edge) | uint8 per padded edge: its tag [4 LW] (only with edge_tag); every
 *   section starts on a multiple of 4 words.
 * rec may be NULL (count only); at most rec_capacity records are written.  Outputs: the number of clusters,
 * the neighbour rows staged over all clusters, the largest padded edge count of a cluster (a caller may re-pack the
 * records with a smaller LW).  Returns GTS_ERR_SHAPE when a row's degree exceeds max_srcs / max_edges or a tag
 * does not fit a byte, GTS_ERR_ARGKIND for limits outside the ranges above or records above 512 words.
 * gts_cluster_lds_bytes: LDS bytes of one ring slot for (max_rows, max_srcs, LW), kind 0 = forward, 1 =
 * backward; the kernels need at least two slots in the CU's 160 KiB.
 * The device entry points take a DEVICE copy of the records (stride W for the LW passed); n_feat must be 256,
 * arg_bytes 0 / 1 (forward) or 1 (backward), n_rows * 1024 < 2^32. */
int64_t gts_cluster_record_words(int32_t max_rows, int32_t max_srcs, int32_t loc_words, int32_t with_tag);
int32_t gts_cluster_schedule(const int32_t* indptr, const int32_t* indices, const int32_t* t_indptr,
                             const int32_t* t_indices, const int32_t* edge_tag, int64_t n_rows,
                             int32_t max_rows, int32_t max_srcs, int32_t max_edges, int32_t* rec,
                             int64_t rec_capacity, int64_t* n_clusters, int64_t* n_staged,
                             int32_t* max_cluster_edges);
int64_t gts_cluster_lds_bytes(int32_t max_rows, int32_t max_srcs, int32_t loc_words, int32_t kind);
int32_t gts_spmm_max_fwd_cluster_f32(const int32_t* rec, int64_t n_clusters, int32_t max_rows, int32_t max_srcs,
                                     int32_t loc_words, const float* x, float* out, void* arg,
                                     int32_t arg_bytes, int32_t relu_input, int64_t n_rows, int64_t n_feat,
                                     uint32_t* counters, void* stream);
int32_t gts_spmm_max_bwd_cluster_f32(const int32_t* rec, int64_t n_clusters, int32_t max_rows, int32_t max_srcs,
                                     int32_t loc_words, const float* gout, const void* arg, int32_t arg_bytes,
                                     float* gx, int64_t n_rows, int64_t n_feat, uint32_t* counters, void* stream);
/* `counters`: GTS_CLUSTER_COUNTER_WORDS words of caller scratch, ZERO on entry; a launch that uses them leaves them zero (the last workgroup
 * of each XCD to finish resets them), so one zeroed buffer per stream serves every launch on it.  On launches of 16 and more units per
 * workgroup (GTS_OPT_CLUSTER_DEALING) the persistent workgroups of an XCD take their units off one counter, which keeps the units in flight
 * neighbours in the walk whatever each workgroup's pace — their halo rows then meet in the XCD's L2 (profiles/r04/gat_l2_replay.log,
 * k12_dealing_ab.log).  NULL: static round-robin dealing (same values, more re-fetched rows at streaming sizes). */
#define GTS_CLUSTER_COUNTER_WORDS 256
/* 1 when a launch over these records would deal its units off `counters` (so that a caller who carves them out of a scratch block knows whether
 * they need zeroing at all), 0 when it deals statically */
int32_t gts_cluster_uses_counters(int64_t n_clusters, int32_t max_rows, int32_t max_srcs, int32_t loc_words, int32_t backward);

/* ---- K3/K4: copy_u + sum / mean / gcn reducers (forward and backward) ---------------
 * Replaces DGL update_all(copy_u, sum|mean) of SAGEConv('mean'|'gcn')
 * (model/networks.py:73,75 via :25-30) and, on the out-CSR, their autograd.
 *   s[v,f]   = sum_k  x[idx_k, f] / (div_in ? div_in[idx_k] : 1)   (slot order)
 *   if add_self: s[v,f] += x[v,f] / (div_in ? div_in[v] : 1)        (needs square graph)
 *   out[v,f] = s[v,f] / (div_out ? div_out[v] : 1)   (+ accum[v,f] when accum is given: the gradient
 *              that reached row v by another path, e.g. through fc_self beside the neighbour term)
 * mean fwd: div_out = max(deg,1); mean bwd: out-CSR, div_in = max(deg,1);
 * gcn  fwd: add_self, div_out = deg+1;  gcn bwd: out-CSR, add_self, div_in = deg+1. */
int32_t gts_spmm_sum_f32(const int32_t* indptr, const int32_t* indices, const float* x,
                         float* out, const float* div_in, const float* div_out, const float* accum,
                         int32_t add_self, int64_t n_out, int64_t n_feat, void* stream);

/* ---- K5-K7: GATConv attention + aggregation (forward) -------------------------------
 * Replaces apply_edges(u_add_v) + leaky_relu + edge_softmax + update_all(u_mul_e, sum)
 * inside GATConv, reached from model/networks.py:46,52,56, plus the layer's tail
 * (+ res_fc(h) + bias, activation) on the way out.
 *   e_k = leaky_relu(el[src_k,h] + er[v,h]);  a_k = softmax_k(e_k) over row v (max-subtracted)
 *   out[v,h,:] = act( sum_k a_k * ft[src_k,h,:] + residual[v,h,:] + bias[h,:] )
 *   ft [n,H,D], el/er [n,H], out [n,H,D], attn [E,H] in in-CSR slot order (saved for K8);
 *   bias [H*D] and residual [n,H,D] optional; activation 0 = none, 1 = ELU (alpha 1). */
int32_t gts_gat_fwd_f32(const int32_t* indptr, const int32_t* indices, const float* ft,
                        const float* el, const float* er, float negative_slope, const float* bias,
                        const float* residual, int32_t activation, float* out, float* attn,
                        int64_t n, int64_t heads, int64_t dim, void* stream);
/* el[n,h] = <ft[n,h,:], attn_l[h,:]>, er[n,h] = <ft[n,h,:], attn_r[h,:]>  (GATConv: (feat * attn).sum(-1)) */
int32_t gts_gat_scores_f32(const float* ft, const float* attn_l, const float* attn_r, float* el,
                           float* er, int64_t n, int64_t heads, int64_t dim, void* stream);

/* ---- K8: GATConv backward ------------------------------------------------------------
 * Atomics-free passes:
 *  (0) gts_gat_act_bwd_f32: g_pre = gout * act'(out) (ELU / ReLU through its output; activation 0 leaves
 *      gout as is and g_pre may be NULL) and g_bias[c] = sum_n g_pre[n,c] (optional), cols = H*D.
 *  (1) per destination row (in-CSR):  ga_k = <g_pre[v,h,:], ft[src_k,h,:]>,
 *      ge_k = a_k*(ga_k - sum_j a_j ga_j) * leaky'(el[src_k]+er[v]);  ger[v,h] = sum_k ge_k;
 *      ge [E,H] written in in-CSR slot order.
 *  (2) per source row (out-CSR): gft[u,h,:] = sum_{e in out(u)} a[pos(e)] * g_pre[dst_e,h,:]
 *      (+ gel[u,h]*attn_l[h,:] + ger[u,h]*attn_r[h,:] when attn_l/attn_r/ger are given: the
 *      gradient of the score dot products), gel[u,h] = sum_e ge[pos(e)].
 *      t_pos[e] = absolute in-CSR position of out-edge e.
 *  (3) gts_gat_param_grad_f32: g_attn_l[h,:] = sum_n gel[n,h] ft[n,h,:], g_attn_r with ger.
 * (0) and (3) reduce over the node axis through `workspace`
 * (>= gts_gat_reduce_workspace(n, H*D) bytes) in a fixed order. */
int32_t gts_gat_bwd_edge_f32(const int32_t* indptr, const int32_t* indices, const float* ft,
                             const float* el, const float* er, const float* attn,
                             const float* gout, float negative_slope, float* ge, float* ger,
                             int64_t n, int64_t heads, int64_t dim, void* stream);
int32_t gts_gat_bwd_src_f32(const int32_t* t_indptr, const int32_t* t_indices,
                            const int32_t* t_pos, const float* attn, const float* ge,
                            const float* gout, const float* attn_l, const float* attn_r,
                            const float* ger, float* gft, float* gel, int64_t n, int64_t heads,
                            int64_t dim, void* stream);
int64_t gts_gat_reduce_workspace(int64_t n, int64_t cols);
int32_t gts_gat_act_bwd_f32(const float* gout, const float* out, int32_t activation, float* g_pre,
                            float* g_bias, float* workspace, int64_t workspace_bytes, int64_t n,
                            int64_t cols, void* stream);
int32_t gts_gat_param_grad_f32(const float* ft, const float* gel, const float* ger, float* g_attn_l,
                               float* g_attn_r, float* workspace, int64_t workspace_bytes, int64_t n,
                               int64_t heads, int64_t dim, void* stream);

/* ---- K5-K8 at D = 256 over a cluster row schedule (LDS-staged neighbour tiles; csrc/gts_gat_cluster.hip) -----------------
 * The same GATConv aggregation (model/networks.py:46,52,56 -> dgl GATConv: edge_softmax, u_mul_e + sum) and the source pass
 * of its backward, computed per cluster of a schedule made by gts_cluster_schedule (records `rec`, limits max_rows / max_srcs,
 * loc_words, tagged = the records carry the tag section): a unit of work is one 512-byte column half of one head of one
 * cluster, its distinct neighbour slices staged once in LDS.  Same values as gts_gat_fwd_f32 / gts_gat_bwd_src_f32 bit for
 * bit (edges in CSR slot order inside every row, the softmax in the plain kernel's association); no residual term here.
 *   gts_gat_attn_f32: attn [E, H] alone (max_degree: the largest in-degree, <= 64).
 *   forward: in-CSR (indptr, indices) + a schedule of the in-CSR;  bias [H*256] optional.
 *   backward: out-CSR (t_indptr, t_pos) + a schedule of the out-CSR;  attn_lr = attn_l | attn_r stacked [2, H*256] (with ger)
 *             or both null; gel [n, H] is written as by gts_gat_bwd_src_f32.  max_degree: the largest out-degree (or -1:
 *             unknown); with rows of at most one 8-edge chunk (<= 8) the weight pass takes a faster form, same values.
 * workspace >= gts_gat_cluster_workspace(n_clusters, max_rows, loc_words, heads, backward) bytes. */
int64_t gts_gat_cluster_workspace(int64_t n_clusters, int32_t max_rows, int32_t loc_words, int64_t heads, int32_t backward);
int32_t gts_gat_attn_f32(const int32_t* indptr, const int32_t* indices, const float* el, const float* er,
                         float negative_slope, float* attn, int64_t n, int64_t heads, int64_t max_degree, void* stream);
int32_t gts_gat_fwd_cluster_f32(const int32_t* indptr, const int32_t* indices, const int32_t* rec, int64_t n_clusters,
                                int32_t max_rows, int32_t max_srcs, int32_t loc_words, int32_t tagged, const float* ft,
                                const float* el, const float* er, float negative_slope, const float* bias,
                                int32_t activation, float* out, float* attn, float* workspace, int64_t workspace_bytes,
                                int64_t n, int64_t heads, int64_t dim, int64_t max_degree, void* stream);
/* edge pass (step 1 of gts_gat_bwd_edge_f32's job, same ge [E, H] / ger [n, H] bit for bit): in-CSR + a schedule of the in-CSR whose
 * images also hold the rows' own gradient slices; rows of one 8-edge chunk only (max_degree <= 8, else GTS_ERR_SHAPE).
 * workspace >= 2 * gts_gat_cluster_workspace(n_clusters, max_rows, loc_words, heads, 0) bytes (the two half dot products). */
int32_t gts_gat_bwd_edge_cluster_f32(const int32_t* indptr, const int32_t* indices, const int32_t* rec, int64_t n_clusters,
                                     int32_t max_rows, int32_t max_srcs, int32_t loc_words, int32_t tagged, const float* ft,
                                     const float* el, const float* er, const float* attn, const float* gout,
                                     float negative_slope, float* ge, float* ger, float* workspace, int64_t workspace_bytes,
                                     int64_t n, int64_t heads, int64_t dim, int64_t max_degree, void* stream);
int32_t gts_gat_bwd_src_cluster_f32(const int32_t* t_indptr, const int32_t* t_pos, const int32_t* rec, int64_t n_clusters,
                                    int32_t max_rows, int32_t max_srcs, int32_t loc_words, int32_t tagged,
                                    const float* attn, const float* ge, const float* gout, const float* attn_lr,
                                    const float* ger, float* gft, float* gel, float* workspace, int64_t workspace_bytes,
                                    int64_t n, int64_t heads, int64_t dim, int64_t max_degree, void* stream);

/* ---- K12: node -> voxel projection ----------------------------------------------------
 * Replaces data_processing/graph_io.py:21-24 (project_nodes_to_img) and
 * scripts/generate_gnn_predictions.py:55-61 (save_voxel_logits):
 *   out[i, :] = (svs[i] < 0 ? bg_row : table[svs[i], :]) for i in [0, n_vox).
 * svs holds supervoxel ids in [-1, n_rows) (int16, mri2graph/graphgen.py:77); any
 * negative id selects the background row, exactly numpy's table_plus_bg[-1].
 * Rows are `row_bytes` opaque bytes (4, 8 or 16): int64 labels, fp32 x4 logits, ... */
int32_t gts_project_rows_i16(const int16_t* svs, const void* table, const void* bg_row,
                             void* out, int64_t n_vox, int64_t n_rows, int32_t row_bytes,
                             void* stream);
/* Fused argmax + projection: out[i] = argmax_c logits[svs[i], c] (first maximum, as
 * torch.max(logits,1), generate_gnn_predictions.py:66), 0 for background; int16 out
 * with an optional 4-entry relabel table (swap_labels_to_brats, preprocess_dataset.py:159). */
int32_t gts_project_argmax_i16(const int16_t* svs, const float* logits, const int16_t* relabel,
                               int16_t* out, int64_t n_vox, int64_t n_rows, int64_t n_classes,
                               void* stream);

/* gts_project_argmax_i16 over a [dim_x, dim_y, dim_z] partitioning that also reports which
 * planes hold tumour: occupancy is a caller-zeroed uint8[dim_x + dim_y + dim_z]; byte x,
 * dim_x + y, dim_x + dim_y + z is set to 1 when some voxel of that plane gets a non-zero
 * label.  These are `mask.any(axis=...)` of determine_tumor_crop (data_processing/
 * image_processing.py:8-17, reached from scripts/generate_joint_predictions.py:66) before its
 * one-voxel dilation, which the host applies to the three vectors. */
int32_t gts_project_argmax_occupancy_i16(const int16_t* svs, const float* logits, int16_t* out,
                                         uint8_t* occupancy, int64_t dim_x, int64_t dim_y,
                                         int64_t dim_z, int64_t n_rows, int64_t n_classes,
                                         void* stream);

/* ---- K16/K17: glue between the GNN and the refinement CNN -----------------------------------
 * Replace the tensor indexing of predict_one_sample (scripts/generate_joint_predictions.py:
 * 59-73) and combine_logits_and_image (model/cnn_model.py:85-88).  A crop is the outer product
 * of three ascending index vectors xs [cx], ys [cy], zs [cz] (np.ix_ of boolean plane masks),
 * int32 on the device, into a volume whose two inner extents are dim_y, dim_z.
 *
 * crop_concat: out[c, i, j, k] (fp32, [img_channels + row_channels, cx, cy, cz] contiguous)
 *   = img[xs[i], ys[j], zs[k], c]                         for c <  img_channels
 *   = table_plus_bg[svs[xs[i], ys[j], zs[k]]][c - img_channels]   otherwise,
 *   table [n_rows, row_channels] node logits, bg_row the background logits (id -1), img
 *   [X, Y, Z, img_channels] channels-last.  Exact copies. */
int32_t gts_crop_concat_f32(const float* img, const int16_t* svs, const float* table,
                            const float* bg_row, const int32_t* xs, const int32_t* ys,
                            const int32_t* zs, float* out, int64_t cx, int64_t cy, int64_t cz,
                            int64_t dim_y, int64_t dim_z, int64_t n_rows, int64_t img_channels,
                            int64_t row_channels, void* stream);
/* argmax_scatter: out[xs[i], ys[j], zs[k]] = argmax_c scores[c, i, j, k] (first maximum, as
 * torch.argmax; optional relabel table), scores [n_classes, cx, cy, cz]; the other voxels of the
 * caller-zeroed int16 volume `out` are left alone. */
int32_t gts_argmax_scatter_i16(const float* scores, const int16_t* relabel, const int32_t* xs,
                               const int32_t* ys, const int32_t* zs, int16_t* out, int64_t cx,
                               int64_t cy, int64_t cz, int64_t dim_y, int64_t dim_z,
                               int64_t n_classes, void* stream);

/* ---- K14: AdamW step over one flat fp32 buffer ---------------------------------------------
 * torch.optim.AdamW(net.parameters(), lr, weight_decay).step() (model/gnn_model.py:28,46), same
 * update rule (decoupled decay, bias-corrected moments, no amsgrad), for parameters, gradients
 * and both moment buffers laid out as ONE contiguous fp32 range each.  `step` is the 1-based
 * count of this update (for the bias corrections).  In-place on param / exp_avg / exp_avg_sq. */
int32_t gts_adamw_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                      double lr, double beta1, double beta2, double eps, double weight_decay,
                      int64_t step, void* stream);

/* ---- K15: label coincidence counts for the Dice metrics -------------------------------------
 * Replaces the numpy mask arithmetic of model/evaluation.py:24-46 (count_node_labels,
 * calculate_node_dices), :64-79 (voxel Dice in calculate_brats_metrics) and :98-106
 * (calculate_dice_from_logical_array) reached from GNN.evaluate, model/gnn_model.py:76-87.
 * counts[5*cp + ct] += #{ i : class(pred[i]) == cp and class(truth[i]) == ct }, where
 * class(v) = v for 0..3 and 4 for any other value.  counts is a caller-owned int64[25] on the
 * device that the call ADDS to (zero it first); workspace is caller-owned device scratch of at
 * least gts_label_confusion_workspace(n) bytes.  Integer sums: exact, independent of scheduling. */
int64_t gts_label_confusion_workspace(int64_t n);
int32_t gts_label_confusion_i16(const int16_t* pred, const int16_t* truth, int64_t* counts,
                                void* workspace, int64_t workspace_bytes, int64_t n, void* stream);

/* ---- K11: dense fp32 layer GEMMs on the matrix cores -------------------------------------
 * Replace the nn.Linear calls inside DGL's SAGEConv / GATConv (fc_pool, fc_self + fc_neigh,
 * fc; reached from model/networks.py:25,28,30,46,52,56) and their autograd.  Exact fp32
 * (v_mfma_f32_32x32x2_f32).  All dims that index a contiguous axis must be multiples of 4.
 *
 * forward:  out[m, n] = act( sum_k a0[m,k] w0[n,k] + (a1 ? sum_k a1[m,k] w1[n,k] : 0) + bias[n] )
 *   a0 [M,K0], w0 [N,K0], a1 [M,K1], w1 [N,K1] row-major (torch Linear layout); bias optional;
 *   relu != 0 applies max(., 0).  The pair form is fc_self(h) + fc_neigh(m) in one pass.
 *   relu_bits (optional, n % 64 == 0, gts_relu_bits_bytes(m, n) bytes): out > 0 as one bit per element, written
 *   while out is stored — the ReLU mask the input-gradient calls below can read instead of the floats of `out`
 *   (1/32 of the bytes, and scalar loads that do not queue behind the epilogue's stores).  Layout: 64-bit word
 *   ((col / 64) * ceil(m / 4) + row / 4) * 4 + e, bit 16 * (row % 4) + (col % 64) / 4, holds
 *   out[row][64 * (col / 64) + 4 * ((col % 64) / 4) + e] > 0  (e = 0..3): one word = one wave-wide comparison of the
 *   epilogue that stores four rows x 64 columns as 16-byte pieces, and the words of a wave's 80 rows are consecutive.
 *   Bits of rows >= m are unspecified. */
int64_t gts_relu_bits_bytes(int64_t m, int64_t n);
/* 1 when the launches for an [m, n] output would write / read these bits inside their epilogues (the panel kernels of
 * the tall shapes); 0 when asking for them would only add a pass (callers then pass no relu_bits at all). */
int32_t gts_relu_bits_pay(int64_t m, int64_t n);
int32_t gts_linear_fwd_f32(const float* a0, const float* w0, const float* a1, const float* w1,
                           const float* bias, float* out, int64_t m, int64_t n, int64_t k0,
                           int64_t k1, int32_t relu, uint64_t* relu_bits, const float* const* packed,
                           void* stream);
/* Weights in FRAGMENT ORDER — the optional `packed` argument of gts_linear_fwd_f32, gts_linear_fwd_chain_f32,
 * gts_linear_bwd_input_t_f32 and gts_linear_bwd_input_chain_t_f32: a HOST array with one device pointer per weight
 * operand of the call, in argument order (w0, w1[, w2] resp. w0t, w1t[, w2t]; NULL entries and a NULL array are
 * allowed), each a copy of that operand made by gts_pack_weights_f32.  The nn.Linear weights inside SAGEConv
 * (model/networks.py:25-30) are read by every workgroup of a layer GEMM; as torch stores them ([out, in] row-major) a
 * wave's 16-row MFMA fragment is 64 pieces of 16 bytes 1 KiB apart — 64 accesses of the vector L1, whose access rate
 * bounds the tall panel kernels (profiles/r04/panel144_l1_bound.log) — in fragment order it is 1 KiB of consecutive
 * bytes (16 accesses).  Same values, same reduction order: results are bit-identical with and without the copies;
 * kernels other than the panel kernels ignore them and read the plain operand.
 *   B [n, k] = the operand as the GEMM reads it (rows = output columns, columns = reduction); G = ceil(k / 16):
 *   packed[((T * G + g) * 64 + lane) * 4 + e] = B[16 T + (lane & 15)][16 g + 4 (lane >> 4) + e], 0 past the edges;
 *   gts_packed_weight_floats(n, k) = ceil(n / 16) * G * 256 floats.
 * gts_pack_weights_f32: n_mats matrices src[q] [rows, cols] of one shape (HOST arrays of device pointers), one launch
 * per 32.  transposed = 0: B = src[q];  transposed = 1: B = src[q]^T (the input gradient's operand), and plain_t
 * (optional, only then) also receives src[q]^T row-major [cols, rows] — what gts_transpose_batch_f32 writes. */
int64_t gts_packed_weight_floats(int64_t n, int64_t k);
int32_t gts_pack_weights_f32(const float* const* src, float* const* dst, float* const* plain_t, int32_t n_mats,
                             int64_t rows, int64_t cols, int32_t transposed, void* stream);
/* input gradient:  gin[m, k] = sum_n g0[m,n] w0[n,k] + (g1 ? sum_n g1[m,n] w1[n,k] : 0)
 *   g0 [M,N0], w0 [N0,K], g1 [M,N1], w1 [N1,K].  relu_mask (optional, [M,K]): gin is zeroed
 *   where relu_mask <= 0, i.e. the ReLU backward of the layer that produced this input. */
int32_t gts_linear_bwd_input_f32(const float* g0, const float* w0, const float* g1, const float* w1,
                                 const float* relu_mask, float* gin, int64_t m, int64_t k,
                                 int64_t n0, int64_t n1, void* stream);
/* the same input gradient from TRANSPOSED weights (w0t [K,N0], w1t [K,N1], e.g. made once per
 * backward pass by gts_transpose_batch_f32): both GEMM operands are then reduction-contiguous, the
 * forward kernel's form — bitwise the same sums in the same order as gts_linear_bwd_input_f32.
 * relu_bits (optional, k % 64 == 0; relu_mask must be given too): relu_mask > 0 in the bit layout above, as written by
 * the forward call that produced relu_mask; the kernels that can use the bits never touch relu_mask, the others read
 * its floats as before — the result is the same either way. */
int32_t gts_linear_bwd_input_t_f32(const float* g0, const float* w0t, const float* g1, const float* w1t,
                                   const float* relu_mask, const uint64_t* relu_bits, float* gin, int64_t m,
                                   int64_t k, int64_t n0, int64_t n1, const float* const* packed, void* stream);
/* the same input gradient carried through the activation of the layer BELOW (a GATConv stack: the next layer's input IS the
 * ELU output of this one, model/networks.py:46-58,61-63), with that layer's bias gradient:
 *   gin[m, k] = (g0 w0 (+ g1 w1)) * act'(act_out),  act' through the activation's OUTPUT act_out [m, k]: activation 1 = ELU
 *   (1 where act_out > 0, act_out + 1 elsewhere), 2 = ReLU;  g_bias[k] = sum_m gin[m, k] (optional).
 * Tall operands with k % 256 == 0 and ELU: both ride in the GEMM's epilogue (no pass over gin, column sums per 80-row block
 * in `workspace` and added in fixed order); otherwise gts_linear_bwd_input_t_f32 followed by gts_gat_act_bwd_f32 in place.
 * gin is bitwise the same either way; g_bias is summed in a different (each time fixed) order.
 * workspace >= gts_linear_bwd_input_t_act_workspace(m, k) bytes when g_bias is asked for. */
int64_t gts_linear_bwd_input_t_act_workspace(int64_t m, int64_t k);
int32_t gts_linear_bwd_input_t_act_f32(const float* g0, const float* w0t, const float* g1, const float* w1t,
                                       const float* act_out, int32_t activation, float* gin, float* g_bias,
                                       float* workspace, int64_t workspace_bytes, int64_t m, int64_t k,
                                       int64_t n0, int64_t n1, const float* const* packed,
                                       void* stream);
/* GATConv's projection with its attention scores (model/networks.py:46,52,56 -> dgl GATConv: feat_src = fc(h).view(N, H, D);
 * el = (feat_src * attn_l).sum(-1); er likewise):  ft [m, heads*dim] = h [m, k] w_fc^T,  el/er [m, heads] = <ft[n,h,:], attn_l/r[h,:]>.
 * Tall operands with dim % 64 == 0: the dot products ride in the GEMM epilogue (partials per 64 columns in `workspace`,
 * >= gts_gat_fc_scores_workspace bytes, summed in fixed order); otherwise the GEMM is followed by gts_gat_scores_f32. */
int64_t gts_gat_fc_scores_workspace(int64_t m, int64_t heads, int64_t dim);
int32_t gts_gat_fc_scores_f32(const float* h, const float* w_fc, const float* attn_l, const float* attn_r,
                              float* ft, float* el, float* er, float* workspace, int64_t workspace_bytes,
                              int64_t m, int64_t heads, int64_t dim, int64_t k, const float* w_fc_packed, void* stream);
/* two GEMMs of a layer chain in one call (one launch when the operands are tall and <= 256 wide, otherwise two):
 *   out  [m, n]  = act (a0 w0^T (+ a1 w1^T) + bias)          exactly gts_linear_fwd_f32
 *   out2 [m, n2] = act2(out w2^T + bias2)                      exactly gts_linear_fwd_f32 on `out`
 * (SAGEConv-pool stack: fc_self + fc_neigh of layer L, then fc_pool of layer L+1 on the rows just produced).
 * The input-gradient form chains  gin = (g0 w0 (+ g1 w1)) (. mask)  and  gin2 = gin w2  on transposed weights
 * (w0t [k, n0], w1t [k, n1], w2t [k2, k]).  n (resp. k) must be a multiple of 4.  relu_bits: the mask bits of
 * `out` (written) resp. of relu_mask (read), as in gts_linear_fwd_f32 / gts_linear_bwd_input_t_f32. */
int32_t gts_linear_fwd_chain_f32(const float* a0, const float* w0, const float* a1, const float* w1,
                                 const float* bias, float* out, const float* w2, const float* bias2,
                                 float* out2, int64_t m, int64_t n, int64_t k0, int64_t k1, int32_t relu,
                                 int64_t n2, int32_t relu2, uint64_t* relu_bits, const float* const* packed,
                                 void* stream);
int32_t gts_linear_bwd_input_chain_t_f32(const float* g0, const float* w0t, const float* g1,
                                         const float* w1t, const float* relu_mask, const uint64_t* relu_bits,
                                         float* gin, const float* w2t, float* gin2, int64_t m, int64_t k,
                                         int64_t n0, int64_t n1, int64_t k2, const float* const* packed,
                                         void* stream);
/* dst[q][c, r] = src[q][r, c] for n_mats matrices of one shape [rows, cols] in one launch per 32
 * (src, dst: HOST arrays of device pointers).  Serves the transposed-weight form above: the
 * weights of a layer stack (torch Linear layout [out, in]) are turned once per backward pass. */
int32_t gts_transpose_batch_f32(const float* const* src, float* const* dst, int32_t n_mats,
                                int64_t rows, int64_t cols, void* stream);
/* weight gradients of n_problems (1..32) same-shape problems in ONE launch (the three weight
 * gradients of a SAGE layer, or those of a whole stack of equal layers):  gw[q][n, k] = sum_m g[q][m,n] a[q][m,k];
 * gb[q][n] = sum_m g[q][m,n] where gb && gb[q].  g, a, gw, gb are HOST arrays of device pointers.
 *   The reduction over the M nodes is split over workgroups into slabs in `workspace`
 *   (>= gts_linear_bwd_weight_workspace(m,n,k,n_problems) bytes, caller-owned scratch) that a
 *   second kernel adds in a fixed order: bitwise reproducible, no float atomics. */
int64_t gts_linear_bwd_weight_workspace(int64_t m, int64_t n, int64_t k, int32_t n_problems);
int32_t gts_linear_bwd_weight_f32(const float* const* g, const float* const* a, float* const* gw,
                                  float* const* gb, int32_t n_problems, float* workspace,
                                  int64_t workspace_bytes, int64_t m, int64_t n, int64_t k,
                                  void* stream);

/* ---- a whole stack of SAGEConv('pool') layers in one call each way -------------------------------
 * Replaces the layer loop of GraphSage.forward (model/networks.py:32-36: `for layer in self.layers: h = layer(graph, h)`)
 * over SAGEConv('pool') layers (model/networks.py:25,28,30: ReLU on all but the last, bias on) and its autograd.  HOST
 * orchestration: the calls enqueue this library's own kernels (K1 / K2, K11 with their chained, transposed and batched
 * forms) in a fixed order on `stream`; nothing is allocated.
 *   widths[0] = in_feats, widths[i + 1] = output width of layer i (all multiples of 4); n_layers <= 64.
 *   params[5 i .. 5 i + 4] = fc_pool.weight [w_i, w_i], fc_pool.bias [w_i], fc_self.weight [w_{i+1}, w_i],
 *                            fc_neigh.weight [w_{i+1}, w_i], bias [w_{i+1}]                       (device pointers)
 *   sched_*: optional cluster row schedule (gts_cluster_schedule) of the in-CSR (forward) / of the out-CSR tagged with
 *            t_slot (backward); NULL = plain K1 / K2.  Used by the 256-wide layers when arg_bytes allows.
 *   flags: 1 = chain consecutive GEMMs into one launch, 2 = record / read ReLU masks as bits, 4 = input gradients on
 *          transposed weights (7 = what the layer-by-layer path does by default).
 * Forward: everything it produces lives in `arena` (>= gts_sage_pool_stack_fwd_arena(...) bytes, which also reports the
 * byte offsets: offsets[4 i .. 4 i + 3] = max-pooled features m_i [n, w_i], winners arg_i [n, w_i] (arg_bytes each; -1
 * when not training), output out_i [n, w_{i+1}] (the next layer's input; the last one holds the logits), ReLU bits of out_i
 * (-1 when absent); offsets[4 L], [4 L + 1] = two scratch buffers).  training = 0: no winners, no bits.
 * Backward: `fwd_arena` is the arena of a training forward with the same arguments; grads[5 i .. 5 i + 4] receive the
 * gradients of params[5 i .. 5 i + 4] (any caller-chosen destinations, e.g. slices of one flat buffer); gx [n, w_0]
 * optional; scratch >= gts_sage_pool_stack_bwd_scratch(...) bytes. */
int64_t gts_sage_pool_stack_fwd_arena(int64_t n_rows, const int64_t* widths, int32_t n_layers, int32_t training,
                                      int32_t arg_bytes, int32_t flags, int64_t* offsets);
int32_t gts_sage_pool_stack_fwd_f32(const int32_t* indptr, const int32_t* indices, const int32_t* sched_rec,
                                    int64_t sched_clusters, int32_t sched_rows, int32_t sched_srcs,
                                    int32_t sched_loc_words, const float* x, const float* const* params,
                                    int64_t n_rows, const int64_t* widths, int32_t n_layers, int32_t training,
                                    int32_t arg_bytes, int32_t flags, void* arena, int64_t arena_bytes, void* stream);
int64_t gts_sage_pool_stack_bwd_scratch(int64_t n_rows, const int64_t* widths, int32_t n_layers, int32_t flags);
int32_t gts_sage_pool_stack_bwd_f32(const int32_t* t_indptr, const int32_t* t_indices, const int32_t* t_slot,
                                    const int32_t* sched_rec, int64_t sched_clusters, int32_t sched_rows,
                                    int32_t sched_srcs, int32_t sched_loc_words, const float* gout, const float* x,
                                    const float* const* params, int64_t n_rows, const int64_t* widths,
                                    int32_t n_layers, int32_t arg_bytes, int32_t flags, const void* fwd_arena,
                                    float* const* grads, float* gx, void* scratch, int64_t scratch_bytes,
                                    void* stream);

/* ---- class-weighted cross-entropy ----------------------------------------------------------
 * Replaces torch.nn.CrossEntropyLoss(weight=class_weights)(logits, labels) of the reference
 * harness (model/gnn_model.py:30,42) and, through grad_unscaled, its backward.
 *   out3 = { num = sum_i w[y_i] nll_i,  den = sum_i w[y_i],  num / den }
 *   grad_unscaled[i,c] = w[y_i] (softmax(x_i)_c - [c == y_i])   (optional; d loss/dx = that / den)
 * logits [n, n_classes] fp32, labels int64 in [0, n_classes) (no ignore_index), class_w optional.
 * Deterministic two-level reduction through `workspace`
 * (>= gts_weighted_ce_workspace(n) bytes).  n_classes <= 32. */
int64_t gts_weighted_ce_workspace(int64_t n);
int32_t gts_weighted_ce_f32(const float* logits, const int64_t* labels, const float* class_w,
                            float* grad_unscaled, float* workspace, int64_t workspace_bytes,
                            float* out3, int64_t n, int64_t n_classes, void* stream);

/* ---- tuning knobs -----------------------------------------------------------------------
 * Process-wide tile selection of the K11 kernels (defaults are the tuned values; used by
 * tools/tune_gemm.py).  Returns GTS_ERR_ARGKIND for an unknown option. */
#define GTS_OPT_GEMM_TILE 1  /* forward tile: -1 automatic, -2 automatic among the 32x32x2 tiles only (a row's result then does
                                not depend on how many rows the call has: batched == per-sample, bit for bit), 1 = 128x256, 3 = 64x256, 5 = 256x128, 8 = 256x256 double-buffered, 10 = 240 / 192 / 144-row panels (16x16x4 MFMA) with direct-to-fragment buffer loads (12 waves).  Other values: GTS_ERR_ARGKIND (the forms measured and rejected — panels staged through LDS, four waves of 240x64 — are built from tools/diag/, not shipped) */
#define GTS_OPT_WGRAD_TILE 2 /* weight-gradient tile: -1 automatic, 1 = 128x128, 2 = 128x256, 4 = 256x256 double-buffered through registers and LDS,
                                6 = the same tile moved by LDS-DMA with a main loop of MFMAs and LDS reads only (immediate-offset fragment
                                    reads, scalar-built DMA descriptors, bias sums in one wave per SIMD): the automatic choice where 4 was
                                    (same bits as 4).  Other values: GTS_ERR_ARGKIND (rejected tiles: tools/diag/gemm_rejected_forms.inc) */
#define GTS_OPT_IGRAD_TILE 3 /* input-gradient tile: same numbering as the forward one (default 1) */
#define GTS_OPT_PROJECT_STREAMING 6  /* K12: non-temporal stores of the projected rows (default 1) */
#define GTS_OPT_SPMM_ROWS_PER_WAVE 4 /* K1-K4: rows one wave walks (0 = automatic) */
#define GTS_OPT_SPMM_STREAMING 5     /* K1/K2: bit 0 = non-temporal stores of write-once rows, bit 1 = non-temporal
                                        loads of read-once rows (K2's relu_src); -1 = per-kernel default */
#define GTS_OPT_GEMM_SCHED 7  /* K11 scheduling bits (default 1): 1 = s_setprio by progress inside a reduction tile (LDS kernels),
                                2 = non-temporal stores of the output panel (direct-to-fragment kernels; no gain measured),
                                4 = the direct-to-fragment kernels keep every epilogue switch a run-time argument (the generic
                                    instantiation) instead of the compile-time epilogues of the layer-stack launches (A/B runs),
                                8 = the chained 240-row launches of the layer stack issue the nine fragment loads of a reduction
                                    group in one burst (rounds 1 - 2) instead of one load per eight MFMAs (A/B runs; same bits) */
#define GTS_OPT_CLUSTER_STREAMING 8 /* clustered K1 / K2: bit 0 = non-temporal stores of out / gx; -1 = per-kernel default */
#define GTS_OPT_GAT_WALK 14          /* K5-K8: 1 = walk the (node, head) rows head-major (default), 0 = node-major */
#define GTS_OPT_GAT_CLUSTER_WAVES 15 /* clustered GAT aggregation: waves per persistent workgroup (0 = default 12; up to 16) */
#define GTS_OPT_GAT_CLUSTER_GROUP 16 /* clustered GAT aggregation UNDER STATIC DEALING (option 17 = 1): clusters of an XCD's span walked together through
                                        all their (head, half) slices (0 = 16; 100000 = the whole span).  Dealt units walk the whole span slice by slice */
#define GTS_OPT_GAT_CLUSTER_DEALING 17 /* clustered GAT aggregation: 0 = the workgroups of an XCD take their units off one counter (default: the units in
                                         flight stay neighbours in the walk whatever each workgroup's pace), 1 = static round-robin (A/B runs) */
#define GTS_OPT_CLUSTER_DEALING 18  /* clustered K1 / K2, where the caller gives counters: 0 = automatic (units dealt off the XCD's counter from 16 units per
                                       workgroup on, static round-robin below), 1 = always dealt, 2 = always static */
#define GTS_OPT_PANEL_ROWS 13        /* K11 direct-to-fragment panels: rows per panel, 0 = automatic among 240 / 192 / 144 */
#define GTS_OPT_CLUSTER_RING 10      /* clustered K1 / K2: units the gathers run ahead of the reduction (0 = automatic: 1; 2 where the LDS holds it) */
#define GTS_OPT_CLUSTER_PER_CU 11    /* clustered K1 / K2: persistent workgroups per CU (0 = automatic: 2, or 3 for short backward launches) */
#define GTS_OPT_CLUSTER_CONSUMERS 12 /* clustered K1 / K2: waves per workgroup (0 = automatic: 16 forward, 12 backward) */
int32_t gts_set_option(int32_t option, int32_t value);
/* Current value of a knob (INT32_MIN for an unknown option): callers that change one temporarily put it back. */
int32_t gts_get_option(int32_t option);

/* ---- K9h: batch collate on the host, straight into the upload layout ---------------------------------
 * HOST functions (host pointers, no GPU call, no allocation, no state; safe to call from any thread — a
 * ctypes caller holds no interpreter lock while they run).
 * Replace, for the training loader's hot path, what the reference does per step in
 * data_processing/data_loader.py:165-169 (`minibatch_graphs`: dgl.batch + np.concatenate of the
 * members' features and labels + FloatTensor / LongTensor conversions) and model/gnn_model.py:37-40
 * (`.to(device)` of graph, features, labels): the members' cached host arrays are written ONCE, already
 * shifted / converted, into one caller-owned (page-locked) block laid out exactly as the device will
 * read it, so that a batch reaches the GPU in ONE asynchronous copy:
 *
 *   features fp32 [N, feat_width] | labels int64 [N] |
 *   indptr [N+1] | indices [E] | t_indptr [N+1] | t_indices [E] | t_slot [E] | t_pos [E] |
 *   max(in-degree, 1) fp32 [N] | in-degree + 1 fp32 [N] |
 *   cluster-schedule records of kind 0 | kind 1 | ...        (every segment 256-byte aligned)
 *
 * Block-diagonal union as dgl.batch defines it: node ids of member j shifted by sum_{i<j} N_i, edges
 * (and CSR positions) by sum_{i<j} E_i, members in order; fp64 features are rounded to fp32 as
 * torch.FloatTensor(np.concatenate(features)) rounds them; labels widen to int64.  Schedule records
 * (gts_cluster_schedule) are concatenated member by member with their row / neighbour ids shifted and
 * their per-edge sections re-packed to the widest member's loc_words.  The bytes equal what the Python
 * path (gts.batch + ClusterSchedule.concat, the tested reference) produces: tests/test_collate_host.py.
 *
 * gts_collate_plan fills `plan` (sizes and byte offsets) for `members`; gts_collate_batch writes the block
 * (`dst_bytes` >= plan.total_bytes) using up to `n_threads` threads of its own (<= 1: the calling thread only).
 * A schedule kind that some member lacks (sched_rec NULL) is dropped for the whole batch (plan.sched_clusters
 * = -1).  Returns GTS_OK, GTS_ERR_NULL, GTS_ERR_SHAPE (sizes beyond int32, dst too small, members whose
 * records do not match the kind's limits) or GTS_ERR_ARGKIND (feature / label element kinds). */
#define GTS_COLLATE_MAX_SCHEDULES 6
typedef struct {
  int64_t n_nodes, n_edges;
  const int32_t *indptr, *indices, *t_indptr, *t_indices, *t_slot, *t_pos;
  const void* features;     /* [n_nodes, feat_width] row-major */
  const void* labels;       /* [n_nodes], or NULL when the batch carries none */
  int32_t feat_bytes;       /* 4 = fp32, 8 = fp64 */
  int32_t label_bytes;      /* 8 = int64, 4 = int32 (0 with labels == NULL) */
  const int32_t* sched_rec[GTS_COLLATE_MAX_SCHEDULES];   /* [clusters, record words] per kind, or NULL */
  int64_t sched_clusters[GTS_COLLATE_MAX_SCHEDULES];
  int32_t sched_loc_words[GTS_COLLATE_MAX_SCHEDULES];
} gts_collate_member_t;
typedef struct {
  int32_t max_rows, max_srcs, tagged, reserved;          /* the limits the kind's records were built with */
} gts_collate_kind_t;
typedef struct {
  int64_t total_bytes, n_nodes, n_edges;
  int64_t features, labels;                              /* byte offsets; labels = -1 when absent */
  int64_t csr[8];                                        /* indptr, indices, t_indptr, t_indices, t_slot, t_pos, deg_clamped, deg_plus1 */
  int64_t sched[GTS_COLLATE_MAX_SCHEDULES];              /* byte offset of each kind's records (-1: dropped) */
  int64_t sched_clusters[GTS_COLLATE_MAX_SCHEDULES];     /* clusters of the union (-1: dropped) */
  int32_t sched_loc_words[GTS_COLLATE_MAX_SCHEDULES];    /* loc_words of the union's records */
  int32_t sched_record_words[GTS_COLLATE_MAX_SCHEDULES];
} gts_collate_plan_t;
int32_t gts_collate_plan(const gts_collate_member_t* members, int32_t n_members, int64_t feat_width,
                         const gts_collate_kind_t* kinds, int32_t n_kinds, gts_collate_plan_t* plan);
int32_t gts_collate_batch(const gts_collate_member_t* members, int32_t n_members, int64_t feat_width,
                          const gts_collate_kind_t* kinds, int32_t n_kinds, void* dst, int64_t dst_bytes,
                          int32_t n_threads, gts_collate_plan_t* plan);

#ifdef __cplusplus
}
#endif
#endif /* GTS_HIP_H */
