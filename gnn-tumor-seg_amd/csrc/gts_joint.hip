// K16/K17: device glue between the GNN and the refinement CNN of the joint predictor
// (/root/reference/scripts/generate_joint_predictions.py:59-73, model/cnn_model.py:85-88).
//
//   crop_concat     cnn_in[c, i, j, k] = c < Ci ? img[xs[i], ys[j], zs[k], c]
//                                               : table_plus_bg[svs[xs[i], ys[j], zs[k]]][c - Ci]
//                   i.e. torch.cat([img, node_logits_plus_bg[svs]], -1)[np.ix_(xs, ys, zs)]
//                   .movedim(-1, 0) in one pass: the [X,Y,Z,4] voxel-logit volume and the
//                   [X,Y,Z,8] concatenation are never materialised.
//   argmax_scatter  out[xs[i], ys[j], zs[k]] = argmax_c scores[c, i, j, k]  (first maximum)
//                   i.e. brain_volume_preds[crop] = argmax(refined_logits, 0).
//
// Both are HBM-bound copies over the cropped box.  One thread per cropped voxel, consecutive
// threads along k (the contiguous axis of img / svs and of every channel plane), so channel-plane
// accesses are fully coalesced and the channels-last reads are 16 B per lane.
#include "gts_common.h"

namespace gts {
namespace {

struct Box {
  const int32_t* xs;
  const int32_t* ys;
  const int32_t* zs;
  int cx, cy, cz;      // cropped extents
  int dim_y, dim_z;    // full extents of the two inner axes
};

__device__ __forceinline__ int64_t source_voxel(const Box& b, int64_t t) {
  const int k = static_cast<int>(t % b.cz);
  const int64_t ij = t / b.cz;
  const int j = static_cast<int>(ij % b.cy), i = static_cast<int>(ij / b.cy);
  return (static_cast<int64_t>(b.xs[i]) * b.dim_y + b.ys[j]) * b.dim_z + b.zs[k];
}

template <int CI, int CT>  // 0 = runtime width
__global__ __launch_bounds__(kBlock) void crop_concat_kernel(
    const float* __restrict__ img, const int16_t* __restrict__ svs, const float* __restrict__ table,
    const float* __restrict__ bg_row, float* __restrict__ out, Box box, int n_rows, int ci_rt,
    int ct_rt) {
  const int ci = CI ? CI : ci_rt, ct = CT ? CT : ct_rt;
  const int64_t n_crop = static_cast<int64_t>(box.cx) * box.cy * box.cz;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; t < n_crop;
       t += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t v = source_voxel(box, t);
    const int id = svs[v];
    const int r = id < 0 ? id + n_rows + 1 : id;  // numpy: table_plus_bg[id]
    const float* row = (r < 0 || r >= n_rows) ? bg_row : table + static_cast<size_t>(r) * ct;
    if constexpr (CI == 4 && CT == 4) {
      const float4 a = *reinterpret_cast<const float4*>(img + 4 * v);
      const float4 b = *reinterpret_cast<const float4*>(row);
      out[t] = a.x, out[n_crop + t] = a.y, out[2 * n_crop + t] = a.z, out[3 * n_crop + t] = a.w;
      out[4 * n_crop + t] = b.x, out[5 * n_crop + t] = b.y;
      out[6 * n_crop + t] = b.z, out[7 * n_crop + t] = b.w;
    } else {
      for (int c = 0; c < ci; ++c) out[c * n_crop + t] = img[static_cast<size_t>(v) * ci + c];
      for (int c = 0; c < ct; ++c) out[(ci + c) * n_crop + t] = row[c];
    }
  }
}

__global__ __launch_bounds__(kBlock) void argmax_scatter_kernel(
    const float* __restrict__ scores, const int16_t* __restrict__ relabel,
    int16_t* __restrict__ out, Box box, int n_classes) {
  const int64_t n_crop = static_cast<int64_t>(box.cx) * box.cy * box.cz;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; t < n_crop;
       t += static_cast<int64_t>(gridDim.x) * kBlock) {
    float best = scores[t];
    int label = 0;
    for (int c = 1; c < n_classes; ++c) {
      const float val = scores[c * n_crop + t];
      if (best < val) best = val, label = c;  // first maximum, like torch.argmax
    }
    if (relabel != nullptr) label = relabel[label];
    out[source_voxel(box, t)] = static_cast<int16_t>(label);
  }
}

inline unsigned crop_grid(int64_t n_crop) {
  const int64_t blocks = (n_crop + kBlock - 1) / kBlock;
  return static_cast<unsigned>(blocks > 8192 ? 8192 : blocks);
}

inline bool bad_box(int64_t cx, int64_t cy, int64_t cz, int64_t dim_y, int64_t dim_z) {
  return cx < 0 || cy < 0 || cz < 0 || cx > 32768 || cy > 32768 || cz > 32768 || dim_y < 1 ||
         dim_z < 1 || dim_y > 32768 || dim_z > 32768 || cy > dim_y || cz > dim_z;
}

}  // namespace
}  // namespace gts

extern "C" int32_t gts_crop_concat_f32(const float* img, const int16_t* svs, const float* table,
                                       const float* bg_row, const int32_t* xs, const int32_t* ys,
                                       const int32_t* zs, float* out, int64_t cx, int64_t cy,
                                       int64_t cz, int64_t dim_y, int64_t dim_z, int64_t n_rows,
                                       int64_t img_channels, int64_t row_channels, void* stream) {
  using namespace gts;
  if (bad_box(cx, cy, cz, dim_y, dim_z) || n_rows < 0 || n_rows > 32768 || img_channels < 0 ||
      row_channels < 1 || img_channels > 64 || row_channels > 64)
    return GTS_ERR_SHAPE;
  const int64_t n_crop = cx * cy * cz;
  if (n_crop == 0) return GTS_OK;
  if (!svs || !bg_row || !xs || !ys || !zs || !out || (img_channels > 0 && !img) ||
      (n_rows > 0 && !table))
    return GTS_ERR_NULL;
  const Box box{xs, ys, zs, static_cast<int>(cx), static_cast<int>(cy), static_cast<int>(cz),
                static_cast<int>(dim_y), static_cast<int>(dim_z)};
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool wide = img_channels == 4 && row_channels == 4 &&
                    ((reinterpret_cast<uintptr_t>(img) | reinterpret_cast<uintptr_t>(table) |
                      reinterpret_cast<uintptr_t>(bg_row)) & 15) == 0;
  if (wide)
    crop_concat_kernel<4, 4><<<crop_grid(n_crop), kBlock, 0, st>>>(
        img, svs, table, bg_row, out, box, static_cast<int>(n_rows), 4, 4);
  else
    crop_concat_kernel<0, 0><<<crop_grid(n_crop), kBlock, 0, st>>>(
        img, svs, table, bg_row, out, box, static_cast<int>(n_rows), static_cast<int>(img_channels),
        static_cast<int>(row_channels));
  return launch_status();
}

extern "C" int32_t gts_argmax_scatter_i16(const float* scores, const int16_t* relabel,
                                          const int32_t* xs, const int32_t* ys, const int32_t* zs,
                                          int16_t* out, int64_t cx, int64_t cy, int64_t cz,
                                          int64_t dim_y, int64_t dim_z, int64_t n_classes,
                                          void* stream) {
  using namespace gts;
  if (bad_box(cx, cy, cz, dim_y, dim_z) || n_classes < 1 || n_classes > 1024) return GTS_ERR_SHAPE;
  const int64_t n_crop = cx * cy * cz;
  if (n_crop == 0) return GTS_OK;
  if (!scores || !xs || !ys || !zs || !out) return GTS_ERR_NULL;
  const Box box{xs, ys, zs, static_cast<int>(cx), static_cast<int>(cy), static_cast<int>(cz),
                static_cast<int>(dim_y), static_cast<int>(dim_z)};
  argmax_scatter_kernel<<<crop_grid(n_crop), kBlock, 0, static_cast<hipStream_t>(stream)>>>(
      scores, relabel, out, box, static_cast<int>(n_classes));
  return launch_status();
}
