// K12: node -> voxel projection ("scatter" in the reference's wording; a gather by supervoxel
// id) + library-level entry points (ABI version, error strings).
//
// HBM-bound streaming: 2 B/voxel of int16 ids in, row_bytes/voxel out, the lookup table
// (<= 32768 rows) stays in L2.  Lane l of a wave owns voxels base + j*64 + l, so every
// store instruction writes 64 consecutive rows (1 KiB for fp32x4 logits) and the ids of a
// 512-voxel wave tile arrive as ONE 16 B/lane load that is re-distributed through LDS.
#include "gts_rows.h"

namespace gts {
namespace {

// write-once output rows leave through non-temporal stores (they are 8x the input bytes and are
// not read again by this kernel); g_project_nt = 0 switches back to plain stores (tuning knob)
template <typename RowT>
__device__ __forceinline__ void store_row(RowT* dst, const RowT& v, int nt) {
  if (!nt) {
    *dst = v;
    return;
  }
  if constexpr (sizeof(RowT) == 16) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(*reinterpret_cast<const u4*>(&v), reinterpret_cast<u4*>(dst));
  } else if constexpr (sizeof(RowT) == 8) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store(*reinterpret_cast<const u2*>(&v), reinterpret_cast<u2*>(dst));
  } else {
    __builtin_nontemporal_store(v, dst);
  }
}

constexpr int kVoxPerLane = 8;                     // one 16-byte load of int16 ids
constexpr int kVoxPerWave = kWave * kVoxPerLane;   // 512
constexpr int kVoxPerBlock = kVoxPerWave * kWavesPerBlock;

// Row index inside [table rows..., background]: numpy semantics of table_plus_bg[id]
// for id in [-(n_rows+1), n_rows]; anything outside selects the background row.
__device__ __forceinline__ int resolve_row(int id, int n_rows) {
  const int r = id < 0 ? id + n_rows + 1 : id;
  return (r < 0 || r > n_rows) ? n_rows : r;
}

// ids of this wave's tile -> per-lane registers, lane l gets voxels j*64 + l.
__device__ __forceinline__ void load_tile_ids(const int16_t* __restrict__ svs, int64_t tile_base,
                                              int64_t n_vox, int16_t* lds_wave,
                                              int (&ids)[kVoxPerLane]) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t mine = tile_base + static_cast<int64_t>(lane) * kVoxPerLane;
  if (mine + kVoxPerLane <= n_vox && (reinterpret_cast<uintptr_t>(svs + mine) & 15) == 0) {
    *reinterpret_cast<uint4*>(lds_wave + lane * kVoxPerLane) =
        *reinterpret_cast<const uint4*>(svs + mine);
  } else {
#pragma unroll
    for (int j = 0; j < kVoxPerLane; ++j)
      lds_wave[lane * kVoxPerLane + j] = mine + j < n_vox ? svs[mine + j] : int16_t(-1);
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wave's own LDS writes have landed
#pragma unroll
  for (int j = 0; j < kVoxPerLane; ++j) ids[j] = lds_wave[j * kWave + lane];
  __builtin_amdgcn_wave_barrier();
}

template <typename RowT>
__global__ __launch_bounds__(kBlock) void project_rows_kernel(
    const int16_t* __restrict__ svs, const RowT* __restrict__ table,
    const RowT* __restrict__ bg_row, RowT* __restrict__ out, int64_t n_vox, int n_rows, int nt) {
  __shared__ int16_t lds[kVoxPerBlock];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  int16_t* lds_wave = lds + wave * kVoxPerWave;
  const RowT bg = *bg_row;
  const int64_t n_tiles = (n_vox + kVoxPerWave - 1) / kVoxPerWave;
  for (int64_t tile = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave; tile < n_tiles;
       tile += static_cast<int64_t>(gridDim.x) * kWavesPerBlock) {
    const int64_t base = tile * kVoxPerWave;
    int ids[kVoxPerLane];
    load_tile_ids(svs, base, n_vox, lds_wave, ids);
    RowT rows[kVoxPerLane];
#pragma unroll
    for (int j = 0; j < kVoxPerLane; ++j) {
      const int r = resolve_row(ids[j], n_rows);
      rows[j] = r == n_rows ? bg : table[r];
    }
#pragma unroll
    for (int j = 0; j < kVoxPerLane; ++j) {
      const int64_t i = base + j * kWave + lane;
      if (i < n_vox) store_row(out + i, rows[j], nt);
    }
  }
}

// labels of this wave's tile, lane l holding voxels j*64 + l -> out, as ONE 16 B/lane store
// (the inverse of load_tile_ids' redistribution), falling back to 2-byte stores on a ragged or
// unaligned tile.
__device__ __forceinline__ void store_tile_labels(int16_t* __restrict__ out, int64_t tile_base,
                                                  int64_t n_vox, int16_t* lds_wave,
                                                  const int (&labels)[kVoxPerLane]) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t mine = tile_base + static_cast<int64_t>(lane) * kVoxPerLane;
  // wave-uniform: the whole tile is inside the volume and its first voxel is 16-byte aligned
  const bool wide = tile_base + kVoxPerWave <= n_vox &&
                    (reinterpret_cast<uintptr_t>(out + tile_base) & 15) == 0;
  if (wide) {
#pragma unroll
    for (int j = 0; j < kVoxPerLane; ++j) lds_wave[j * kWave + lane] = static_cast<int16_t>(labels[j]);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    *reinterpret_cast<uint4*>(out + mine) = *reinterpret_cast<const uint4*>(lds_wave + lane * kVoxPerLane);
    __builtin_amdgcn_wave_barrier();
  } else {
#pragma unroll
    for (int j = 0; j < kVoxPerLane; ++j) {
      const int64_t i = tile_base + j * kWave + lane;
      if (i < n_vox) out[i] = static_cast<int16_t>(labels[j]);
    }
  }
}

// occ (optional): [X + Y + Z] plane flags, set to 1 for every x / y / z plane that holds a voxel
// with a non-zero (tumour) label — all writers store the same byte, so the race is benign.
// CLASSES = 4 reads a node's logits as one 16-byte row; 0 = runtime class count.
template <int CLASSES>
__global__ __launch_bounds__(kBlock) void project_argmax_kernel(
    const int16_t* __restrict__ svs, const float* __restrict__ logits,
    const int16_t* __restrict__ relabel, int16_t* __restrict__ out, int64_t n_vox, int n_rows,
    int n_classes, uint8_t* __restrict__ occ, int dim_x, int dim_y, int dim_z) {
  __shared__ int16_t lds[kVoxPerBlock];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  int16_t* lds_wave = lds + wave * kVoxPerWave;
  const int64_t n_tiles = (n_vox + kVoxPerWave - 1) / kVoxPerWave;
  for (int64_t tile = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave; tile < n_tiles;
       tile += static_cast<int64_t>(gridDim.x) * kWavesPerBlock) {
    const int64_t base = tile * kVoxPerWave;
    int ids[kVoxPerLane];
    load_tile_ids(svs, base, n_vox, lds_wave, ids);
    int labels[kVoxPerLane];
#pragma unroll
    for (int j = 0; j < kVoxPerLane; ++j) {
      const int r = resolve_row(ids[j], n_rows);
      int label = 0;  // background voxels are healthy (graph_io.py:22-23)
      if (r != n_rows) {
        if constexpr (CLASSES == 4) {
          const float4 row = reinterpret_cast<const float4*>(logits)[r];
          float best = row.x;  // first maximum, like torch.max(dim=1)
          if (best < row.y) best = row.y, label = 1;
          if (best < row.z) best = row.z, label = 2;
          if (best < row.w) label = 3;
        } else {
          const float* row = logits + static_cast<size_t>(r) * n_classes;
          float best = row[0];
          for (int c = 1; c < n_classes; ++c) {
            const float val = row[c];
            if (best < val) best = val, label = c;
          }
        }
      }
      if (occ != nullptr && label != 0) {
        const unsigned i = static_cast<unsigned>(base) + j * kWave + lane;  // n_vox < 2^31 here
        if (i < n_vox) {
          const unsigned plane = i / static_cast<unsigned>(dim_z);
          occ[plane / static_cast<unsigned>(dim_y)] = 1;
          occ[dim_x + plane % static_cast<unsigned>(dim_y)] = 1;
          occ[dim_x + dim_y + i % static_cast<unsigned>(dim_z)] = 1;
        }
      }
      labels[j] = relabel != nullptr ? relabel[label] : label;
    }
    store_tile_labels(out, base, n_vox, lds_wave, labels);
  }
}

inline unsigned stream_grid(int64_t n_vox) {
  const int64_t blocks = (n_vox + kVoxPerBlock - 1) / kVoxPerBlock;
  return static_cast<unsigned>(blocks < 4096 ? (blocks > 0 ? blocks : 1) : 4096);
}

inline void launch_project_argmax(const int16_t* svs, const float* logits, const int16_t* relabel,
                                  int16_t* out, int64_t n_vox, int n_rows, int n_classes,
                                  uint8_t* occ, int dx, int dy, int dz, hipStream_t st) {
  const unsigned grid = stream_grid(n_vox);
  if (n_classes == 4 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0)
    project_argmax_kernel<4><<<grid, kBlock, 0, st>>>(svs, logits, relabel, out, n_vox, n_rows, 4,
                                                      occ, dx, dy, dz);
  else
    project_argmax_kernel<0><<<grid, kBlock, 0, st>>>(svs, logits, relabel, out, n_vox, n_rows,
                                                      n_classes, occ, dx, dy, dz);
}

}  // namespace
}  // namespace gts

extern "C" int32_t gts_abi_version(void) { return 21; }

extern "C" const char* gts_error_string(int32_t code) {
  switch (code) {
    case GTS_OK: return "ok";
    case GTS_ERR_NULL: return "gts: required pointer is NULL";
    case GTS_ERR_SHAPE: return "gts: unsupported or inconsistent shape";
    case GTS_ERR_ARGKIND: return "gts: unsupported arg_bytes/row_bytes/mode";
    default: return code > 0 ? hipGetErrorString(static_cast<hipError_t>(code)) : "gts: unknown error";
  }
}

extern "C" int32_t gts_project_rows_i16(const int16_t* svs, const void* table,
                                        const void* bg_row, void* out, int64_t n_vox,
                                        int64_t n_rows, int32_t row_bytes, void* stream) {
  using namespace gts;
  if (!svs || !bg_row || !out || (n_rows > 0 && !table)) return GTS_ERR_NULL;
  if (n_vox < 0 || n_rows < 0 || n_rows > 32768) return GTS_ERR_SHAPE;
  if (n_vox == 0) return GTS_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned grid = stream_grid(n_vox);
  const int nr = static_cast<int>(n_rows);
  switch (row_bytes) {
    case 4:
      project_rows_kernel<uint32_t><<<grid, kBlock, 0, st>>>(
          svs, static_cast<const uint32_t*>(table), static_cast<const uint32_t*>(bg_row),
          static_cast<uint32_t*>(out), n_vox, nr, g_project_nt);
      break;
    case 8:
      project_rows_kernel<uint2><<<grid, kBlock, 0, st>>>(
          svs, static_cast<const uint2*>(table), static_cast<const uint2*>(bg_row),
          static_cast<uint2*>(out), n_vox, nr, g_project_nt);
      break;
    case 16:
      project_rows_kernel<uint4><<<grid, kBlock, 0, st>>>(
          svs, static_cast<const uint4*>(table), static_cast<const uint4*>(bg_row),
          static_cast<uint4*>(out), n_vox, nr, g_project_nt);
      break;
    default:
      return GTS_ERR_ARGKIND;
  }
  return launch_status();
}

extern "C" int32_t gts_project_argmax_i16(const int16_t* svs, const float* logits,
                                          const int16_t* relabel, int16_t* out, int64_t n_vox,
                                          int64_t n_rows, int64_t n_classes, void* stream) {
  using namespace gts;
  if (!svs || !out || (n_rows > 0 && !logits)) return GTS_ERR_NULL;
  if (n_vox < 0 || n_rows < 0 || n_rows > 32768 || n_classes < 1 || n_classes > 1024)
    return GTS_ERR_SHAPE;
  if (n_vox == 0) return GTS_OK;
  launch_project_argmax(svs, logits, relabel, out, n_vox, static_cast<int>(n_rows),
                        static_cast<int>(n_classes), nullptr, 1, 1, 1,
                        static_cast<hipStream_t>(stream));
  return launch_status();
}

extern "C" int32_t gts_project_argmax_occupancy_i16(const int16_t* svs, const float* logits,
                                                    int16_t* out, uint8_t* occupancy,
                                                    int64_t dim_x, int64_t dim_y, int64_t dim_z,
                                                    int64_t n_rows, int64_t n_classes,
                                                    void* stream) {
  using namespace gts;
  if (!svs || !out || !occupancy || (n_rows > 0 && !logits)) return GTS_ERR_NULL;
  if (dim_x < 0 || dim_y < 0 || dim_z < 0 || dim_x > 32768 || dim_y > 32768 || dim_z > 32768 ||
      n_rows < 0 || n_rows > 32768 || n_classes < 1 || n_classes > 1024)
    return GTS_ERR_SHAPE;
  const int64_t n_vox = dim_x * dim_y * dim_z;
  if (n_vox >= (1LL << 31)) return GTS_ERR_SHAPE;  // plane arithmetic is 32-bit
  if (n_vox == 0) return GTS_OK;
  launch_project_argmax(svs, logits, nullptr, out, n_vox, static_cast<int>(n_rows),
                        static_cast<int>(n_classes), occupancy, static_cast<int>(dim_x),
                        static_cast<int>(dim_y), static_cast<int>(dim_z),
                        static_cast<hipStream_t>(stream));
  return launch_status();
}
