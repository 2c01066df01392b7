// Shared device helpers for the gfx950 kernels of libgts_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gts_hip.h"

namespace gts {

constexpr int kWave = 64;        // CDNA4 wavefront
constexpr int kBlock = 256;      // 4 waves per workgroup
constexpr int kWavesPerBlock = kBlock / kWave;

// Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one L2).  Give each
// XCD a contiguous span of tiles so that neighbouring rows of a supervoxel graph — whose
// in-neighbours are nearby rows — are fetched through the same 4 MiB L2.  Bijective for
// any grid size.  Speed only: correctness never depends on placement.
__device__ __forceinline__ int xcd_contiguous_tile(int b, int nb) {
  const int q = nb >> 3, r = nb & 7, x = b & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

template <int VEC>
struct Vec;
template <>
struct Vec<4> {
  float v[4];
  __device__ __forceinline__ static Vec load(const float* p) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    return Vec{{t.x, t.y, t.z, t.w}};
  }
  __device__ __forceinline__ void store(float* p) const {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
  // streaming variants: the line is not kept in L2 (write-once outputs / read-once inputs)
  __device__ __forceinline__ void store_nt(float* p) const {
    typedef float f4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(f4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f4*>(p));
  }
  __device__ __forceinline__ static Vec load_nt(const float* p) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
    return Vec{{t[0], t[1], t[2], t[3]}};
  }
};
template <>
struct Vec<1> {
  float v[1];
  __device__ __forceinline__ static Vec load(const float* p) { return Vec{{*p}}; }
  __device__ __forceinline__ void store(float* p) const { *p = v[0]; }
  __device__ __forceinline__ void store_nt(float* p) const { __builtin_nontemporal_store(v[0], p); }
  __device__ __forceinline__ static Vec load_nt(const float* p) { return Vec{{__builtin_nontemporal_load(p)}}; }
};

// Entry `index` of an array member of the kernel's (single, by-value) argument struct, read
// straight from the kernarg segment with a scalar load.  Indexing the by-value copy with a
// runtime value would make the compiler move the whole struct to scratch memory.
template <typename T>
__device__ __forceinline__ T kernarg_entry(size_t member_offset, int index) {
  typedef const __attribute__((address_space(4))) char* KernargBytes;
  typedef const __attribute__((address_space(4))) T* KernargT;
  KernargBytes base = (KernargBytes)__builtin_amdgcn_kernarg_segment_ptr();
  return *(KernargT)(base + member_offset + sizeof(T) * index);
}

inline int launch_status() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? GTS_OK : static_cast<int>(e);
}

// out[c] = sum over chunks of partial[chunk][c], fixed association (gts_gat_reduce.hip); cols % 4 == 0
int sum_chunks(const float* partial, float* out, int cols, int chunks, hipStream_t st);

// smallest power of two >= x, capped at 64 (lanes that cooperate on one row)
inline int lanes_per_row(int64_t vec_cols) {
  int l = 1;
  while (l < 64 && l < vec_cols) l <<= 1;
  return l;
}

}  // namespace gts
