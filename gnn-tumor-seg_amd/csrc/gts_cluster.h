// Pieces the clustered kernels share (gts_spmm_cluster.hip: K1 / K2 at F = 256; gts_gat_cluster.hip: the GATConv
// aggregation over the same schedule records): record layout, LDS-DMA gathers, counted waits, count-specialised loops.
#pragma once
#include "gts_rows.h"

namespace gts {
namespace {

constexpr int kF = 256;          // feature width the clustered kernels are built for
constexpr int kHalfBytes = 512;  // one column half of a row
constexpr int kArgHalfBytes = 128;
constexpr unsigned kRsrcFlags = 0x00020000;
constexpr int kRecRegs = 8;      // a record is at most 64 * kRecRegs words
constexpr int kMaxRing = 6;

__host__ __device__ inline int pad4(int words) { return (words + 3) & ~3; }

// Word offsets inside one schedule record (all sections padded to 16 bytes):
//   [0..3] n_rows, n_srcs, n_edges, 0 | row ids | neighbour ids (tail repeats the last) | per row: first 8-edge
//   chunk (low 16 bits) and degree (high 16 bits) | uint8 per edge: neighbour's position, every row's edges padded
//   to whole 8-byte chunks (pads repeat the row's last edge) | uint8 per edge: tag, same shape (K2 only)
struct RecLayout {
  int rows, srcs, eoff, loc, tag, words;
};
__host__ __device__ inline RecLayout rec_layout(int max_rows, int max_srcs, int loc_words, bool tag) {
  RecLayout r;
  r.rows = 4;
  r.srcs = r.rows + pad4(max_rows);
  r.eoff = r.srcs + pad4(max_srcs);
  r.loc = r.eoff + pad4(max_rows);
  r.tag = r.loc + loc_words;
  r.words = r.tag + (tag ? loc_words : 0);
  return r;
}


__device__ __forceinline__ void barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// wait until at most n of this wave's vector-memory operations (the youngest) are outstanding
#define GTS_VMCNT_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
__device__ __forceinline__ void wait_vm_all_but(int n) {
  switch (n) {
    GTS_VMCNT_CASE(1) GTS_VMCNT_CASE(2) GTS_VMCNT_CASE(3) GTS_VMCNT_CASE(4) GTS_VMCNT_CASE(5) GTS_VMCNT_CASE(6)
    GTS_VMCNT_CASE(7) GTS_VMCNT_CASE(8) GTS_VMCNT_CASE(9) GTS_VMCNT_CASE(10) GTS_VMCNT_CASE(11) GTS_VMCNT_CASE(12)
    GTS_VMCNT_CASE(13) GTS_VMCNT_CASE(14) GTS_VMCNT_CASE(15) GTS_VMCNT_CASE(16) GTS_VMCNT_CASE(17) GTS_VMCNT_CASE(18)
    GTS_VMCNT_CASE(19) GTS_VMCNT_CASE(20) GTS_VMCNT_CASE(21) GTS_VMCNT_CASE(22) GTS_VMCNT_CASE(23) GTS_VMCNT_CASE(24)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // waiting for more is always correct
  }
}
__device__ __forceinline__ void barrier_all() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }


// `cnt` (1..8, wave-uniform) as a compile-time constant: straight-line code for the exact edge count of a chunk
template <typename Body>
__device__ __forceinline__ void for_count(int cnt, Body&& body) {
  switch (cnt) {
    case 1: body(IC<1>{}); break;
    case 2: body(IC<2>{}); break;
    case 3: body(IC<3>{}); break;
    case 4: body(IC<4>{}); break;
    case 5: body(IC<5>{}); break;
    case 6: body(IC<6>{}); break;
    case 7: body(IC<7>{}); break;
    default: body(IC<8>{}); break;
  }
}
__device__ __forceinline__ unsigned chunk_byte(const uint2& w, int q) { return ((q < 4 ? w.x : w.y) >> (8 * (q & 3))) & 0xFF; }


// ---- gathers (LDS-DMA) ---------------------------------------------------------------------------------------
// Two ways to issue `buffer_load_dwordx4 ... lds` (64 lanes x 16 B from per-lane byte offsets into one contiguous
// KiB of LDS):
//   * BuiltinDma: the clang builtin.  hipcc's wait-count pass then treats the transfer as a pending LDS write and
//     puts `s_waitcnt vmcnt(0)` in front of every later LDS read it cannot tell apart from it — right for the forms
//     that gather, wait, and only then read;
//   * RawDma: the same instruction as inline assembly, for the streaming form, which reads one image while the
//     gathers into the OTHER image are in flight and orders the two itself (counted vmcnt + workgroup barrier).
typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned char* LdsBytes;

struct BuiltinDma {
  __amdgpu_buffer_rsrc_t rsrc;
  __device__ __forceinline__ BuiltinDma(const void* base, unsigned bytes)
      : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, kRsrcFlags)) {}
  __device__ __forceinline__ void operator()(unsigned char* lds_dst, unsigned voff) const {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, reinterpret_cast<float*>(lds_dst), 16, voff, 0, 0, 0);
  }
};
struct RawDma {
  v4i rsrc;   // buffer resource words: base[31:0] | base[47:32] (stride 0) | bytes | flags
  __device__ __forceinline__ RawDma(const void* base, unsigned bytes) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(base);
    rsrc = v4i{static_cast<int>(b), static_cast<int>((b >> 32) & 0xFFFF), static_cast<int>(bytes), static_cast<int>(kRsrcFlags)};
  }
  __device__ __forceinline__ void operator()(unsigned char* lds_dst, unsigned voff) const {
    const unsigned at = static_cast<unsigned>(reinterpret_cast<uintptr_t>((LdsBytes)lds_dst));   // wave-uniform
    unsigned keep;   // M0 holds the LDS destination; it is handed back as found
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(at), "v"(voff), "s"(rsrc) : "memory");
  }
};

// One column half of n_srcs neighbour rows -> image [n_srcs][512 B].  `src_of(i)` gives this lane's neighbour id
// for the pair of rows (i, i + 1): lanes 0-31 fetch row i, lanes 32-63 row i + 1 (the record repeats the last id
// after the end, and the host reserves an even number of image rows).  `first`, `step` in rows (even).
template <typename Dma, typename SrcOf>
__device__ __forceinline__ void gather_halves(const Dma& dma, int part, unsigned char* image, int n_srcs,
                                              int first, int step, SrcOf&& src_of) {
  const int hl = threadIdx.x & 31;
  for (int i = first; i < n_srcs; i += step) {
    const unsigned voff = static_cast<unsigned>(src_of(i)) * (kF * 4u) + static_cast<unsigned>(part) * kHalfBytes + hl * 16u;
    dma(image + i * kHalfBytes, voff);
  }
}

constexpr int64_t kMaxLds = 160 * 1024;

inline int device_cus() {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    return n;
  }();
  return cus;
}

template <typename Kernel>
inline void allow_big_lds(Kernel k) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kMaxLds));
}

}  // namespace
}  // namespace gts
