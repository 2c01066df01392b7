// K1 / K2 at F = 256 over a CLUSTER ROW SCHEDULE: LDS-staged neighbour tiles.
//
// The plain kernels (gts_spmm.hip) fetch one 1 KiB source row per edge through the CU's L1 — 5.76 rows per
// destination on a supervoxel graph, three of four out of the XCD's L2 — behind a chain of dependent index
// loads (row extent -> edge ids -> rows) that every wave pays for every row.  Measured in the training step at
// 8 graphs per GPU: 84 / 89 us on the lattice, 59 / 66 us when every row has one neighbour (the same kernels as
// plain row copies): both the per-edge gather and the dependent chain cost time.
//
// Here the rows of the graph are dealt to workgroups as CLUSTERS computed once per graph on the host
// (gts_cluster_schedule, below): a cluster is a set of <= max_rows rows whose edges touch <= max_srcs distinct
// neighbour rows (about 2 per row on the 6-neighbour lattice instead of 5.76: the greedy builder finds
// same-parity rows, which share all their neighbours).  One UNIT of work = one column half (128 floats) of one
// cluster.  Persistent workgroups (two per CU) walk their units through a double-buffered LDS image:
//   * at the top of unit t every wave issues ITS share of the gathers of unit t + 1: the 512-byte half of every
//     distinct neighbour row goes into the other image by LDS-DMA (`buffer_load_dwordx4 ... lds` with per-lane
//     source addresses: one wave instruction = two row halves, no staging registers), and one more LDS-DMA brings
//     the index record of unit t + 2 (row ids, per-row edge ranges, one byte per edge = the neighbour's position in
//     the cluster's list) into a ring of three record slots.  No wave ever waits for an index before it can ask for
//     a row, and row fetches are in flight while the rows of unit t are reduced;
//   * then it reduces its rows of unit t out of LDS — half a wave per row, 16 B per lane, edges in CSR slot order,
//     so results are bit-identical to the plain kernels — and stores 512 B of `out` (+128 B of winners) per row;
//   * one s_barrier per unit; in front of it a wave waits for its own gathers with a COUNTED vmcnt that leaves the
//     stores it has just issued in flight (vector-memory operations retire in order: the gathers are older).
// Units of neighbouring clusters run on one XCD at the same time (halo rows meet in its L2).  The outputs land in
// the rows the reference's node order gives them; the schedule only decides which workgroup produces which row.
//
// Two other launch forms were measured and rejected (one workgroup per unit: 63 - 71 / 73 us; loader waves feeding a ring
// of slots to consumer waves: 81 - 90 / 108 - 120 us, against 64 - 67 us): their kernels live in
// tools/diag/spmm_cluster_rejected_forms.inc, not in the library.
#include <algorithm>
#include <queue>
#include <vector>

#include "gts_cluster.h"

namespace gts {
namespace {

struct ClusterArgs {
  const int32_t* rec;      // [n_clusters][layout.words]
  RecLayout layout;
  int n_clusters, max_srcs;
  const float* table;      // K1: x;  K2: gout
  const uint8_t* winners;  // K2: arg
  float* out;              // K1: out; K2: gx
  uint8_t* arg;            // K1: winners out (or null)
  unsigned table_bytes, winners_bytes;
  int relu_input, nt;
  int ring, slot_bytes, image_off, win_off;   // LDS: ring slots of slot_bytes = [record | image | winner image]
  int deal_off;            // LDS: 64 bytes of dealt units behind the images
  unsigned* counters;      // dynamic dealing: per XCD a unit counter (word 32 x) and a count of finished workgroups (word 32 x + 16),
                           // zero on entry and on exit (null: static round-robin dealing)
};

// ---- reduce the rows of one unit out of its LDS slot ------------------------------------------------------
// `first` / `step`: this wave's share of the unit's row pairs (lanes 0-31 one row, lanes 32-63 the next).  Per
// 8-edge chunk: one 8-byte read gives the chunk's neighbour positions, then CNT 16-byte reads and 3 CNT vector
// operations per column — straight-line code for the exact count the longer of the wave's two rows needs (CNT is
// wave-uniform).  The pads of a row's last chunk repeat its last edge: a repeated value never beats the maximum
// it already is (strict '<'), so the shorter row needs no mask in K1; K2 masks by the degree.
template <int ARGB, int WHATIF = 0>   // WHATIF (tools/diag only): 3 = no stores, 4 = stores without the reduction
__device__ __forceinline__ int reduce_max_rows(const ClusterArgs& a, const int32_t* l_rec, const unsigned char* image,
                                               int part, int first, int step) {
  const int lane = threadIdx.x & (kWave - 1), half = lane >> 5, hl = lane & 31;
  const int n_rows = l_rec[0];
  const uint32_t* info = reinterpret_cast<const uint32_t*>(l_rec + a.layout.eoff);
  const uint2* loc = reinterpret_cast<const uint2*>(l_rec + a.layout.loc);
  const unsigned char* mine = image + hl * 16;
  int trips = 0;
  for (int j0 = 2 * first; j0 < n_rows; j0 += 2 * step, ++trips) {   // j0 is wave-uniform: the even row of the pair
    const int j = j0 + half;
    const bool have = j < n_rows;
    const uint32_t ri = have ? info[j] : 0u;
    const int c0 = ri & 0xFFFF, deg = ri >> 16;
    const int deg_w = max(__builtin_amdgcn_readlane(deg, 0), __builtin_amdgcn_readlane(deg, 32));
    float best[4];
    int slot[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) best[t] = -INFINITY, slot[t] = -1;
    for (int c = 0; 8 * c < deg_w && WHATIF != 4; ++c) {
      const bool mine_too = 8 * c < deg;                     // the shorter row of the pair may have run out of chunks
      const uint2 w = loc[mine_too ? c0 + c : c0];
      for_count(min(8, deg_w - 8 * c), [&](auto cnt_c) {
        constexpr int CNT = decltype(cnt_c)::value;
        // at most six of a chunk's rows in registers at a time (a seventh and eighth follow): 24 instead of 32 value registers at this
        // kernel's cap of 64, which the form that deals its units needs for its own state
        constexpr int FIRST = CNT > 6 ? 6 : CNT;
        float4 val[FIRST];
#pragma unroll
        for (int q = 0; q < FIRST; ++q) val[q] = *reinterpret_cast<const float4*>(mine + chunk_byte(w, q) * kHalfBytes);
#pragma unroll
        for (int q = 0; q < FIRST; ++q) {
          const float v4[4] = {val[q].x, val[q].y, val[q].z, val[q].w};
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const bool up = mine_too && best[t] < v4[t];
            best[t] = up ? v4[t] : best[t];
            slot[t] = up ? 8 * c + q : slot[t];
          }
        }
        if constexpr (CNT > FIRST) {
          float4 rest[CNT - FIRST];
#pragma unroll
          for (int q = FIRST; q < CNT; ++q) rest[q - FIRST] = *reinterpret_cast<const float4*>(mine + chunk_byte(w, q) * kHalfBytes);
#pragma unroll
          for (int q = FIRST; q < CNT; ++q) {
            const float v4[4] = {rest[q - FIRST].x, rest[q - FIRST].y, rest[q - FIRST].z, rest[q - FIRST].w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const bool up = mine_too && best[t] < v4[t];
              best[t] = up ? v4[t] : best[t];
              slot[t] = up ? 8 * c + q : slot[t];
            }
          }
        }
      });
    }
    Vec<4> o;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bool dead = isinf(best[t]);   // same rules as spmm_max_fwd_kernel (gts_spmm.hip)
      o.v[t] = dead ? 0.0f : best[t];
      slot[t] = (dead || (a.relu_input && !(best[t] > 0.0f))) ? -1 : slot[t];
    }
    if (WHATIF == 3) {
      if (o.v[0] == 123.456f) a.out[0] = static_cast<float>(slot[0] + slot[1] + slot[2] + slot[3]) + o.v[1] + o.v[2] + o.v[3];
    } else if (have) {
      const size_t off = static_cast<size_t>(l_rec[a.layout.rows + j]) * kF + part * (kF / 2) + hl * 4;
      if (a.nt & 1) o.store_nt(a.out + off); else o.store(a.out + off);
      if constexpr (ARGB == 1) {
        const uint32_t w = (slot[0] & 0xFF) | ((slot[1] & 0xFF) << 8) | ((slot[2] & 0xFF) << 16) |
                           (static_cast<uint32_t>(slot[3] & 0xFF) << 24);
        *reinterpret_cast<uint32_t*>(a.arg + off) = w;
      }
    }
  }
  return trips * (ARGB == 1 ? 2 : 1);   // vector-memory instructions this wave has just issued
}

// K2: rows = sources u; staged = gradient half-rows and winner half-rows of the destinations of their out-edges
__device__ __forceinline__ int reduce_winner_rows(const ClusterArgs& a, const int32_t* l_rec, const unsigned char* image,
                                                  const unsigned char* winners, int part, int first, int step) {
  const int lane = threadIdx.x & (kWave - 1), half = lane >> 5, hl = lane & 31;
  const int n_rows = l_rec[0];
  const uint32_t* info = reinterpret_cast<const uint32_t*>(l_rec + a.layout.eoff);
  const uint2* loc = reinterpret_cast<const uint2*>(l_rec + a.layout.loc);
  const uint2* tag = reinterpret_cast<const uint2*>(l_rec + a.layout.tag);
  const unsigned char* my_g = image + hl * 16;
  const unsigned char* my_w = winners + hl * 4;
  int trips = 0;
  for (int j0 = 2 * first; j0 < n_rows; j0 += 2 * step, ++trips) {
    const int j = j0 + half;
    const bool have = j < n_rows;
    const uint32_t ri = have ? info[j] : 0u;
    const int c0 = ri & 0xFFFF, deg = ri >> 16;
    const int deg_a = __builtin_amdgcn_readlane(deg, 0), deg_b = __builtin_amdgcn_readlane(deg, 32);
    const int deg_w = max(deg_a, deg_b);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (deg_a == deg_b && deg_w <= 8) {
      // the common pair: equal degrees, one chunk — the exact count is both rows' count, no per-edge masks (the kernel issues
      // vector instructions for about as long as its memory traffic takes: every one saved shows, profiles/r04)
      if (deg_w > 0) {
        const uint2 w = loc[c0], tg = tag[c0];
        for_count(deg_w, [&](auto cnt_c) {
          constexpr int CNT = decltype(cnt_c)::value;
          // at most six of a chunk's rows in registers at a time (30 instead of 40 at this kernel's cap of 80: room for the state of the form
          // that deals its units); the sums keep their order
          constexpr int FIRST = CNT > 6 ? 6 : CNT;
          auto group = [&](auto q0_c, auto n_c) {
            constexpr int Q0 = decltype(q0_c)::value, N = decltype(n_c)::value;
            float4 g[N];
            uint32_t win[N];
#pragma unroll
            for (int q = 0; q < N; ++q) {
              const unsigned s = chunk_byte(w, Q0 + q);
              g[q] = *reinterpret_cast<const float4*>(my_g + s * kHalfBytes);
              win[q] = *reinterpret_cast<const uint32_t*>(my_w + s * kArgHalfBytes);
            }
#pragma unroll
            for (int q = 0; q < N; ++q) {
              const unsigned want = chunk_byte(tg, Q0 + q);
              acc[0] += (win[q] & 0xFF) == want ? g[q].x : 0.0f;
              acc[1] += ((win[q] >> 8) & 0xFF) == want ? g[q].y : 0.0f;
              acc[2] += ((win[q] >> 16) & 0xFF) == want ? g[q].z : 0.0f;
              acc[3] += (win[q] >> 24) == want ? g[q].w : 0.0f;
            }
          };
          group(IC<0>{}, IC<FIRST>{});
          if constexpr (CNT > FIRST) group(IC<FIRST>{}, IC<CNT - FIRST>{});
        });
      }
    } else {
      for (int c = 0; 8 * c < deg_w; ++c) {
        const int at = 8 * c < deg ? c0 + c : c0;
        const uint2 w = loc[at], tg = tag[at];
        for_count(min(8, deg_w - 8 * c), [&](auto cnt_c) {
          constexpr int CNT = decltype(cnt_c)::value;
          constexpr int FIRST = CNT > 6 ? 6 : CNT;
          auto group = [&](auto q0_c, auto n_c) {
            constexpr int Q0 = decltype(q0_c)::value, N = decltype(n_c)::value;
            float4 g[N];
            uint32_t win[N];
#pragma unroll
            for (int q = 0; q < N; ++q) {
              const unsigned s = chunk_byte(w, Q0 + q);
              g[q] = *reinterpret_cast<const float4*>(my_g + s * kHalfBytes);
              win[q] = *reinterpret_cast<const uint32_t*>(my_w + s * kArgHalfBytes);
            }
#pragma unroll
            for (int q = 0; q < N; ++q) {
              const unsigned want = chunk_byte(tg, Q0 + q);
              const bool live = 8 * c + Q0 + q < deg;
              const float g4[4] = {g[q].x, g[q].y, g[q].z, g[q].w};
#pragma unroll
              for (int t = 0; t < 4; ++t) acc[t] += (live && ((win[q] >> (8 * t)) & 0xFF) == want) ? g4[t] : 0.0f;
            }
          };
          group(IC<0>{}, IC<FIRST>{});
          if constexpr (CNT > FIRST) group(IC<FIRST>{}, IC<CNT - FIRST>{});
        });
      }
    }
    if (have) {
      const size_t off = static_cast<size_t>(l_rec[a.layout.rows + j]) * kF + part * (kF / 2) + hl * 4;
      const Vec<4> o{{acc[0], acc[1], acc[2], acc[3]}};
      if (a.nt & 1) o.store_nt(a.out + off); else o.store(a.out + off);
    }
  }
  return trips;
}

// winner halves (128 B per row): 8 lanes x 16 B per row, eight rows per wave instruction
template <typename Dma, typename SrcOf>
__device__ __forceinline__ void gather_winner_halves(const Dma& dma, int part, unsigned char* image,
                                                     int n_srcs, int first, int step, SrcOf&& src_of) {
  const int l8 = threadIdx.x & 7;
  for (int i = first; i < n_srcs; i += step) {
    const unsigned voff = static_cast<unsigned>(src_of(i)) * static_cast<unsigned>(kF) +
                          static_cast<unsigned>(part) * kArgHalfBytes + l8 * 16u;
    dma(image + i * kArgHalfBytes, voff);
  }
}

// ---- the persistent streaming kernel --------------------------------------------------------------------------
// gridDim.x is a multiple of 8: the workgroups with blockIdx % 8 == x (one XCD under round-robin placement; speed
// only) share the x-th eighth of the units and take them round-robin.  LDS: three record slots, two images.
// WHATIF != 0 only in the timing experiments of tools/diag (wrong results): 1 = no row gathers, 2 = no reduction and
// no stores, 3 = no stores, 4 = stores without the reduction (K1).
template <bool BWD, int ARGB, int WHATIF = 0, int DEPTH = 1, bool DEAL = false>   // DEAL: units dealt off the XCD's counter (a.counters)
// Two workgroups per CU: 16 waves each for K1 (<= 64 registers), 12 for K2 (its 68 registers: 6 waves per SIMD).  With 8 waves
// the reduction of a unit is latency-bound (lattice, 8 graphs: 67.6 / 72.2 us against 63.9 / 67.8; k-NN graphs of mean
// degree 7 - 10: 78 - 144 us against 65 - 96, profiles/r03_cluster_other_graphs.log).
__global__ __launch_bounds__(1024, BWD ? 6 : 8) void spmm_cluster_stream_kernel(const ClusterArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int n_waves = blockDim.x / kWave;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  const long long n_units = 2LL * a.n_clusters;
  const int lo = static_cast<int>(n_units * xcd / 8), hi = static_cast<int>(n_units * (xcd + 1) / 8);
  // Dealing (as in gat_cluster_stream_kernel, see there).  STATIC (a.counters == nullptr): unit lo + j + t * per_xcd is this workgroup's
  // t-th.  DYNAMIC (the default): the XCD's workgroups take their units off one counter, in the order "every cluster of the XCD's span for
  // the left half, then for the right half" — the units in flight stay neighbours in that walk whatever each workgroup's pace, and their
  // halo rows meet in the XCD's L2 (a workgroup one unit behind under static dealing is per_xcd units behind in the walk).
  constexpr bool dynamic = DEAL;           // a compile-time choice: as a run-time one it cost both forms spilled registers
  const int n_static = lo + j < hi ? (hi - lo - j + per_xcd - 1) / per_xcd : 0;   // units lo + j + t * per_xcd, t < n_static
  if (!dynamic && n_static == 0) return;
  const int clo = static_cast<int>(static_cast<long long>(a.n_clusters) * xcd / 8);
  const int span = static_cast<int>(static_cast<long long>(a.n_clusters) * (xcd + 1) / 8) - clo;
  const int words = a.layout.words;
  const int pieces = (words + 255) / 256;   // a record arrives as one or two whole 1 KiB LDS-DMA pieces (lanes past its end deliver zeros)
  const int rec_bytes = 1024 * pieces, image_bytes = a.slot_bytes - a.image_off;
  // gathers run `depth` units ahead of the reduction: depth + 1 images, depth + 2 record slots (the record of the unit whose
  // gathers are issued next is fetched one iteration before that)
  // (DEPTH is a template constant: with run-time slot counts the K1 form spills a register and both forms lose 8 - 10 %)
  constexpr int depth = DEPTH, n_images = DEPTH + 1, n_recs = DEPTH + 2;
  unsigned char* images = lds + n_recs * rec_bytes;
  const RawDma rr(a.rec, static_cast<unsigned>(a.n_clusters) * words * 4u);
  const RawDma rt(a.table, a.table_bytes);
  const RawDma rw(a.winners, BWD ? a.winners_bytes : 0);
  // units in the pipeline (2 * cluster + column half; < 0: none): [0] = the one fetch_record took last ... [PIPE - 1] = the one being reduced
  constexpr int PIPE = DEPTH + 2;
  // (kept only by the dealt form: under static dealing unit t is a closed form — its pipeline as registers cost spills)
  int u_unit[dynamic ? PIPE : 1];
#pragma unroll
  for (int q = 0; q < (dynamic ? PIPE : 1); ++q) u_unit[q] = -1;
  int32_t* l_deal = reinterpret_cast<int32_t*>(lds + a.deal_off);   // 0 .. PIPE - 2: the prologue's units, 6 / 7: the loop's
  // l_deal holds the counter's values as taken; every wave turns one into its unit (2 * cluster + half; < 0: the span is exhausted)
  auto next_unit = [&](int deal_slot) {
    if constexpr (dynamic) {
#pragma unroll
      for (int q = PIPE - 1; q > 0; --q) u_unit[q] = u_unit[q - 1];
      const unsigned idx = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(l_deal[deal_slot]));
      const unsigned part = idx >= static_cast<unsigned>(span) ? 1u : 0u;
      u_unit[0] = idx < 2u * static_cast<unsigned>(span) ? static_cast<int>(((clo + idx - part * span) << 1) | part) : -1;
    }
  };
  // stage k of the pipeline while unit `newest` is the one fetch_record takes: its unit, or < 0
  auto unit_at = [&](int k, int newest) {
    if constexpr (dynamic) {
      return u_unit[k];
    } else {
      const int t = newest - k;
      return t < n_static ? lo + j + t * per_xcd : -1;
    }
  };
  // the dealer (the workgroup's last wave): `count` units off the XCD's counter (take) into l_deal[slot0 ..] (publish).  In the loop
  // the add is issued in one iteration and published at the top of the next, behind the wait every wave makes there anyway: the dealer
  // never waits for the counter's round trip on its own.
  auto take = [&](int count) {
    unsigned base = 0;
    if (lane == 0) base = __hip_atomic_fetch_add(a.counters + 32 * xcd, static_cast<unsigned>(count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return base;
  };
  auto publish = [&](unsigned taken, int count, int slot0) {
    if (lane == 0) {
      for (int q = 0; q < count; ++q) l_deal[slot0 + q] = static_cast<int>(taken + q);
    }
  };
  auto fetch_record = [&](int t, int unit) {   // record of unit t -> its slot, by LDS-DMA (waves 0 / 1: one piece each)
    if (wave < pieces) {
      const unsigned cluster = static_cast<unsigned>(unit) >> 1;
      const int word = 256 * wave + 4 * lane;
      const unsigned voff = word < words ? (cluster * static_cast<unsigned>(words) + word) * 4u : 0xFFFFFFF0u;
      rr(lds + (t % n_recs) * rec_bytes + 1024 * wave, voff);
    }
  };
  auto issue_gathers = [&](int t, int unit) {   // this wave's share of unit t's gathers; its record is in LDS.  Returns their count
    const int32_t* l_rec = reinterpret_cast<const int32_t*>(lds + (t % n_recs) * rec_bytes);
    unsigned char* image = images + (t % n_images) * image_bytes;
    const int n_srcs = l_rec[1];
    const int part = unit & 1;
    const int half = lane >> 5;
    // lane l keeps the id of row 2 (wave + n_waves (l / 2)) + l % 2: every row pair this wave fetches, one LDS read
    const int last = pad4(a.max_srcs) - 1;
    const int32_t ids = l_rec[a.layout.srcs + min(2 * (wave + n_waves * (lane >> 1)) + (lane & 1), last)];
    int k = 0;
    gather_halves(rt, part, image, n_srcs, 2 * wave, 2 * n_waves, [&](int) {
      const int s0 = __builtin_amdgcn_readlane(ids, 2 * k), s1 = __builtin_amdgcn_readlane(ids, 2 * k + 1);
      ++k;
      return half ? s1 : s0;
    });
    if constexpr (BWD) {
      // lane l keeps the id of row 8 (wave + n_waves (l / 8)) + l % 8; piece k wants lanes 8 k .. 8 k + 7 spread by eights
      const int32_t ids8 = l_rec[a.layout.srcs + min(8 * (wave + n_waves * (lane >> 3)) + (lane & 7), last)];
      int k8 = 0;
      gather_winner_halves(rw, part, image + ((a.max_srcs + 1) & ~1) * kHalfBytes, n_srcs, 8 * wave, 8 * n_waves, [&](int) {
        const int32_t id = __builtin_amdgcn_ds_bpermute(4 * (8 * k8 + (lane >> 3)), ids8);
        ++k8;
        return id;
      });
      return k + k8;
    }
    return k;
  };

  // Per iteration a wave issues, in this order: [record fetch of unit it + depth + 1 (waves 0 / 1)] [gathers of unit it + depth]
  // [stores of unit it].  At the top of iteration `it` the gathers of unit `it` (issued `depth` iterations ago) and the record
  // fetched in iteration it - 1 must have landed; vector-memory operations retire in order, so everything YOUNGER than that
  // record fetch may stay in flight: the previous iteration's stores and, for depth >= 2, its gathers (for depth 1 those ARE
  // the gathers of unit `it`).
  if constexpr (dynamic) {
    if (wave == n_waves - 1) {
      const unsigned first = take(PIPE);      // ONE round trip for the prologue's units and the one iteration 0 enters
      publish(first, PIPE - 1, 0);
      publish(first + (PIPE - 1), 1, 6);
    }
    barrier_all();
  }
  unsigned taken = 0;                         // the dealer's add of the iteration before
#pragma unroll
  for (int t = 0; t <= depth; ++t) {        // afterwards stage depth - t holds unit t
    next_unit(t);
    if (unit_at(0, t) >= 0) fetch_record(t, unit_at(0, t));
  }
  barrier_all();
  int pending_gathers = 0, stores = 0;
#pragma unroll
  for (int t = 0; t < depth; ++t) pending_gathers = (WHATIF != 1 && unit_at(depth - t, depth) >= 0) ? issue_gathers(t, unit_at(depth - t, depth)) : 0;
  if (depth == 1) pending_gathers = 0;
  // WHATIF == 9 (tools/diag only): shader-clock stamps of wave 0 around the phases of an iteration, summed per workgroup into
  // `arg` (which then is a buffer of 8 x uint64 per workgroup, not the winners): wait for gathers | barrier | issue | reduce
  unsigned long long phase[4] = {0, 0, 0, 0}, stamp = 0;
  auto lap = [&](int which) {
    if constexpr (WHATIF == 9) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      phase[which] += now - stamp;
      stamp = now;
    }
  };
  if constexpr (WHATIF == 9) stamp = __builtin_amdgcn_s_memtime();
  int it = 0;
  for (; dynamic || it < n_static; ++it) {
    wait_vm_all_but(pending_gathers + stores);
    if constexpr (dynamic) {
      if (it > 0 && wave == n_waves - 1) publish(taken, 1, 6 + (it & 1));   // this slot was last read two barriers ago
    }
    lap(0);
    barrier_lds();                          // ... and everyone else's; the oldest image and the oldest record slot are free
    lap(1);
    next_unit(6 + (it & 1));                // stage 0 = unit it + depth + 1, stage 1 = unit it + depth, stage PIPE - 1 = unit it
    const int newest = it + depth + 1;
    const int unit_now = unit_at(PIPE - 1, newest);
    if constexpr (dynamic) {
      if (unit_now < 0) break;              // units come in walk order: nothing behind an empty slot
    }
    if constexpr (dynamic) {
      if (wave == n_waves - 1) taken = take(1);   // published at the top of the next iteration
    }
    if constexpr (DEPTH == 1) {             // the next unit's gathers first, then the record of the unit after it
      if (unit_at(1, newest) >= 0 && WHATIF != 1) issue_gathers(it + 1, unit_at(1, newest));
      if (unit_at(0, newest) >= 0) fetch_record(it + 2, unit_at(0, newest));
    } else {
      if (unit_at(0, newest) >= 0) fetch_record(it + depth + 1, unit_at(0, newest));
      pending_gathers = (unit_at(1, newest) >= 0 && WHATIF != 1) ? issue_gathers(it + depth, unit_at(1, newest)) : 0;
    }
    lap(2);
    const int32_t* l_rec = reinterpret_cast<const int32_t*>(lds + (it % n_recs) * rec_bytes);
    const unsigned char* image = images + (it % n_images) * image_bytes;
    const int part = unit_now & 1;
    if constexpr (WHATIF == 2)
      stores = 0;
    else if constexpr (WHATIF >= 3 && WHATIF <= 4 && !BWD)
      stores = WHATIF == 3 ? (reduce_max_rows<ARGB, WHATIF>(a, l_rec, image, part, wave, n_waves), 0)
                           : reduce_max_rows<ARGB, WHATIF>(a, l_rec, image, part, wave, n_waves);
    else if constexpr (BWD)
      stores = reduce_winner_rows(a, l_rec, image, image + ((a.max_srcs + 1) & ~1) * kHalfBytes, part, wave, n_waves);
    else
      stores = reduce_max_rows<ARGB>(a, l_rec, image, part, wave, n_waves);
    lap(3);
  }
  if constexpr (WHATIF == 9) {
    if (threadIdx.x == 0) {
      unsigned long long* dbg = reinterpret_cast<unsigned long long*>(a.arg) + 8 * static_cast<size_t>(blockIdx.x);
      for (int q = 0; q < 4; ++q) dbg[q] = phase[q];
      dbg[4] = static_cast<unsigned long long>(it);
    }
  }
  // the last workgroup of the XCD to finish leaves the counters zero for the next launch
  if (dynamic && threadIdx.x == 0) {   // (compile-time)
    unsigned* done = a.counters + 32 * xcd + 16;
    if (__hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == static_cast<unsigned>(per_xcd) - 1u) {
      __hip_atomic_store(a.counters + 32 * xcd, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

struct LdsPlan {
  int slot_bytes, image_off, win_off;
};
inline LdsPlan lds_plan(int max_rows, int max_srcs, int loc_words, bool bwd) {
  const RecLayout l = rec_layout(max_rows, max_srcs, loc_words, bwd);
  LdsPlan p;
  p.image_off = (l.words * 4 + 15) & ~15;
  const int image = ((max_srcs + 1) & ~1) * kHalfBytes;
  p.win_off = p.image_off + image;
  p.slot_bytes = p.win_off + (bwd ? ((max_srcs + 7) & ~7) * kArgHalfBytes : 0);
  return p;
}

// launch geometry of the streaming form (shared with gts_cluster_uses_counters): LDS per workgroup, gather depth, persistent workgroups
struct ClusterGeometry {
  int64_t wg_lds, grid;
  int depth, waves;
  bool deals;      // units dealt off the XCD's counter (where the caller gives counters)
};
inline ClusterGeometry cluster_geometry(int64_t n_clusters, int words, int slot_bytes, int image_off, bool bwd, bool whatif) {
  ClusterGeometry g;
  const int64_t units = 2LL * n_clusters;
  // depth + 2 record slots + depth + 1 images per workgroup; two workgroups per CU.  Gathers run one unit ahead; two units ahead
  // (GTS_OPT_CLUSTER_RING = 2, where that fits) is kept for A/B runs: no gain measured
  const int64_t rec_slot = 1024LL * ((words + 255) / 256), image = slot_bytes - image_off;
  g.depth = (g_cluster_ring == 2 && !whatif && 4 * rec_slot + 3 * image + 64 <= kMaxLds) ? 2 : 1;
  g.wg_lds = (g.depth + 2) * rec_slot + (g.depth + 1) * image + 64;   // + the dealt units
  g.waves = g_cluster_consumers > 0 ? std::min(16, g_cluster_consumers) : (bwd ? 12 : 16);
  // persistent workgroups per CU: two; three (where the LDS holds them) for a backward launch that gives a workgroup only a
  // few units — C2: 8.8 units per workgroup, the reference's batches: 5 — where the pipeline's fill and drain weigh most
  // (profiles/r04/tune_k2_small.log, operands from HBM: 37.0 -> 35.2 us at 60 000 rows, 22.8 -> 22.0 at 35 000; nothing either way from
  // 120 000 rows on, and nothing in the training step, where the gradient rows were just written by the GEMM in front: 36.7 / 36.9 us)
  const int auto_per_cu = (bwd && units <= 24LL * device_cus()) ? 3 : 2;
  const int per_cu = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(g_cluster_per_cu > 0 ? g_cluster_per_cu : auto_per_cu, kMaxLds / std::max<int64_t>(1, g.wg_lds))));
  g.grid = static_cast<int64_t>(device_cus()) * per_cu;
  g.grid = std::max<int64_t>(8, std::min(g.grid, (units + 7) / 8 * 8)) / 8 * 8;
  // Units dealt off the XCD's counter (GTS_OPT_CLUSTER_DEALING: 0 = automatic, 1 = wherever counters are given, 2 = never) from 16 units per
  // workgroup on: the dealt form brings the fabric traffic to its compulsory figure (32 graphs per GPU: 1.31 / 1.18 -> 1.05 / 1.03 x) at the
  // static form's time or a little under (8 graphs: K1 69.7 -> 70.4 us, K2 67.9 -> 66.1), but its first units cost a counter round trip
  // before anything else starts, which a launch of five to nine units per workgroup (the reference's batches, C2) does not earn back
  // (K2 25.4 -> 27.2 us at 35 000 rows; profiles/r04/k12_dealing_ab.log).
  g.deals = g_cluster_dealing == 1 || (g_cluster_dealing == 0 && units >= 16 * g.grid);
  return g;
}

template <bool BWD, int ARGB, int WHATIF = 0>
inline int launch_cluster(ClusterArgs a, int max_rows, int loc_words, hipStream_t st) {
  const LdsPlan p = lds_plan(max_rows, a.max_srcs, loc_words, BWD);
  a.image_off = p.image_off, a.win_off = p.win_off, a.slot_bytes = p.slot_bytes;
  if (p.slot_bytes > kMaxLds) return GTS_ERR_SHAPE;
  const int64_t units = 2LL * a.n_clusters;
  {
    const ClusterGeometry geo = cluster_geometry(a.n_clusters, a.layout.words, p.slot_bytes, p.image_off, BWD, WHATIF != 0);
    const int depth = geo.depth, waves = geo.waves;
    const int64_t wg_lds = geo.wg_lds, grid = geo.grid;
    a.deal_off = static_cast<int>(wg_lds) - 64;
    if (wg_lds > kMaxLds || a.layout.words > 512) return GTS_ERR_SHAPE;
    a.ring = depth;
    if (a.max_srcs > 64 * waves) return GTS_ERR_SHAPE;
    if (!geo.deals) a.counters = nullptr;
    static const bool once = (allow_big_lds(spmm_cluster_stream_kernel<BWD, ARGB, WHATIF, 1, false>),
                              allow_big_lds(spmm_cluster_stream_kernel<BWD, ARGB, WHATIF, 1, true>),
                              allow_big_lds(spmm_cluster_stream_kernel<BWD, ARGB, WHATIF == 0 ? 0 : WHATIF, WHATIF == 0 ? 2 : 1, false>),
                              allow_big_lds(spmm_cluster_stream_kernel<BWD, ARGB, WHATIF == 0 ? 0 : WHATIF, WHATIF == 0 ? 2 : 1, true>), true);
    (void)once;
    const dim3 g3(static_cast<unsigned>(grid));
    if constexpr (WHATIF == 0) {
      if (depth == 2) {
        if (a.counters != nullptr) spmm_cluster_stream_kernel<BWD, ARGB, 0, 2, true><<<g3, waves * kWave, wg_lds, st>>>(a);
        else spmm_cluster_stream_kernel<BWD, ARGB, 0, 2, false><<<g3, waves * kWave, wg_lds, st>>>(a);
        return launch_status();
      }
    }
    if (a.counters != nullptr) spmm_cluster_stream_kernel<BWD, ARGB, WHATIF, 1, true><<<g3, waves * kWave, wg_lds, st>>>(a);
    else spmm_cluster_stream_kernel<BWD, ARGB, WHATIF, 1, false><<<g3, waves * kWave, wg_lds, st>>>(a);
    return launch_status();
  }
}

}  // namespace
}  // namespace gts

// ---------------------------------------------------------------------------------------------------------
// Host: the greedy cluster builder.  Plain C++, no GPU call.
extern "C" int64_t gts_cluster_record_words(int32_t max_rows, int32_t max_srcs, int32_t loc_words, int32_t with_tag) {
  if (max_rows < 1 || max_srcs < 1 || loc_words < 0 || loc_words % 4 != 0) return -1;
  return gts::rec_layout(max_rows, max_srcs, loc_words, with_tag != 0).words;
}

extern "C" int32_t gts_cluster_schedule(const int32_t* indptr, const int32_t* indices, const int32_t* t_indptr,
                                        const int32_t* t_indices, const int32_t* edge_tag, int64_t n_rows,
                                        int32_t max_rows, int32_t max_srcs, int32_t max_edges, int32_t* rec,
                                        int64_t rec_capacity, int64_t* n_clusters, int64_t* n_staged,
                                        int32_t* max_cluster_edges) {
  using namespace gts;
  if (!indptr || !t_indptr || !n_clusters || !n_staged || !max_cluster_edges) return GTS_ERR_NULL;
  if (n_rows < 0 || n_rows >= (1LL << 31)) return GTS_ERR_SHAPE;
  if (max_rows < 1 || max_srcs < 1 || max_srcs > 256 || max_edges < 1 || max_edges > 65535) return GTS_ERR_ARGKIND;
  const int n = static_cast<int>(n_rows);
  if (n > 0 && (!indices || !t_indices) && indptr[n] > 0) return GTS_ERR_NULL;
  // a malformed CSR must come back as an error, not as reads and writes outside the builder's tables: row extents
  // ascending from 0, the transpose with the same edge total, every id inside [0, n)
  if (n > 0) {
    if (indptr[0] != 0 || t_indptr[0] != 0) return GTS_ERR_SHAPE;
    for (int v = 0; v < n; ++v)
      if (indptr[v + 1] < indptr[v] || t_indptr[v + 1] < t_indptr[v]) return GTS_ERR_SHAPE;
    if (indptr[n] != t_indptr[n]) return GTS_ERR_SHAPE;
    const int e_all = indptr[n];
    for (int k = 0; k < e_all; ++k)
      if (indices[k] < 0 || indices[k] >= n || t_indices[k] < 0 || t_indices[k] >= n) return GTS_ERR_SHAPE;
  }
  for (int v = 0; v < n; ++v) {   // every row must fit a cluster of its own
    const int deg = indptr[v + 1] - indptr[v];
    if (deg > max_srcs || ((deg + 7) & ~7) > max_edges || deg > 65535) return GTS_ERR_SHAPE;
  }
  if (edge_tag != nullptr) {
    const int e = n > 0 ? indptr[n] : 0;
    for (int k = 0; k < e; ++k)
      if (edge_tag[k] < 0 || edge_tag[k] > 255) return GTS_ERR_SHAPE;
  }
  const int loc_words = pad4((max_edges + 3) / 4);   // max_edges counts PADDED edges: every row takes whole 8-edge chunks
  const RecLayout L = rec_layout(max_rows, max_srcs, loc_words, edge_tag != nullptr);
  if (L.words > 64 * kRecRegs) return GTS_ERR_ARGKIND;
  std::vector<uint8_t> assigned(n, 0);
  std::vector<int> gain(n, 0), stamp(n, -1);      // gain[w] valid when stamp[w] == current cluster
  std::vector<int> local(n, -1), lstamp(n, -1);   // neighbour id -> position in this cluster's list (valid by lstamp)
  using Cand = std::pair<int, int>;                // (gain, -row): most shared neighbours first, then lowest id
  std::priority_queue<Cand> heap;
  std::priority_queue<int, std::vector<int>, std::greater<int>> seeds;   // rows that touch finished clusters
  std::vector<int> rows, srcs, touched;
  int next_unassigned = 0, most_edges = 0;
  int64_t n_cl = 0, staged = 0;
  while (true) {
    int seed = -1;
    while (!seeds.empty()) {
      const int s = seeds.top();
      seeds.pop();
      if (!assigned[s]) { seed = s; break; }
    }
    if (seed < 0) {
      while (next_unassigned < n && assigned[next_unassigned]) ++next_unassigned;
      if (next_unassigned == n) break;
      seed = next_unassigned;
    }
    const int cl = static_cast<int>(n_cl);
    rows.clear(), srcs.clear(), touched.clear();
    int n_e = 0;
    while (!heap.empty()) heap.pop();
    heap.push({0, -seed});
    stamp[seed] = cl, gain[seed] = 0;
    bool full = false, misfit = false;
    while (!full) {
      if (heap.empty()) {
        if (misfit) break;   // candidates were left out for lack of room: the cluster is as full as it gets
        // the component is exhausted: go on with the next unassigned row, so that small components and
        // isolated rows share clusters
        while (next_unassigned < n && assigned[next_unassigned]) ++next_unassigned;
        if (next_unassigned == n) break;
        const int s = next_unassigned;
        if (stamp[s] != cl) stamp[s] = cl, gain[s] = 0;
        heap.push({gain[s], -s});
      }
      const Cand top = heap.top();
      heap.pop();
      const int v = -top.second;
      if (assigned[v] || stamp[v] != cl || gain[v] != top.first) continue;   // stale entry
      const int beg = indptr[v], end = indptr[v + 1];
      int fresh = 0;
      for (int k = beg; k < end; ++k) {
        const int u = indices[k];
        if (lstamp[u] != cl) lstamp[u] = cl, local[u] = -1;
        if (local[u] == -1) local[u] = -2, ++fresh;          // -2: counted as new for this row
      }
      const int padded = (end - beg + 7) & ~7;
      const bool fits = static_cast<int>(rows.size()) < max_rows &&
                        static_cast<int>(srcs.size()) + fresh <= max_srcs && n_e + padded <= max_edges;
      if (!fits) {
        for (int k = beg; k < end; ++k)
          if (local[indices[k]] == -2) local[indices[k]] = -1;
        if (rows.empty()) return GTS_ERR_SHAPE;   // cannot happen after the degree check
        misfit = true;
        continue;                                  // another candidate may still fit
      }
      assigned[v] = 1;
      rows.push_back(v);
      n_e += padded;
      for (int k = beg; k < end; ++k) {
        const int u = indices[k];
        if (local[u] != -2) continue;
        local[u] = static_cast<int>(srcs.size());
        srcs.push_back(u);
        for (int q = t_indptr[u]; q < t_indptr[u + 1]; ++q) {   // every row that lists u shares it now
          const int w = t_indices[q];
          if (assigned[w]) continue;
          if (stamp[w] != cl) stamp[w] = cl, gain[w] = 0, touched.push_back(w);
          ++gain[w];
          heap.push({gain[w], -w});
        }
      }
      if (static_cast<int>(rows.size()) >= max_rows) full = true;
    }
    for (int w : touched)
      if (!assigned[w]) seeds.push(w);
    most_edges = std::max(most_edges, n_e);
    staged += static_cast<int64_t>(srcs.size());
    if (rec != nullptr && n_cl < rec_capacity) {
      // the record: rows in ascending id (neighbouring output rows leave together), edges in CSR slot order
      std::sort(rows.begin(), rows.end());
      int32_t* r = rec + n_cl * L.words;
      std::fill(r, r + L.words, 0);
      r[0] = static_cast<int32_t>(rows.size()), r[1] = static_cast<int32_t>(srcs.size()), r[2] = n_e;
      uint8_t* loc = reinterpret_cast<uint8_t*>(r + L.loc);
      uint8_t* tag = reinterpret_cast<uint8_t*>(r + L.tag);
      int e_at = 0;   // padded edge position: a multiple of 8 at every row start
      for (size_t i = 0; i < rows.size(); ++i) {
        const int v = rows[i], deg = indptr[v + 1] - indptr[v];
        r[L.rows + i] = v;
        r[L.eoff + i] = static_cast<int32_t>(static_cast<uint32_t>(e_at / 8) | (static_cast<uint32_t>(deg) << 16));
        for (int k = indptr[v]; k < indptr[v + 1]; ++k, ++e_at) {
          loc[e_at] = static_cast<uint8_t>(local[indices[k]]);
          if (edge_tag != nullptr) tag[e_at] = static_cast<uint8_t>(edge_tag[k]);
        }
        for (; e_at % 8 != 0; ++e_at) {   // pads repeat the row's last edge
          loc[e_at] = loc[e_at - 1];
          if (edge_tag != nullptr) tag[e_at] = tag[e_at - 1];
        }
      }
      for (int i = 0; i < pad4(max_srcs); ++i)   // the tail repeats the last id: gathers run in whole pairs / octets
        r[L.srcs + i] = srcs.empty() ? 0 : srcs[std::min<size_t>(i, srcs.size() - 1)];
    }
    ++n_cl;
  }
  *n_clusters = n_cl;
  *n_staged = staged;
  *max_cluster_edges = most_edges;
  return GTS_OK;
}

// LDS bytes of one ring slot (= of one workgroup of the simple form) for a schedule with these limits
// (kind 0 = K1 forward, 1 = K2 backward).
extern "C" int64_t gts_cluster_lds_bytes(int32_t max_rows, int32_t max_srcs, int32_t loc_words, int32_t kind) {
  if (max_rows < 1 || max_srcs < 1 || loc_words < 0 || loc_words % 4 != 0 || (kind != 0 && kind != 1)) return -1;
  return gts::lds_plan(max_rows, max_srcs, loc_words, kind == 1).slot_bytes;
}

namespace {
inline bool bad_cluster_shape(int64_t n_clusters, int32_t max_rows, int32_t max_srcs, int32_t loc_words, bool tag,
                              int64_t n_rows, int64_t n_feat) {
  if (n_feat != gts::kF || n_rows < 0 || n_rows * gts::kF * 4 >= (1LL << 32) || n_clusters < 0 || n_clusters >= (1 << 30))
    return true;
  if (max_rows < 1 || max_srcs < 1 || max_srcs > 256 || loc_words < 0 || loc_words % 4 != 0 || loc_words > 16384) return true;
  return gts::rec_layout(max_rows, max_srcs, loc_words, tag).words > 64 * gts::kRecRegs;
}
}  // namespace

extern "C" int32_t gts_cluster_uses_counters(int64_t n_clusters, int32_t max_rows, int32_t max_srcs, int32_t loc_words, int32_t backward) {
  using namespace gts;
  if (n_clusters <= 0 || max_rows < 1 || max_srcs < 1 || loc_words < 0) return 0;
  const LdsPlan p = lds_plan(max_rows, max_srcs, loc_words, backward != 0);
  return cluster_geometry(n_clusters, rec_layout(max_rows, max_srcs, loc_words, backward != 0).words, p.slot_bytes, p.image_off, backward != 0, false).deals ? 1 : 0;
}

extern "C" int32_t gts_spmm_max_fwd_cluster_f32(const int32_t* rec, int64_t n_clusters, int32_t max_rows, int32_t max_srcs,
                                                int32_t loc_words, const float* x, float* out, void* arg,
                                                int32_t arg_bytes, int32_t relu_input, int64_t n_rows, int64_t n_feat,
                                                uint32_t* counters, void* stream) {
  using namespace gts;
  if (!rec || !x || !out || (arg_bytes != 0 && !arg)) return GTS_ERR_NULL;
  if (bad_cluster_shape(n_clusters, max_rows, max_srcs, loc_words, false, n_rows, n_feat)) return GTS_ERR_SHAPE;
  if (arg_bytes != 0 && arg_bytes != 1) return GTS_ERR_ARGKIND;
  if (n_clusters == 0) return GTS_OK;
  ClusterArgs a{};
  a.rec = rec, a.layout = rec_layout(max_rows, max_srcs, loc_words, false);
  a.n_clusters = static_cast<int>(n_clusters), a.max_srcs = max_srcs;
  a.table = x, a.out = out, a.arg = static_cast<uint8_t*>(arg), a.counters = counters;
  a.table_bytes = static_cast<unsigned>(n_rows * kF * 4);
  a.relu_input = relu_input, a.nt = g_cluster_nt < 0 ? 1 : g_cluster_nt;
  hipStream_t st = static_cast<hipStream_t>(stream);
  return arg_bytes == 0 ? launch_cluster<false, 0>(a, max_rows, loc_words, st) : launch_cluster<false, 1>(a, max_rows, loc_words, st);
}

extern "C" int32_t gts_spmm_max_bwd_cluster_f32(const int32_t* rec, int64_t n_clusters, int32_t max_rows, int32_t max_srcs,
                                                int32_t loc_words, const float* gout, const void* arg, int32_t arg_bytes,
                                                float* gx, int64_t n_rows, int64_t n_feat, uint32_t* counters, void* stream) {
  using namespace gts;
  if (!rec || !gout || !arg || !gx) return GTS_ERR_NULL;
  if (bad_cluster_shape(n_clusters, max_rows, max_srcs, loc_words, true, n_rows, n_feat)) return GTS_ERR_SHAPE;
  if (arg_bytes != 1) return GTS_ERR_ARGKIND;
  if (n_clusters == 0) return GTS_OK;
  ClusterArgs a{};
  a.rec = rec, a.layout = rec_layout(max_rows, max_srcs, loc_words, true);
  a.n_clusters = static_cast<int>(n_clusters), a.max_srcs = max_srcs;
  a.table = gout, a.winners = static_cast<const uint8_t*>(arg), a.out = gx, a.counters = counters;
  a.table_bytes = static_cast<unsigned>(n_rows * kF * 4), a.winners_bytes = static_cast<unsigned>(n_rows * kF);
  a.nt = g_cluster_nt < 0 ? 1 : g_cluster_nt;
  return launch_cluster<true, 1>(a, max_rows, loc_words, static_cast<hipStream_t>(stream));
}
