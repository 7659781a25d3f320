// lsq_pc_geom.hpp -- the window-mode work decomposition of the per-channel kernels, shared by the
// fake-quantize kernels (lsq_per_channel.hip) and the observer-statistics kernels (lsq_observe.hip).
//
// Data view: dense memory as [outer][L], L = C*inner; a workgroup owns a window of 256 x V positions of
// the row and walks down a slab of rows, so a lane keeps the same positions -- hence the same
// channel(s) -- for its whole life.  Short rows (L < 256*V) fold R rows into one tile.
#pragma once

#include "lsq_kernels.hpp"

namespace lsq {

// floor(a / b) for non-negative operands; 32-bit path when the whole row index space fits
__device__ __forceinline__ int64_t udiv(int64_t a, int64_t b, bool fits32) {
    return fits32 ? static_cast<int64_t>(static_cast<uint32_t>(a) / static_cast<uint32_t>(b)) : a / b;
}

// V elements per lane: one element, a full 16-byte packet (V == IO::VEC), or -- 16-bit storage only -- half a
// packet (V == 4, one global_load/store_dwordx2).
template <typename IO, int V, bool NTL>
__device__ __forceinline__ void load_elems(const void* base, int64_t e, typename IO::elem (&out)[V]) {
    using E = typename IO::elem;
    if constexpr (V == 1) {
        out[0] = static_cast<const E*>(base)[e];
    } else if constexpr (V * sizeof(E) == 8) {
        using V2 = __attribute__((ext_vector_type(2))) unsigned int;
        typedef V2 V2u __attribute__((aligned(2)));      // element alignment is all a view promises (lsq_math.hpp, PacketWord)
        const V2u* src = reinterpret_cast<const V2u*>(static_cast<const E*>(base) + e);
        const V2 raw = NTL ? __builtin_nontemporal_load(src) : *src;
        __builtin_memcpy(&out[0], &raw, 8);
    } else {
        static_assert(V == IO::VEC, "a lane moves 1 element, 8 bytes or 16 bytes");
        const Packet<IO> pk = NTL ? load_packet_nt<IO>(base, e) : load_packet<IO>(base, e);
#pragma unroll
        for (int j = 0; j < V; ++j) out[j] = pk.v[j];
    }
}

template <typename IO, int V, bool NTS>
__device__ __forceinline__ void store_elems(void* base, int64_t e, const typename IO::elem (&in)[V]) {
    using E = typename IO::elem;
    if constexpr (V == 1) {
        static_cast<E*>(base)[e] = in[0];
    } else if constexpr (V * sizeof(E) == 8) {
        using V2 = __attribute__((ext_vector_type(2))) unsigned int;
        V2 raw;
        __builtin_memcpy(&raw, &in[0], 8);
        typedef V2 V2u __attribute__((aligned(2)));
        V2u* dst = reinterpret_cast<V2u*>(static_cast<E*>(base) + e);
        if (NTS) __builtin_nontemporal_store(raw, dst); else *dst = raw;
    } else {
        static_assert(V == IO::VEC, "a lane moves 1 element, 8 bytes or 16 bytes");
        Packet<IO> pk;
#pragma unroll
        for (int j = 0; j < V; ++j) pk.v[j] = in[j];
        if (NTS) store_packet_nt<IO>(base, e, pk); else store_packet<IO>(base, e, pk);
    }
}

// ---- LDS-DMA (global -> LDS without passing through VGPRs) ------------------------------------------------------------
// One wave-wide 16-byte-per-lane copy: lane l's 16 bytes at `gsrc` land at LDS byte offset `lds_dst` + 16 l (the
// destination is wave-uniform base + lane x size; M0 carries the base).  The load is in flight without holding any
// VGPR, which is the point: a streaming kernel whose registers are taken by its arithmetic can still keep several rows
// of HBM requests outstanding per wave.  hipcc does not count asm memory operations in its s_waitcnt bookkeeping:
// the reader orders itself with wait_vm<N>() below (cdna_hip_programming.md: LDS-DMA recipe, M0 rule).
template <bool NT = false>
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    if constexpr (NT) {     // streaming hint, as global_load ... nt
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(gsrc), "s"(lds_dst)
                     : "memory");
    } else {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(gsrc), "s"(lds_dst)
                     : "memory");
    }
}
// the 4-byte form: lane l's dword lands at LDS byte offset `lds_dst` + 4 l (staging of per-channel parameters)
__device__ __forceinline__ void glds4(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ void glds16_rt(const void* gsrc, uint32_t lds_dst, int nt) {     // nt wave-uniform
    if (nt) glds16<true>(gsrc, lds_dst);
    else glds16<false>(gsrc, lds_dst);
}
// Wait until at most N of this wave's vector-memory operations are outstanding.  They complete in issue order on gfx9
// (loads, stores and LDS-DMA share the one counter), so "at most N outstanding" = "everything but the youngest N is done".
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_vm_upto(int n) {     // n uniform; n > 15 waits as for 15 (longer than needed)
    switch (n < 15 ? n : 15) {
        case 0: wait_vm<0>(); break;
        case 1: wait_vm<1>(); break;
        case 2: wait_vm<2>(); break;
        case 3: wait_vm<3>(); break;
        case 4: wait_vm<4>(); break;
        case 5: wait_vm<5>(); break;
        case 6: wait_vm<6>(); break;
        case 7: wait_vm<7>(); break;
        case 8: wait_vm<8>(); break;
        case 9: wait_vm<9>(); break;
        case 10: wait_vm<10>(); break;
        case 11: wait_vm<11>(); break;
        case 12: wait_vm<12>(); break;
        case 13: wait_vm<13>(); break;
        case 14: wait_vm<14>(); break;
        default: wait_vm<15>(); break;
    }
}
// byte offset of a __shared__ address inside the workgroup's LDS allocation
__device__ __forceinline__ uint32_t lds_offset_of(const void* p) {
    return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) char*)p));
}
constexpr int kDmaStageBytes = 2 * 64 * 16;     // one row of one wave: 64 grad packets, then 64 x packets

struct PcGeom {
    int64_t outer, C, inner, L;
    int64_t wpos;            // positions per window (R == 1) or L (R > 1)
    int64_t n_windows;       // windows per row
    int64_t n_tiles;         // ceil(outer / R): row tiles, dealt out to the splits in balanced contiguous runs
    int64_t rows_per_split;  // the most rows any workgroup walks (multiple of R)
    int32_t splits;          // workgroups along the row axis
    int32_t R;               // rows folded into one tile
    int32_t k_slots;         // channel slots per window (LDS table / partial row length)
    int32_t vec;             // elements per lane per row (IO::VEC or 1)
    int32_t fits32;          // L < 2^31: index divisions in 32 bits
    int32_t ww_lanes;        // row-group windows (make_geom_ww): lanes per row group; 0 otherwise
    int32_t block_threads;   // workgroup size to launch (kBlock except for row-group windows)
    int32_t ring_nt;         // LDS-DMA ring: issue its copies with the streaming hint (global_load_lds ... nt)
    int32_t direct;          // forward, a lane's components are different channels: no LDS table, the lane reads its own scale / shift
    int32_t own;             // OWNER windows (make_geom_own): lanes per row of the owner's run; 0 otherwise
    int32_t own_prio;        // owner windows: the waves of a SIMD take turns at the higher issue priority (bwd_pc_kernel)
    int32_t xcds;            // owner windows: XCDs the owners are dealt over (workgroups go to the XCDs round-robin by blockIdx.x)
#ifdef LSQ_TIMELINE
    unsigned long long* timeline;   // experiment build (tools/exp_timeline.py): 8 x u64 per wave of the window backward
#endif
};

// How many workgroups along the row axis?  `want` is what the caller asked for (workgroups per CU x CUs / windows).
// A kernel whose workgroups all take about the same time finishes in whole "rounds" of what the chip holds at once
// (`resident_blocks`, 0 = unknown): 980 workgroups on a chip that holds 768 take as long as 1536 would -- the VALU-heavy
// 16-bit backward measured exactly that, two rounds of ~19 us with the second three quarters empty.  So among the split
// counts around `want` ([want/2, 2 want]) take the one closest to `want` whose last round is at least 90 % full, or the
// fullest one if there is none.
static inline int64_t pick_splits(int64_t n_windows, int64_t want, int64_t max_splits, int64_t resident_blocks) {
    want = std::max<int64_t>(1, std::min(want, max_splits));
    if (resident_blocks <= 0 || n_windows >= 2 * resident_blocks) return want;      // many rounds anyway: the tail is small
    const int64_t lo = std::max<int64_t>(1, want / 2), hi = std::min(max_splits, 2 * want);
    int64_t best = want;
    double best_fill = -1.0;
    int64_t best_dist = 0;
    for (int64_t sp = lo; sp <= hi; ++sp) {
        const int64_t total = n_windows * sp, rounds = (total + resident_blocks - 1) / resident_blocks;
        const double fill = static_cast<double>(total) / static_cast<double>(rounds * resident_blocks);
        const int64_t dist = sp > want ? sp - want : want - sp;
        const bool good = fill >= 0.9, best_good = best_fill >= 0.9;
        if (best_fill < 0 || (good && !best_good) || (good && best_good && dist < best_dist) ||
            (!good && !best_good && fill > best_fill)) {
            best = sp; best_fill = fill; best_dist = dist;
        }
    }
    return best;
}

// per_slot_rows: rows a workgroup should walk per (slot / lane-position) of per-workgroup overhead.  27 for the
// kernels that write a 16-byte partial per slot (backward, statistics); the forward only rebuilds its channel
// table per workgroup and passes 4.
// resident_blocks: how many workgroups of THIS kernel the chip holds at once (CUs x the occupancy its registers and LDS
// allow), 0 = unknown: the split count is chosen so that the last round of workgroups is nearly full (pick_splits); the
// rows are dealt out evenly (RowWalk).
static inline PcGeom make_geom(int64_t outer, int64_t C, int64_t inner, int vec, int target_blocks, int per_slot_rows = 27,
                               int resident_blocks = 0) {
    PcGeom g;
    g.outer = outer; g.C = C; g.inner = inner; g.L = C * inner; g.vec = vec;
    g.fits32 = (g.L + static_cast<int64_t>(kBlock) * vec) < 0x7fffffffLL ? 1 : 0;
    const int64_t W = static_cast<int64_t>(kBlock) * vec;
    if (g.L >= W) {
        g.R = 1;
        g.wpos = W;
        g.n_windows = (g.L + W - 1) / W;
        g.k_slots = static_cast<int32_t>(std::min<int64_t>(C, (W - 1) / inner + 2));
    } else {
        g.R = static_cast<int32_t>(std::max<int64_t>(1, std::min<int64_t>(W / g.L, outer)));
        g.wpos = g.L;
        g.n_windows = 1;
        g.k_slots = static_cast<int32_t>(C);
    }
    g.n_tiles = (outer + g.R - 1) / g.R;
    g.ww_lanes = 0;
    g.block_threads = kBlock;
    g.ring_nt = 0;
    g.direct = 0;
    g.own = 0;
    g.own_prio = 0;
    g.xcds = 1;
#ifdef LSQ_TIMELINE
    g.timeline = nullptr;
#endif
    // keep the partial-sum traffic (16 B per slot per workgroup) below ~5 % of the streamed bytes -- but never at the price
    // of idle CUs: a tensor too small to give every CU a workgroup of that many tiles is latency- not traffic-bound, and its
    // workgroups take the tiles that one workgroup per CU would ([128, 768]: 64 workgroups of one tile instead of 7 walking
    // 21 rows one group after the other).  A smooth rule: the former switch at 2^21 elements was a cliff (just below it
    // a [2730, 768] bf16 backward ran 1024 two-row workgroups in 21.9 us, just above 170 sixteen-row ones in 13.6 us;
    // profiles/r03_policy_cliffs.txt).
    const int64_t floor_tiles = std::max<int64_t>(1, (per_slot_rows * static_cast<int64_t>(g.k_slots) + W - 1) / W);
    const int64_t spread_tiles = g.n_tiles * g.n_windows / std::max(1, device_info().cu_count);
    const int64_t min_tiles = std::max<int64_t>(1, std::min(floor_tiles, spread_tiles));
    const int64_t max_splits = std::max<int64_t>(1, g.n_tiles / min_tiles);       // every split gets >= min_tiles tiles
    const int64_t want_splits = std::max<int64_t>(1, (target_blocks + g.n_windows - 1) / g.n_windows);
    // whole tiles per workgroup first (ceil), then as many workgroups as that needs: [64,197,768] forward = 3152
    // workgroups of 4 rows, not 4096 of 3-or-4 (every workgroup rebuilds the window's channel table)
    const int64_t tiles_each = std::max<int64_t>((g.n_tiles + want_splits - 1) / want_splits, min_tiles);
    int64_t splits = std::max<int64_t>(1, (g.n_tiles + tiles_each - 1) / tiles_each);
    if (resident_blocks > 0) splits = pick_splits(g.n_windows, splits, max_splits, resident_blocks);
    splits = std::min<int64_t>(splits, 65535);
    g.splits = static_cast<int32_t>(splits);
    g.rows_per_split = (g.n_tiles + splits - 1) / splits * g.R;
    return g;
}

// ROW-GROUP windows for the case "the quantized axis is the last one" (inner == 1, position == channel, L % V == 0).
// A workgroup is R row groups of w lanes; row group r owns row r of every tile of R rows, all over the same w x V channels:
//   * rows of at most 256 lanes (L / V <= 256: NHWC with 256 channels, [tokens, 768]): w = L / V -- the whole row --
//     and R = 256 / w row groups, the workgroup size rounded up to whole waves (L / V = 96 -> 192 threads, none idle);
//   * longer rows ([8192, 4096]): w = 64, one wave per row group, R = 4, ceil(L / 64 V) windows per row.
// A workgroup so has up to four wave64 streams in flight but only w x V partial slots to flush -- a quarter of the
// 256-lane window's ([8192, 4096] bf16: 768 workgroups write 6 MB of partials instead of 25 MB).  The lane's V channels
// are its own: their constants live in registers, there is no LDS table, and the epilogue is a fixed-order sum of the
// row groups through LDS.
static inline PcGeom make_geom_ww(int64_t outer, int64_t C, int vec, int target_blocks, int min_rows, int resident_blocks,
                                  bool split64 = false, int block = kBlock) {
    PcGeom g;
    g.ring_nt = 0;
    g.direct = 0;
    g.own = 0;
    g.own_prio = 0;
    g.xcds = 1;
#ifdef LSQ_TIMELINE
    g.timeline = nullptr;
#endif
    g.outer = outer; g.C = C; g.inner = 1; g.L = C; g.vec = vec;
    g.fits32 = (g.L + static_cast<int64_t>(kBlock) * vec) < 0x7fffffffLL ? 1 : 0;
    const int64_t lanes_per_row = g.L / vec;
    // (block > kBlock: the 1024-lane workgroups of mid-sized tensors -- more row groups per workgroup, never narrower windows)
    if (lanes_per_row <= kBlock && !(block == kBlock && split64 && lanes_per_row >= 128 && lanes_per_row % 64 == 0)) {
        g.ww_lanes = static_cast<int32_t>(lanes_per_row);
        g.R = static_cast<int32_t>(std::max<int64_t>(1, std::min<int64_t>(block / lanes_per_row, outer)));
        g.n_windows = 1;
    } else {
        g.ww_lanes = 64;
        g.R = block / 64;
        g.n_windows = (lanes_per_row + 63) / 64;
    }
    g.block_threads = static_cast<int32_t>((static_cast<int64_t>(g.R) * g.ww_lanes + 63) / 64 * 64);
    g.wpos = static_cast<int64_t>(g.ww_lanes) * vec;
    g.k_slots = static_cast<int32_t>(g.wpos);
    g.n_tiles = (outer + g.R - 1) / g.R;
    // (the rows-per-workgroup floor gives way to "one workgroup per CU first" for small tensors: see make_geom)
    const int64_t floor_tiles = std::max<int64_t>(1, (min_rows + g.R - 1) / g.R);
    const int64_t spread_tiles = g.n_tiles * g.n_windows / std::max(1, device_info().cu_count);
    const int64_t min_tiles = std::max<int64_t>(1, std::min(floor_tiles, spread_tiles));
    const int64_t max_splits = std::max<int64_t>(1, g.n_tiles / min_tiles);
    int64_t splits = std::max<int64_t>(1, (target_blocks + g.n_windows - 1) / g.n_windows);
    splits = std::min(splits, max_splits);
    if (resident_blocks > 0) splits = pick_splits(g.n_windows, splits, max_splits, resident_blocks);
    splits = std::min<int64_t>(splits, 65535);
    g.splits = static_cast<int32_t>(splits);
    g.rows_per_split = (g.n_tiles + splits - 1) / splits * g.R;
    return g;
}

// OWNER windows (backward, round 4): a workgroup OWNS k whole channels -- a run of k * inner contiguous positions of every
// row -- for ALL rows, so its channel sums are final: no partials, no finalize launch (on a 25 M-element activation that
// launch was 5-6 us of a 36 us backward).  k is the smallest channel count whose run is whole 16-byte packets; the workgroup
// is R row slots of run / V lanes each (lane t: row slot t / lanes, packet t % lanes of the run), row slot r takes row r of
// every tile of R rows -- the lane keeps its packet column, hence its channel(s), for its whole life like in every other
// window mode.  Neighbouring owners share a 128-byte line at each end of their runs (784-byte runs at BASELINE config 5),
// so the owner index is dealt XCD-major (own_window): consecutive runs meet in one XCD's L2 instead of being fetched twice
// (tools/exp_owner_probe.py, profiles/r04_owner_pattern_probe.txt: 28.1 us against 31.4 without, 29-31 for 4 KiB windows).
struct OwnPlan {
    int k;                // channels per owner (0: the shape does not take owner windows)
    int lanes_per_row;    // run / V
    int R;                // row slots per workgroup
    int block_threads;    // R * lanes_per_row rounded up to whole waves
    int per_cu;           // owners resident per CU the plan was sized for
    int run_bytes;        // bytes of one row an owner reads (k * inner * element size)
};
// min_run_bytes: the shortest run an owner may have (a run is what a workgroup reads of ONE row); the plan grows the channel
// group until the run is that long.  The default is no minimum: short runs are fine on small tensors ([64,1024,5,5] fp32,
// 400-byte runs: 11.3 -> 8.1 us; [256,2048,7] fp32, 112-byte runs: 16.7 -> 14.1 us) and it is the launch policy that keeps
// them off larger ones (lsq_per_channel.hip, kOwnMaxElemsShortRun; profiles/r04_owner_min_run.txt).  Tools builds can set one.
constexpr int kOwnMinRunBytes = 1;
constexpr int kOwnFatDefault = 1;
// one candidate: owners of k channels
static inline OwnPlan plan_own_k(int64_t k, int64_t outer, int64_t C, int64_t inner, int vec, int elem_bytes, int dma_depth, int cus,
                                 int block_limit) {
    OwnPlan o{0, 0, 0, 0, 0, 0};
    const int64_t lanes = k * inner / vec;
    if (lanes > 256) return o;                                     // long channel rows: the 256-lane windows / segment walk
    const int64_t owners = C / k;
    const int per_cu = static_cast<int>((owners + cus - 1) / cus);
    if (owners < (3 * static_cast<int64_t>(cus)) / 4 || per_cu > 4) return o;   // idle CUs / more than one round of fat workgroups
    // waves a CU can hold per owner: 16 of the 1024-lane launch bound, and the ring's LDS (dma_depth stages of 2 KiB per wave)
    // (the front area -- channel table, one pair of fp64 slots per channel and wave, staged raw parameters -- of a full
    //  workgroup of this owner: 32 + 16 * 16 + 8 bytes per channel at most, rounded to 1 KiB like bwd_lds_front_bytes)
    const int front = static_cast<int>((k * (32 + 16 * 16 + 8) + 1023) / 1024 * 1024);
    const int lds_waves = static_cast<int>(((160 * 1024) / per_cu - front) / (dma_depth * kDmaStageBytes));
    const int max_lanes = std::min(block_limit, std::min(1024 / per_cu, lds_waves * 64)) / 64 * 64;
    const int r_max = static_cast<int>(std::min<int64_t>(max_lanes / lanes, outer));
    if (r_max < 2) return o;
    // a divisor of the row count close to the maximum: every lane then walks the same number of rows.  None (58 rows = 2 x 29
    // against 10 row slots): all the slots, and a short last tile -- the kernel's blocks stop one tile early and the rest is
    // walked row by row (bwd_pc_kernel, `ragged`).
    int R = r_max;
    for (int d = r_max; d * 5 >= r_max * 3; --d)
        if (outer % d == 0) { R = d; break; }
    // the ring wants 2 x its depth of row tiles per lane: few rows -> fewer row slots (a thinner workgroup), down to two
    while (R > 2 && (outer + R - 1) / R < 2 * dma_depth) {
        int next = R - 1;
        for (int d = R - 1; d >= 2 && d * 5 >= (R - 1) * 3; --d)
            if (outer % d == 0) { next = d; break; }
        R = next;
    }
    if ((outer + R - 1) / R < 2 * dma_depth) return o;              // too few rows per lane to run the ring
    o.k = static_cast<int>(k); o.lanes_per_row = static_cast<int>(lanes); o.R = R; o.per_cu = per_cu;
    o.block_threads = static_cast<int>((R * lanes + 63) / 64 * 64);
    o.run_bytes = static_cast<int>(k * inner * elem_bytes);
    return o;
}
// fat: among the channel groups that work, 0 = the smallest (most owners), 1 = the LARGEST that still gives every CU an owner
// (fewer, fatter owners with longer runs; the smallest when none does)
static inline OwnPlan plan_own(int64_t outer, int64_t C, int64_t inner, int vec, int elem_bytes, int dma_depth, int cus,
                               int block_limit = 512, int min_run_bytes = kOwnMinRunBytes, int fat = kOwnFatDefault) {
    OwnPlan first{0, 0, 0, 0, 0, 0}, best{0, 0, 0, 0, 0, 0};
    if (vec <= 1 || inner < vec) return first;                     // (inner < V: the row-group / last-axis kernels' ground)
    // channel counts whose run is whole packets, divides C, is long enough and no wider than 256 lanes
    // (fewer than 3/4 CUs' worth of owners from some kk on: nothing beyond it can work -- at most C / (3/4 CUs) candidates)
    for (int64_t kk = 1; kk <= C && kk * inner <= static_cast<int64_t>(256) * vec && 4 * (C / kk) >= 3 * static_cast<int64_t>(cus); ++kk) {
        if ((kk * inner) % vec != 0 || C % kk != 0 || kk * inner * elem_bytes < min_run_bytes) continue;
        const OwnPlan o = plan_own_k(kk, outer, C, inner, vec, elem_bytes, dma_depth, cus, block_limit);
        if (o.k == 0) continue;
        if (first.k == 0) first = o;
        if (!fat) break;
        if (C / kk >= cus) best = o;
    }
    return (fat && best.k) ? best : first;
}
static inline PcGeom make_geom_own(int64_t outer, int64_t C, int64_t inner, int vec, const OwnPlan& o) {
    PcGeom g;
    g.outer = outer; g.C = C; g.inner = inner; g.L = C * inner; g.vec = vec;
    g.fits32 = (g.L + static_cast<int64_t>(1024) * vec) < 0x7fffffffLL ? 1 : 0;
    g.wpos = static_cast<int64_t>(o.k) * inner;
    g.n_windows = C / o.k;
    g.R = o.R;
    g.k_slots = o.k;
    g.n_tiles = (outer + o.R - 1) / o.R;
    g.splits = 1;
    g.rows_per_split = g.n_tiles * o.R;
    g.ww_lanes = 0;
    g.block_threads = o.block_threads;
    g.ring_nt = 0;
    g.direct = 0;
    g.own = o.lanes_per_row;
    g.own_prio = 1;
    // HIP does not expose the XCD count; gfx950 parts have 32 CUs per XCD (MI355X: 256 CUs = 8 XCDs).  A wrong guess only
    // costs the shared-cache-line benefit of dealing neighbouring owners to one XCD, never correctness.
    g.xcds = std::max(1, std::min(8, device_info().cu_count / 32));
#ifdef LSQ_TIMELINE
    g.timeline = nullptr;
#endif
    return g;
}
// the owner a workgroup serves: blockIdx.x dealt XCD-major (workgroups go to the XCDs round-robin: blockIdx.x % xcds is the
// XCD), so that owners j and j + 1 -- whose runs share a cache line -- sit on the same XCD
__device__ __forceinline__ int64_t own_window(const PcGeom& g) {
    const uint32_t n = gridDim.x, k = static_cast<uint32_t>(g.xcds);
    const uint32_t x = blockIdx.x % k, i = blockIdx.x / k;
    const uint32_t per = n / k, rem = n % k;
    return static_cast<int64_t>(x * per + (x < rem ? x : rem) + i);
}

// Dynamic LDS of the window-mode backward in front of the LDS-DMA ring: the channel table + fp64 slots of the 256-lane
// windows (the ring starts at the next 1 KiB boundary).  Row-group windows have nothing in front: their combine buffer
// is only used after the last row has been consumed and takes the ring's place (one barrier in between).
// (+ 8 bytes per slot: the raw scale / shift of the window's channels, staged by LDS-DMA before the row copies are issued)
// (owner windows keep one pair of fp64 slots per channel AND WAVE: their sums are final, so the waves' contributions are added
//  in wave order at the end instead of by LDS atomics in arrival order -- bit-reproducible d_scale / d_shift / wide)
static inline __host__ __device__ uint32_t bwd_lds_sum_sets(const PcGeom& g) {
    return g.own ? static_cast<uint32_t>(g.block_threads) / 64u : 1u;
}
static inline __host__ __device__ uint32_t bwd_lds_front_bytes(const PcGeom& g, uint32_t slot_bytes) {
    if (g.ww_lanes) return 0u;
    return (static_cast<uint32_t>(g.k_slots) * (slot_bytes + 16u * bwd_lds_sum_sets(g) + 8u) + 1023u) & ~1023u;
}

// Where a lane sits: position p0 of its first element, its row inside the tile, and whether it is live.
struct LaneSite {
    int64_t p0;
    int32_t row_in_tile;
    bool live;
    int64_t c_lo;  // first channel of the window
    bool counts;   // the lane's sums count (== live, except the stand-in lanes of owner windows: they walk, store duplicates, add nothing)
};
__device__ __forceinline__ LaneSite lane_site(const PcGeom& g, int V) {
    LaneSite s;
    const int64_t idx = static_cast<int64_t>(threadIdx.x) * V;
    const bool f32 = g.fits32 != 0;
    if (g.R == 1) {
        const int64_t base = static_cast<int64_t>(blockIdx.x) * g.wpos;
        s.p0 = base + idx;
        s.row_in_tile = 0;
        s.live = s.p0 < g.L;
        s.c_lo = udiv(base, g.inner, f32);
    } else {
        s.row_in_tile = static_cast<int32_t>(udiv(idx, g.L, f32));
        s.p0 = idx - static_cast<int64_t>(s.row_in_tile) * g.L;
        s.live = s.row_in_tile < g.R;
        s.c_lo = 0;
    }
    s.counts = s.live;
    return s;
}

// owner windows: thread t = packet (t % lanes) of row slot (t / lanes) of owner own_window()'s run
__device__ __forceinline__ LaneSite lane_site_own(const PcGeom& g, int V) {
    LaneSite s;
    const uint32_t lanes = static_cast<uint32_t>(g.own);
    const uint32_t slot = threadIdx.x / lanes;
    const int64_t j = own_window(g);
    s.p0 = j * g.wpos + static_cast<int64_t>(threadIdx.x - slot * lanes) * V;
    // The lanes past the last row slot (the workgroup is whole waves) are STAND-INS of the last row slot's lanes: they walk its
    // rows and compute the very same values a second time; their stores are masked off and their sums dropped (a duplicate
    // store is not merged with the original's: measured +10 % HBM write traffic).  Every wave so runs the loop's fast
    // form (all lanes valid for all rows); with idle lanes the last wave took the generic row-at-a-time loop, whose waits are
    // longer -- and an owner workgroup, alone on its CU for the whole launch, is as slow as its slowest wave.
    s.counts = static_cast<int32_t>(slot) < g.R;
    s.row_in_tile = s.counts ? static_cast<int32_t>(slot) : g.R - 1;
    s.live = true;
    s.c_lo = j * g.k_slots;
    return s;
}

// row-group windows: thread t = lane (t % w) of row group (t / w); `lane_in_group` comes back through c_lo's neighbour
__device__ __forceinline__ LaneSite lane_site_ww(const PcGeom& g, int V, int32_t& lane_in_group) {
    LaneSite s;
    const uint32_t w = static_cast<uint32_t>(g.ww_lanes);
    const uint32_t rg = threadIdx.x / w;
    lane_in_group = static_cast<int32_t>(threadIdx.x - rg * w);
    const int64_t base = static_cast<int64_t>(blockIdx.x) * g.wpos;
    s.p0 = base + static_cast<int64_t>(lane_in_group) * V;
    s.row_in_tile = static_cast<int32_t>(rg);
    s.live = static_cast<int32_t>(rg) < g.R && s.p0 < g.L;
    s.counts = s.live;
    s.c_lo = base;     // inner == 1: position == channel
    return s;
}

// The rows a lane walks: o_begin, o_begin + step, ... (n_rows of them)
struct RowWalk {
    int64_t o_begin, step, n_rows;
    int64_t n_tiles_split;      // tiles of this workgroup's split: the same for every lane (n_rows is per lane)
    __device__ __forceinline__ RowWalk(const PcGeom& g, const LaneSite& site) {
        // (dealing the tiles round-robin instead -- one compact advancing band of rows per launch -- was measured: within +-5 %
        // on the backward, +9 .. +16 % on two forwards, profiles/r03_row_interleave_ab.txt; the code is in the history)
        // split y of `splits` owns the row tiles [y * n_tiles / splits, (y + 1) * n_tiles / splits): sizes differ by at
        // most one tile (a uniform ceil(n / splits) leaves the last workgroup a short remainder and the rest too much)
        const int64_t t0 = static_cast<int64_t>(blockIdx.y) * g.n_tiles / g.splits;
        const int64_t t1 = static_cast<int64_t>(blockIdx.y + 1) * g.n_tiles / g.splits;
        n_tiles_split = t1 - t0;
        o_begin = t0 * g.R + site.row_in_tile;
        const int64_t o_end = std::min<int64_t>(g.outer, t1 * g.R);
        step = g.R;
        n_rows = (site.live && o_begin < o_end) ? (o_end - o_begin + step - 1) / step : 0;
    }
    __device__ __forceinline__ int64_t row(int64_t i) const { return o_begin + i * step; }
};

// ---- SEGMENT mode: one channel per workgroup (few rows, long packet-aligned channel rows) ----------
struct SegGeom {
    int64_t outer, C, inner;
    int64_t n_sub;        // sub-rows (of W positions) per channel row
    int64_t sub_per_seg;  // sub-rows one workgroup owns
    int64_t o_per_split;  // outer indices one workgroup owns
    int32_t segs;         // workgroups per channel row
    int32_t osplits;      // workgroups along outer
};

static inline SegGeom make_seg_geom(int64_t outer, int64_t C, int64_t inner, int vec, int target_blocks) {
    SegGeom g;
    g.outer = outer; g.C = C; g.inner = inner;
    const int64_t W = static_cast<int64_t>(kBlock) * vec;
    g.n_sub = (inner + W - 1) / W;
    const int64_t per_channel = std::max<int64_t>(1, target_blocks / C);       // workgroups we would like per channel
    const int64_t iters = g.n_sub * outer;                                    // lane iterations per channel
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(per_channel, iters / 4));  // >= 4 packets per lane
    int64_t segs = std::min<int64_t>(g.n_sub, blocks);
    g.sub_per_seg = (g.n_sub + segs - 1) / segs;
    g.segs = static_cast<int32_t>((g.n_sub + g.sub_per_seg - 1) / g.sub_per_seg);
    int64_t osplits = std::max<int64_t>(1, std::min<int64_t>(outer, blocks / g.segs));
    g.o_per_split = (outer + osplits - 1) / osplits;
    g.osplits = static_cast<int32_t>((outer + g.o_per_split - 1) / g.o_per_split);
    return g;
}

// The (o, sub-row) pairs a workgroup walks, flattened: it -> (o_begin + it / n_r, r_begin + it % n_r)
struct SegWalk {
    int64_t c, o_begin, r_begin, n_r, n_it;
    // a whole channel by one workgroup (segs == osplits == 1): the multi-tensor kernels, channel `ch` of their item
    __device__ __forceinline__ SegWalk(const SegGeom& g, int64_t ch) : c(ch), o_begin(0), r_begin(0), n_r(g.n_sub), n_it(g.n_sub * g.outer) {}
    __device__ __forceinline__ SegWalk(const SegGeom& g) {
        c = blockIdx.x / g.segs;
        const int64_t seg = blockIdx.x - c * g.segs;
        r_begin = seg * g.sub_per_seg;
        const int64_t r_end = std::min<int64_t>(g.n_sub, r_begin + g.sub_per_seg);
        o_begin = static_cast<int64_t>(blockIdx.y) * g.o_per_split;
        const int64_t o_end = std::min<int64_t>(g.outer, o_begin + g.o_per_split);
        n_r = r_end - r_begin;
        n_it = n_r * (o_end - o_begin);
    }
};

// few rows + packet-aligned channel rows -> one channel per workgroup (conv / linear weights on axis 0).
// Rows of at least one workgroup's span (W = 256 lanes x 16 bytes) always.  `short_rows_cus` > 0 (the forward and the
// backward op; the CU count) adds SHORT rows, where part of the workgroup idles but the walk still wins because it needs no
// partials and no finalize launch (profiles/r03_seg_min_ab.txt, forward / backward op against the window kernels):
//   up to 8 workgroups per CU, rows of at least W/8 (up to 16: W/2):  [768,768] fp32 -14 % / -47 %, [64,64,3,3] -20 % / -50 %,
//       [3072,768] bf16 -1 % / -5 %, [1000,512] bf16 -1 % / -27 %;
//   any channel count, rows of at least 3W/4:  [50257,768] fp32 -13 % / -33 %;
//   not shorter rows on many channels: [32768,256] fp32 +60 % / +50 %, [16384,768] bf16 (3W/8) +18 % / +53 %.
static inline bool pick_segment_mode(int vec, int64_t outer, int64_t C, int64_t inner, int short_rows_cus = 0, bool fused = false) {
    if (vec == 1 || inner % vec != 0) return false;
    const int64_t W = static_cast<int64_t>(kBlock) * vec;
    if (outer >= 8 || C * ((inner + W - 1) / W) > 0x7fffffffLL) return false;
    if (inner >= W) return true;
    if (short_rows_cus <= 0) return false;
    const int div = knob::get(knob::kSegMinDiv);     // tools builds, lsq_hip_debug_set_seg_min_div: rows of at least W/div
    if (div > 0) return inner * div >= W;
    // (round 4, profiles/r04_seg_weights.txt: 48 weight shapes x two storage types -- the walk is 40-57 % ahead on nearly all of
    //  them, but at 16 workgroups per CU rows under W/2 are behind in 16-bit storage: [4096,288] +32 %, [4096,576] +18 %,
    //  [4096,768] +11 %, level from [4096,1024] on; at 8 per CU still ahead, [2048,288] -11 %.  So W/8 holds up to 8 per CU.)
    //  `fused`: the question is asked for the multi-tensor launch (lsq_multi.hip: the tensor is one of many in ONE launch, which
    //  is worth more than the family choice of a single call) -- W/8 up to 16 per CU as before.
    if (C <= static_cast<int64_t>(short_rows_cus) * (fused ? 16 : 8)) return inner * 8 >= W;
    if (C <= static_cast<int64_t>(short_rows_cus) * 16) return inner * 2 >= W;
    return inner * 4 >= 3 * W;
}

static inline bool grid_fits(const SegGeom& g) { return g.C * g.segs <= 0x7fffffffLL && g.osplits <= 65535; }
static inline int pick_vec(int io_vec, int64_t L, bool aligned) { return (aligned && (L % io_vec) == 0) ? io_vec : 1; }
static inline bool grid_fits(const PcGeom& g) { return g.n_windows <= 0x7fffffffLL && g.splits <= 65535 && g.splits >= 1; }

}  // namespace lsq
