// lsq_per_channel.hip -- K3 (forward) and K4 (fused backward + per-channel reduction) on gfx950.
//
// Replaces the reference's per-channel CUDA backend (/root/reference/torchlsq/csrc/ops/cuda/lsq_cuda.cu:147-297):
// TensorIterator-broadcast scale/shift with a division per element (lsq_kernel.h:157-158), three
// elementwise backward kernels, three N-sized temporaries and two `sum(axes != axis)`.
//
// Data view: dense memory as [outer][C][inner]; a "row" is the L = C*inner elements of one outer
// index, position p of a row belongs to channel p / inner.
//
// CDNA4 design: CHANNEL-STATIONARY LANES.  A lane keeps the same channel(s) for its whole life, so
// the per-channel constants {s, 1/s, zp} sit in registers, no per-element index arithmetic or
// division is left in the loop (the reference divides per element), and every wave instruction still
// moves 1 KiB of contiguous HBM.  Two work decompositions, chosen on the host:
//
//  * WINDOW mode (many rows: activations, [batch, features], channels-last).  A workgroup owns a
//    window of W = 256 lanes x V positions of the row and walks down a slab of rows; short rows
//    (L < W) fold R = W/L rows into one tile.  The window's channel table is computed ONCE per
//    workgroup into LDS and fanned out to the lanes (1 lane = 1, 2 or V channels, template CPL).
//    d_scale/d_shift: fp64 lane accumulators -> segmented wave64 shuffle reduction keyed by runs of
//    equal channel -> LDS fp64 atomics (ds_add_f64) on the window's channel slots -> one partial per
//    (workgroup, slot) -> fixed-order finalize.
//  * SEGMENT mode (few rows, long channels: conv/linear weights on axis 0).  A workgroup owns a
//    segment of ONE channel row (sub-rows of W positions) for a range of outer indices: one channel
//    per workgroup, its constants computed in registers, reduction = wave64 butterfly + 4 partials
//    through LDS -> one partial per workgroup -> fixed-order finalize.
//
// Loads are never predicated (a predicated version serialised them behind s_waitcnt) and no slot of a load group
// is padding: a walk of n rows is cut into groups of UNROLL, then UNROLL/2, ..., 1 rows.  Only the software-pipelined
// loop of the 16-bit backward, whose prefetch can run past the end, re-reads the last row (clamped address, served
// by L2) and masks its effects.  No global atomics, no zero-initialised buffers.
#include "lsq_kernels.hpp"
#include "lsq_pc_geom.hpp"
#include "lsq_seg_body.hpp"

namespace lsq {

// =================================================================================================
// shared pieces
// =================================================================================================
template <typename T>
struct alignas(16) QSlot {  // LDS image of one channel's constants
    T s, inv_s, zp, pad;
};

// =================================================================================================
// WINDOW mode
// =================================================================================================
// Build the window's channel table in LDS (lsq_kernel.h:157-158 + :12, once per channel).
template <typename T>
__device__ __forceinline__ void build_channel_table(QSlot<T>* table, int k_count, int64_t c_lo, int64_t C,
                                                    const T* __restrict__ scale, const T* __restrict__ shift,
                                                    const Range<T>& r) {
    for (int k = threadIdx.x; k < k_count; k += kBlock) {
        const int64_t c = c_lo + k;
        QSlot<T> e;
        if (c < C) {
            const QParams<T> q = make_qparams<T>(sanitize_scale_per_channel<T>(scale[c]), shift[c], r);
            e.s = q.s; e.inv_s = q.inv_s; e.zp = q.zp; e.pad = static_cast<T>(0);
        } else {
            e.s = static_cast<T>(1); e.inv_s = static_cast<T>(1); e.zp = static_cast<T>(0); e.pad = static_cast<T>(0);
        }
        table[k] = e;
    }
}

// The same in two steps, for kernels that put their first rows in flight before the table exists: the scale / shift loads
// are ISSUED first (vector-memory operations retire in issue order: a wait for a load behind the rows' loads would be a wait
// for the rows), the row loads follow, and the table is finished when its raw values are needed.  Up to kRawSlots table
// slots per thread travel in registers; wider windows (last-axis windows of more than 512 channels) use build_channel_table.
constexpr int kRawSlots = 2;
template <typename T>
struct ChannelRaw {
    T s[kRawSlots], b[kRawSlots];
};
template <typename T>
__device__ __forceinline__ ChannelRaw<T> load_channel_raw(int k_count, int64_t c_lo, int64_t C, const T* __restrict__ scale,
                                                          const T* __restrict__ shift) {
    ChannelRaw<T> raw;
#pragma unroll
    for (int i = 0; i < kRawSlots; ++i) {
        const int k = threadIdx.x + i * kBlock;
        int64_t c = c_lo + k;
        c = (k < k_count && c < C) ? c : (C - 1);        // (a valid address for the lanes without a slot: the value is unused)
        raw.s[i] = scale[c];
        raw.b[i] = shift[c];
    }
    return raw;
}
template <typename T>
__device__ __forceinline__ void finish_channel_table(QSlot<T>* table, int k_count, int64_t c_lo, int64_t C, const ChannelRaw<T>& raw,
                                                     const Range<T>& r) {
#pragma unroll
    for (int i = 0; i < kRawSlots; ++i) {
        const int k = threadIdx.x + i * kBlock;
        if (k < k_count) {
            QSlot<T> e;
            if (c_lo + k < C) {
                const QParams<T> q = make_qparams<T>(sanitize_scale_per_channel<T>(raw.s[i]), raw.b[i], r);
                e.s = q.s; e.inv_s = q.inv_s; e.zp = q.zp; e.pad = static_cast<T>(0);
            } else {
                e.s = static_cast<T>(1); e.inv_s = static_cast<T>(1); e.zp = static_cast<T>(0); e.pad = static_cast<T>(0);
            }
            table[k] = e;
        }
    }
}
// first channel of workgroup blockIdx.x's window (LaneSite::c_lo without the per-lane part)
__device__ __forceinline__ int64_t window_first_channel(const PcGeom& g) {
    if (g.own) return own_window(g) * g.k_slots;
    return g.R == 1 ? udiv(static_cast<int64_t>(blockIdx.x) * g.wpos, g.inner, g.fits32 != 0) : 0;
}

// Where an OWNER-window backward stores its channels' finished sums (d_scale / d_shift, rounded once; wide: un-rounded).
template <typename T>
struct PcDirect {
    T* ds;
    T* db;
    double* wide;
    T sym_term;         // the constant per-element d_shift term of the symmetric case, 0 * grad_scaler (lsq_kernel.h:118,122)
    int32_t sym;
};

// CPL = channels a lane can touch: 1 (inner % V == 0), 2 (inner >= V), V (anything).
template <typename T, int V, int CPL>
struct LaneChannels {
    static constexpr int N = (CPL == 1) ? 1 : (CPL == 2 ? 2 : V);
    QParams<T> q[N];
    int32_t key[N];   // slot index in the window table
    int32_t split;    // CPL == 2: components j >= split belong to q[1]
    __device__ __forceinline__ void init(const QSlot<T>* table, const LaneSite& s, const PcGeom& g) {
        // dead lanes (past the row end / beyond the tile rows) point at slot 0 and never accumulate.
        // Everything is computed into scalars first so the struct stays in registers.
        const bool f32 = g.fits32 != 0;
        const int64_t p0 = s.live ? s.p0 : s.c_lo * g.inner;
        const int64_t c0 = udiv(p0, g.inner, f32);
        int32_t sp = V;
        if (CPL == 2) {
            const int64_t left = (c0 + 1) * g.inner - p0;  // elements of channel c0 from p0 on
            sp = (s.live && left < V) ? static_cast<int32_t>(left) : V;
        }
        split = sp;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            int32_t k;
            if (N == 1 || j == 0) k = static_cast<int32_t>(c0 - s.c_lo);
            else if (CPL == 2) k = static_cast<int32_t>(c0 - s.c_lo) + (sp < V ? 1 : 0);
            else k = s.live ? static_cast<int32_t>(udiv(p0 + j, g.inner, f32) - s.c_lo) : 0;
            key[j] = k;
            const QSlot<T> e = table[k];
            q[j].s = e.s; q[j].inv_s = e.inv_s; q[j].zp = e.zp;
        }
    }
    // CPL == V without a table: a lane whose V components are V (mostly) different channels -- the quantized axis is the last
    // or nearly the last one -- reads ITS channels' scale / shift itself.  A 256-lane window then shares nothing through the
    // table (2048 channels, 2048 lanes' worth of slots), so building one is pure latency: global loads -> divisions -> LDS
    // writes -> barrier -> LDS reads.  Two steps, like load_channel_raw / finish_channel_table: the loads are issued before
    // the first rows' loads, the divisions happen when the rows are in flight.
    __device__ __forceinline__ void load_direct(const T* __restrict__ scale, const T* __restrict__ shift, const LaneSite& s,
                                                const PcGeom& g, T (&rs)[N], T (&rb)[N]) {
        const bool f32 = g.fits32 != 0;
        const int64_t p0 = s.live ? s.p0 : s.c_lo * g.inner;
        if constexpr (CPL == V && V > 2) {
            split = V;
            const bool wide = g.inner == 1 && p0 + V <= g.C &&
                              ((reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift)) & 15u) == 0;
            if (wide) {      // V consecutive channels from a multiple of V on: 16-byte loads
                struct alignas(16) Pack { T v[N]; };
                const Pack a = *reinterpret_cast<const Pack*>(scale + p0);
                const Pack b = *reinterpret_cast<const Pack*>(shift + p0);
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    key[j] = static_cast<int32_t>(p0 + j - s.c_lo);
                    rs[j] = a.v[j];
                    rb[j] = b.v[j];
                }
                return;
            }
#pragma unroll
            for (int j = 0; j < N; ++j) {
                int64_t c = udiv(p0 + j, g.inner, f32);
                c = c < g.C ? c : g.C - 1;
                key[j] = static_cast<int32_t>(c - s.c_lo);
                rs[j] = scale[c];
                rb[j] = shift[c];
            }
        } else {
            // one channel, or two with a split point (init() above, from global memory instead of the table)
            const int64_t c0 = udiv(p0, g.inner, f32);
            int32_t sp = V;
            if (CPL == 2) {
                const int64_t left = (c0 + 1) * g.inner - p0;
                sp = (s.live && left < V) ? static_cast<int32_t>(left) : V;
            }
            split = sp;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                int64_t c = c0 + ((j > 0 && sp < V) ? 1 : 0);
                c = c < g.C ? c : g.C - 1;
                key[j] = static_cast<int32_t>(c - s.c_lo);
                rs[j] = scale[c];
                rb[j] = shift[c];
            }
        }
    }
    __device__ __forceinline__ void finish_direct(const T (&rs)[N], const T (&rb)[N], const Range<T>& r) {
#pragma unroll
        for (int j = 0; j < N; ++j) q[j] = make_qparams<T>(sanitize_scale_per_channel<T>(rs[j]), rb[j], r);
    }
    // constants of component j, by select (never a runtime-indexed register array -> no scratch)
    __device__ __forceinline__ QParams<T> params(int j) const {
        if (N == 1) return q[0];
        if (CPL == 2) {
            const bool hi = j >= split;
            QParams<T> o;
            o.s = hi ? q[N - 1].s : q[0].s;
            o.inv_s = hi ? q[N - 1].inv_s : q[0].inv_s;
            o.zp = hi ? q[N - 1].zp : q[0].zp;
            return o;
        }
        return q[j < N ? j : 0];
    }
    // constants of the component pair (2 pr, 2 pr + 1), fp32 arithmetic: what backward_pair / forward_pair take.  Built
    // field by field from scalars (selects for the two-channel form) -- once, before the row loop.
    __device__ __forceinline__ QPair pair(int pr) const {
        static_assert(std::is_same<T, float>::value || N == 0, "pairs are an fp32 construct");
        QPair o;
        if (N == 1) {
            o.s = f2{q[0].s, q[0].s}; o.inv_s = f2{q[0].inv_s, q[0].inv_s}; o.zp = f2{q[0].zp, q[0].zp};
        } else if (CPL == 2) {
            const bool h0 = 2 * pr >= split, h1 = 2 * pr + 1 >= split;
            o.s = f2{h0 ? q[N - 1].s : q[0].s, h1 ? q[N - 1].s : q[0].s};
            o.inv_s = f2{h0 ? q[N - 1].inv_s : q[0].inv_s, h1 ? q[N - 1].inv_s : q[0].inv_s};
            o.zp = f2{h0 ? q[N - 1].zp : q[0].zp, h1 ? q[N - 1].zp : q[0].zp};
        } else {
            const int a = 2 * pr < N ? 2 * pr : 0, b = 2 * pr + 1 < N ? 2 * pr + 1 : 0;
            o.s = f2{q[a].s, q[b].s}; o.inv_s = f2{q[a].inv_s, q[b].inv_s}; o.zp = f2{q[a].zp, q[b].zp};
        }
        return o;
    }
};

// ------------------------------------------------------------------------------------------------
// K3 (window mode): forward
// ------------------------------------------------------------------------------------------------
// DMA > 0: the rows arrive through an LDS-DMA ring of DMA stages per wave (see bwd_pc_kernel): DMA rows in flight per
// wave and no load registers.
template <typename IO, int V, int CPL, bool INIT, bool LEVELS, int UNROLL, bool NTL, bool NTS, int DMA = 0>
__global__ __launch_bounds__(kBlock) void fwd_pc_kernel(const void* __restrict__ x, void* __restrict__ y,
                                                        int8_t* __restrict__ levels, int level_bias, int aux_kind, PcGeom g,
                                                        const typename IO::arith* __restrict__ scale,
                                                        const typename IO::arith* __restrict__ shift,
                                                        Range<typename IO::arith> r) {
    using T = typename IO::arith;
    using E = typename IO::elem;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    QSlot<T>* table = reinterpret_cast<QSlot<T>*>(smem);

    // no table at all (LaneChannels::load_direct) where a lane's components are different channels: forward_per_channel's choice
    using LC = LaneChannels<T, V, CPL>;
#ifdef LSQ_TOOLS     // (knob 4: also lanes of one or two channels -- measured neutral, +-2 %, profiles/r03_fwd_direct_ab.txt)
    constexpr bool kDirectAble = DMA == 0;
#else
    constexpr bool kDirectAble = CPL == V && V > 2 && DMA == 0;     // (V == 2: CPL == 2 is the two-channel form)
#endif
    const bool direct = kDirectAble && g.direct != 0;
    // the window's raw scale / shift first (issue order = retirement order), then the first rows, then the table
    const bool raw_first = !direct && g.k_slots <= kRawSlots * kBlock;
    ChannelRaw<T> raw;
    if (raw_first) raw = load_channel_raw<T>(g.k_slots, window_first_channel(g), g.C, scale, shift);
    const LaneSite site = lane_site(g, V);
    const RowWalk walk(g, site);
    LC ch;
    T direct_s[LC::N], direct_b[LC::N];
    if constexpr (kDirectAble) {
        if (direct) ch.load_direct(scale, shift, site, g, direct_s, direct_b);
    }
    // the first group of loads does not depend on the channel constants: put it in flight before the
    // table build (a division + a barrier) so the two latencies overlap
    E first[UNROLL][V];
    const bool first_full = DMA > 0 ? false : walk.n_rows >= UNROLL;
    if (first_full) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) load_elems<IO, V, NTL>(x, walk.row(u) * g.L + site.p0, first[u]);
    }
    // ---- LDS-DMA ring (DMA > 0): this wave's DMA stages of 64 x packets ----
    static_assert(DMA == 0 || V * sizeof(E) == 16, "the LDS-DMA ring moves 16-byte packets");
    constexpr int kStage = 64 * 16;
    const int64_t dma_n = walk.n_tiles_split;
    const uint32_t front = (static_cast<uint32_t>(g.k_slots) * static_cast<uint32_t>(sizeof(QSlot<T>)) + 1023u) & ~1023u;
    unsigned char* ring = smem + front + (threadIdx.x >> 6) * (DMA * kStage);
    const uint32_t ring_lds = DMA > 0 ? __builtin_amdgcn_readfirstlane(lds_offset_of(ring)) : 0u;
    auto dma_issue = [&](int64_t i) {
        int64_t row = walk.row(i);
        row = row < g.outer ? row : g.outer - 1;
        const int64_t e = row * g.L + (site.live ? site.p0 : 0);
        glds16_rt(static_cast<const E*>(x) + e, ring_lds + static_cast<uint32_t>(i % (DMA > 0 ? DMA : 1)) * kStage, g.ring_nt);
    };
    if constexpr (DMA > 0) {
        for (int64_t i = 0; i < DMA && i < dma_n; ++i) dma_issue(i);
    }
    if (direct) {
        if constexpr (kDirectAble) ch.finish_direct(direct_s, direct_b, r);
    } else {
        if (raw_first) finish_channel_table<T>(table, g.k_slots, site.c_lo, g.C, raw, r);
        else build_channel_table<T>(table, g.k_slots, site.c_lo, g.C, scale, shift, r);
        __syncthreads();
        ch.init(table, site, g);
    }
    const T bias = static_cast<T>(level_bias);

    // fp32 arithmetic on packets: two elements at a time (forward_pair: packed multiplies and adds), the lane's constants
    // per component pair in registers for the whole walk
    constexpr bool PAIRS = std::is_same<T, float>::value && V >= 2;
    QPair qp[PAIRS ? V / 2 : 1];
    if constexpr (PAIRS) {
#pragma unroll
        for (int pr = 0; pr < V / 2; ++pr) qp[pr] = ch.pair(pr);
    }
    auto emit_row = [&](int64_t oo, const E (&in)[V], bool valid) {
        const int64_t e = oo * g.L + site.p0;
        E out[V];
        LevelPack<V> lv;
        if constexpr (PAIRS) {
#pragma unroll
            for (int pr = 0; pr < V / 2; ++pr) {
                const f2 xv = f2{static_cast<T>(in[2 * pr]), static_cast<T>(in[2 * pr + 1])};
                f2 c;
                const f2 yv = forward_pair(xv, qp[pr], r, c);
                out[2 * pr] = out_elem<IO, INIT>(INIT ? xv.x : yv.x);
                out[2 * pr + 1] = out_elem<IO, INIT>(INIT ? xv.y : yv.y);
                if (LEVELS) {
                    lv.b[2 * pr] = aux_byte<T>(c.x, r, bias, aux_kind);
                    lv.b[2 * pr + 1] = aux_byte<T>(c.y, r, bias, aux_kind);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const QParams<T> q = ch.params(j);
                const T xv = static_cast<T>(in[j]);
                const T c = clamped<T>(xv, q, r);
                out[j] = out_elem<IO, INIT>(INIT ? xv : dequant<T>(rne(c), q));
                if (LEVELS) lv.b[j] = aux_byte<T>(c, r, bias, aux_kind);
            }
        }
        if (valid) {
            if (!LEVELS || y != nullptr) store_elems<IO, V, NTS>(y, e, out);     // y == NULL: the one-byte output only
            if (LEVELS) lv.store(levels + e);
        }
    };

    // rows = full groups of UNROLL (every load issued before the first use) + one group of UNROLL/2 + ... + one
    // single row: no padded slots (a lane walks only a handful of rows at the BASELINE shapes)
    auto group = [&](int64_t i0, auto width) {
        constexpr int H = decltype(width)::value;
        E in[H][V];
#pragma unroll
        for (int u = 0; u < H; ++u) load_elems<IO, V, NTL>(x, walk.row(i0 + u) * g.L + site.p0, in[u]);
#pragma unroll
        for (int u = 0; u < H; ++u) emit_row(walk.row(i0 + u), in[u], true);
    };
    int64_t i = 0;
    if constexpr (DMA > 0) {
        // one copy per row: younger than row i's are the copies of rows i+1 .. i+DMA-1 (the y stores in between are not
        // counted: the wait is never too short)
        const int lane = threadIdx.x & 63;
        using V4 = __attribute__((ext_vector_type(4))) unsigned int;
        auto consume = [&](int64_t it, bool refill) {
            const V4 raw = *reinterpret_cast<const V4*>(ring + static_cast<uint32_t>(it % DMA) * kStage + lane * 16);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (refill) dma_issue(it + DMA);
            E in[V];
            __builtin_memcpy(&in[0], &raw, 16);
            emit_row(walk.row(it) < g.outer ? walk.row(it) : g.outer - 1, in, it < walk.n_rows);
        };
        for (; i + DMA < dma_n; ++i) {
            wait_vm<DMA - 1>();
            consume(i, true);
        }
        for (; i < dma_n; ++i) {
            wait_vm_upto(static_cast<int>(dma_n - 1 - i));
            consume(i, false);
        }
        return;
    }
    if (first_full) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) emit_row(walk.row(u), first[u], true);
        i = UNROLL;
    }
    for (; i + UNROLL <= walk.n_rows; i += UNROLL) group(i, std::integral_constant<int, UNROLL>{});
    if constexpr (UNROLL >= 8) if (i + 4 <= walk.n_rows) { group(i, std::integral_constant<int, 4>{}); i += 4; }
    if constexpr (UNROLL >= 4) if (i + 2 <= walk.n_rows) { group(i, std::integral_constant<int, 2>{}); i += 2; }
    if constexpr (UNROLL >= 2) if (i < walk.n_rows) group(i, std::integral_constant<int, 1>{});
}

// ------------------------------------------------------------------------------------------------
// K4 (window mode): backward
// ------------------------------------------------------------------------------------------------
// Segmented wave64 reduction: lanes hold (key, s, b).  A RUN is a maximal group of ADJACENT lanes
// with the same key (equal keys may re-appear further away -- folded rows, inner < V -- so runs are
// numbered with a ballot + popcount and the scan is keyed by run id, not by channel).  After
// log2(64) shuffle steps the first lane of every run owns the run total and adds it to the
// window's LDS slot with an LDS fp64 atomic (ds_add_f64).
// The two halves are separate so that the shuffles of all waves run side by side while the ADDS can be made in wave order
// (bwd_pc_kernel's epilogue): segmented_wave_reduce leaves the run total in (s, b) of the run's first lane and says whether
// this lane is one that adds; segmented_wave_commit adds.
template <bool SYM>
__device__ __forceinline__ bool segmented_wave_reduce(int key, double& s, double& b) {
    const int lane = threadIdx.x & 63;
    const int prev = __shfl_up(key, 1, 64);
    const bool head = (lane == 0) || (prev != key);
    const unsigned long long heads = __ballot(head);
    const int run = __popcll(heads & (~0ull >> (63 - lane)));  // heads at or below this lane: unique per run
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int orun = __shfl_down(run, d, 64);
        const double os = shfl_down_f64(s, d);
        const double ob = SYM ? 0.0 : shfl_down_f64(b, d);
        if (lane + d < 64 && orun == run) {
            s += os;
            if (!SYM) b += ob;
        }
    }
    return head && key >= 0;
}
template <bool SYM>
__device__ __forceinline__ void segmented_wave_commit(bool adds, int key, double s, double b, double* lds_s, double* lds_b) {
    if (adds) {
        __hip_atomic_fetch_add(&lds_s[key], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (!SYM) __hip_atomic_fetch_add(&lds_b[key], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// WW: row-group windows (make_geom_ww: inner == 1, CPL == V) -- the lane's channels are its own, their constants are
// computed from global memory into registers (no LDS table) and the epilogue sums the row groups in a fixed order.
// DMA > 0: the rows reach the wave through an LDS ring of DMA stages filled by LDS-DMA (glds16): DMA rows of HBM requests
// stay in flight per wave without holding registers.  The arithmetic-heavy 16-bit kernels have no registers to spare for
// more than one row of ordinary loads, and one row in flight per wave does not cover the HBM latency (the dx-only
// kernel, which has the registers, streams the same tensor 20 % faster when all of a workgroup's loads are issued up
// front: profiles/r02_pc_variants_eval.txt).
template <typename IO, int V, int CPL, bool SYM, bool INIT, bool EVAL, int UNROLL, bool NTL, bool NTS, bool PIPE, bool WW = false,
          int DMA = 0, int BLOCK = kBlock>
__global__ __launch_bounds__(BLOCK) void bwd_pc_kernel(const void* __restrict__ grad, const void* __restrict__ x,
                                                        void* __restrict__ dx, PcGeom g,
                                                        const typename IO::arith* __restrict__ scale,
                                                        const typename IO::arith* __restrict__ shift,
                                                        Range<typename IO::arith> r, typename IO::arith grad_scaler,
                                                        double2* __restrict__ partials, PcDirect<typename IO::arith> direct) {
    using T = typename IO::arith;
    using E = typename IO::elem;
    using LC = LaneChannels<T, V, CPL>;
    // OWN: owner windows (make_geom_own) -- a fat workgroup of R row slots over the run of k whole channels, all rows: the
    // LDS slots end up holding FINAL sums, stored straight to d_scale / d_shift (`direct`); no partials, no finalize launch
    constexpr bool OWN = !WW && BLOCK > kBlock;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef LSQ_TIMELINE     // experiment build: shader-clock stamps per wave (tools/exp_timeline.py)
    const unsigned long long tl0 = __builtin_readcyclecounter();
    unsigned long long tl1 = 0, tl2 = 0, tl_wait = 0;
#define LSQ_TL_WAIT(stmt) do { const unsigned long long a_ = __builtin_readcyclecounter(); stmt; tl_wait += __builtin_readcyclecounter() - a_; } while (0)
#else
#define LSQ_TL_WAIT(stmt) stmt
#endif
    static_assert(!WW || (CPL == V && V > 1), "row-group windows: one channel per packet component");
    static_assert(BLOCK == kBlock || DMA > 0, "768/1024-lane workgroups: row-group and owner windows, on the ring only");
    static_assert(!OWN || (!EVAL && V > 1 && CPL <= 2), "owner windows: whole packets of one or two channels, training modes");
    QSlot<T>* table = reinterpret_cast<QSlot<T>*>(smem);
    // fp64 slots of the window's channels: [k_slots] d_scale sums, [k_slots] d_shift sums -- one such set per workgroup, added
    // to with LDS atomics; owner windows (their sums are FINAL) keep one set per WAVE and add the sets in wave order at the
    // end, so that d_scale / d_shift / wide come out the same bits launch after launch
    const uint32_t sum_sets = OWN ? bwd_lds_sum_sets(g) : 1u;
    double* lds_s0 = reinterpret_cast<double*>(smem + static_cast<size_t>(g.k_slots) * sizeof(QSlot<T>));
    double* lds_s = lds_s0 + (OWN ? static_cast<size_t>(threadIdx.x >> 6) * 2u * g.k_slots : 0u);
    double* lds_b = lds_s + g.k_slots;

    // The window's raw scale / shift are requested FIRST: vector-memory operations retire in issue order, so a wait for loads
    // issued behind the first rows' loads would be a wait for those rows (measured, tools/exp_timeline.py: the prologue of a
    // workgroup took 3-7 us, a quarter of its life, most of it that wait).
    //  * ring kernels (fp32 parameters): LDS-DMA dword copies into a staging area behind the fp64 slots -- asm like the
    //    row copies, invisible to the compiler's own s_waitcnt bookkeeping, ordered below with a counted wait;
    //  * register-loop kernels: ordinary loads into registers (the compiler counts its own loads in issue order).
    constexpr bool STAGE = DMA > 0 && !WW && std::is_same<T, float>::value;
    const bool raw_first = !WW && DMA == 0 && g.k_slots <= kRawSlots * kBlock;
    float* raw_stage = reinterpret_cast<float*>(smem + static_cast<size_t>(g.k_slots) * (sizeof(QSlot<T>) + 16 * sum_sets));   // [k_slots] scale, [k_slots] shift
    ChannelRaw<T> raw;
    if constexpr (STAGE) {
        const int64_t c_first = window_first_channel(g);
        const uint32_t stage_lds = __builtin_amdgcn_readfirstlane(lds_offset_of(raw_stage));
        for (int k0 = 0; k0 < g.k_slots; k0 += kBlock) {          // uniform trip count
            const int k = k0 + threadIdx.x;
            if (k < g.k_slots) {                                  // (the other lanes stay out: their dwords would land in a neighbour's slots)
                int64_t c = c_first + k;
                c = c < g.C ? c : g.C - 1;                        // slots past the last channel copy a valid address
                const uint32_t dst = __builtin_amdgcn_readfirstlane(stage_lds + static_cast<uint32_t>(k0 + (threadIdx.x & ~63)) * 4u);
                glds4(scale + c, dst);
                glds4(shift + c, dst + static_cast<uint32_t>(g.k_slots) * 4u);
            }
        }
    } else if (raw_first) {
        raw = load_channel_raw<T>(g.k_slots, window_first_channel(g), g.C, scale, shift);
    }
    int32_t lane_in_group = 0;
    const LaneSite site = WW ? lane_site_ww(g, V, lane_in_group) : (OWN ? lane_site_own(g, V) : lane_site(g, V));
    const RowWalk walk(g, site);
    // A group = UNROLL rows.  load_group never predicates: rows past the lane's last one re-read the last row.
    auto load_group = [&](E (&gb)[UNROLL][V], E (&xb)[UNROLL][V], int64_t i0) {
        const int64_t last = walk.n_rows - 1;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t e = walk.row(i0 + u < last ? i0 + u : last) * g.L + site.p0;
            load_elems<IO, V, NTL>(grad, e, gb[u]);
            load_elems<IO, V, NTL>(x, e, xb[u]);
        }
    };
    E first_g[UNROLL][V], first_x[UNROLL][V];   // first group in flight before the table build (see K3)
    const bool first_full = DMA > 0 ? false : (PIPE ? walk.n_rows > 0 : walk.n_rows >= UNROLL);
    if (first_full) load_group(first_g, first_x, 0);
    // ---- LDS-DMA ring (DMA > 0): this wave's DMA stages; stage = [64 grad packets][64 x packets] ----
    static_assert(DMA == 0 || V * sizeof(E) == 16, "the LDS-DMA ring moves 16-byte packets");
    const int64_t dma_n = walk.n_tiles_split;                          // the same for every lane of the workgroup
    unsigned char* ring = smem + bwd_lds_front_bytes(g, sizeof(QSlot<T>)) + (threadIdx.x >> 6) * (DMA * kDmaStageBytes);
    const uint32_t ring_lds = DMA > 0 ? __builtin_amdgcn_readfirstlane(lds_offset_of(ring)) : 0u;
    // row i of this lane, clamped into the tensor (rows past the lane's last one and dead lanes re-read valid memory)
    auto dma_issue = [&](int64_t i) {
        int64_t row = walk.row(i);
        row = row < g.outer ? row : g.outer - 1;
        const int64_t e = row * g.L + (site.live ? site.p0 : 0);
        const uint32_t dst = ring_lds + static_cast<uint32_t>(i % (DMA > 0 ? DMA : 1)) * kDmaStageBytes;
        glds16_rt(static_cast<const E*>(grad) + e, dst, g.ring_nt);
        glds16_rt(static_cast<const E*>(x) + e, dst + 64 * 16, g.ring_nt);
    };
    // Row-group windows on the ring, fp32 parameters: the lane's own V scale / shift values are requested BEFORE its rows, as
    // LDS-DMA copies into the last one or two ring stages (16 bytes per lane and copy; the row copies that belong into those
    // stages are issued once the parameters have been read out): see STAGE above for why the order matters.
    constexpr int kParCopies = (V * 4) / 16;                                       // 16-byte copies per parameter: 2 (V = 8), 1 (V = 4)
    constexpr int kParStages = DMA > 0 ? (2 * kParCopies * 1024 + kDmaStageBytes - 1) / kDmaStageBytes : 0;
    constexpr bool WSTAGE_ABLE = WW && DMA > kParStages && std::is_same<T, float>::value && LC::N == V && (V == 8 || V == 4);
    const bool wstage = WSTAGE_ABLE && ((reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift)) & 15u) == 0;
    if constexpr (DMA > 0) {
        if (wstage) {
            const int64_t c0 = site.live ? site.p0 : 0;
            const uint32_t par_lds = ring_lds + static_cast<uint32_t>(DMA - kParStages) * kDmaStageBytes;
#pragma unroll
            for (int q4 = 0; q4 < kParCopies; ++q4) {
                glds16<false>(scale + c0 + 4 * q4, par_lds + static_cast<uint32_t>(q4) * 1024u);
                glds16<false>(shift + c0 + 4 * q4, par_lds + static_cast<uint32_t>(kParCopies + q4) * 1024u);
            }
            for (int64_t i = 0; i < DMA - kParStages && i < dma_n; ++i) dma_issue(i);
        } else {
            for (int64_t i = 0; i < DMA && i < dma_n; ++i) dma_issue(i);    // in flight before the constants are built
        }
    }
    LC ch;
    if constexpr (WW) {
        // channel p0 + j is component j's own: constants straight into registers (lsq_kernel.h:157-158 + :12)
        ch.split = (CPL == 2) ? 1 : V;     // V == 2 (8-byte elements): LaneChannels' two-channel form, component 1 = channel 1
        if (wstage) {
            if constexpr (WSTAGE_ABLE) {
                // younger than the parameter copies: the copies of the rows issued so far (two each)
                const int64_t rows_out = dma_n < DMA - kParStages ? dma_n : DMA - kParStages;
                wait_vm_upto(static_cast<int>(2 * rows_out));
                const unsigned char* par = ring + (DMA - kParStages) * kDmaStageBytes + (threadIdx.x & 63) * 16;
                float sv[V], bv[V];
#pragma unroll
                for (int q4 = 0; q4 < kParCopies; ++q4) {
                    __builtin_memcpy(&sv[4 * q4], par + q4 * 1024, 16);
                    __builtin_memcpy(&bv[4 * q4], par + (kParCopies + q4) * 1024, 16);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the stages are in registers: the rows may land
                for (int64_t i = DMA - kParStages; i < DMA && i < dma_n; ++i) dma_issue(i);
#pragma unroll
                for (int j = 0; j < LC::N; ++j) {
                    ch.q[j] = make_qparams<T>(sanitize_scale_per_channel<T>(sv[j]), bv[j], r);
                    ch.key[j] = j * g.ww_lanes + lane_in_group;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < LC::N; ++j) {
                const int64_t c = site.live ? site.p0 + j : 0;
                ch.q[j] = make_qparams<T>(sanitize_scale_per_channel<T>(scale[c]), shift[c], r);
                ch.key[j] = j * g.ww_lanes + lane_in_group;
            }
        }
    } else {
        if constexpr (STAGE) {
            // younger than this wave's staging copies: the row copies just issued (two per row)
            wait_vm_upto(static_cast<int>(2 * (dma_n < DMA ? dma_n : DMA)));
            for (int k = threadIdx.x; k < g.k_slots; k += kBlock) {     // slot k was staged by this very wave
                const int64_t c = site.c_lo + k;
                QSlot<T> e;
                if (c < g.C) {
                    const QParams<T> q = make_qparams<T>(sanitize_scale_per_channel<T>(raw_stage[k]), raw_stage[g.k_slots + k], r);
                    e.s = q.s; e.inv_s = q.inv_s; e.zp = q.zp; e.pad = static_cast<T>(0);
                } else {
                    e.s = static_cast<T>(1); e.inv_s = static_cast<T>(1); e.zp = static_cast<T>(0); e.pad = static_cast<T>(0);
                }
                table[k] = e;
            }
        } else if (raw_first) {
            finish_channel_table<T>(table, g.k_slots, site.c_lo, g.C, raw, r);
        } else {
            build_channel_table<T>(table, g.k_slots, site.c_lo, g.C, scale, shift, r);
        }
        if (!EVAL) {
            for (int k = threadIdx.x; k < static_cast<int>(2u * sum_sets) * g.k_slots; k += static_cast<int>(blockDim.x)) lds_s0[k] = 0.0;
        }
        __syncthreads();
        ch.init(table, site, g);
    }

    // CPL == 1: one accumulator pair.  CPL == 2 / V: one pair per COMPONENT of the packet (a cvt + an add per
    // term in the loop, no selects); CPL == 2 folds them into its two channels after the walk, by `split`.
    // PAIRS (fp32 arithmetic on packets): the row is computed two elements at a time (backward_pair: packed multiplies and
    // adds) and the accumulators are 2-vectors too; CPL == 1 then keeps TWO accumulators (even / odd components), added
    // up after the walk.
    constexpr bool PAIRS = std::is_same<T, float>::value && V >= 2;
    constexpr int kAcc = (LC::N == 1) ? (PAIRS ? 2 : 1) : V;
    double acc_s[kAcc], acc_b[kAcc];
#pragma unroll
    for (int j = 0; j < kAcc; ++j) { acc_s[j] = 0.0; acc_b[j] = 0.0; }
    // PRE32 (16-bit storage on the LDS-DMA ring): the kernel is VALU-bound (profiles/r02_sq_counters_before_cfg5_bf16.txt)
    // and every wave64 VALU instruction costs ~4 cycles whatever its width (profiles/r02_valu_issue_rates.txt), so the
    // reduction is made cheaper per term, within the parity bar of 1e-6 x sum|terms| (profiles/r02_pre32_ab.txt: -6 %):
    //  * the terms of up to kPreRows consecutive rows are first added per component in fp32 (one packed add for two
    //    components), then the pre-sum joins the fp64 accumulator -- a convert and an fp64 add per kPreRows
    //    terms instead of per term; at most kPreRows - 1 fp32 roundings per pre-sum, <= 1.8e-7 of the sum of the |terms|
    //    in the worst case, ~1e-9 typically (range: a pre-sum overflows where four terms of one sign exceed FLT_MAX together --
    //    gradient x level products around 1e38, where the fp32 terms themselves are about to);
    //  * the gradient scaler multiplies the fp64 sums once (backward_elem<.., RAW>) instead of every term.
    constexpr bool PRE32 = DMA > 0 && sizeof(E) < 4 && !EVAL;
    static_assert(!PRE32 || PAIRS, "16-bit storage on the ring moves packets of 8");
    constexpr int kPreRows = 4;
    f2 pre_s[PRE32 ? kAcc / 2 : 1], pre_b[PRE32 ? kAcc / 2 : 1];
#pragma unroll
    for (int j = 0; j < (PRE32 ? kAcc / 2 : 1); ++j) { pre_s[j] = f2{0.0f, 0.0f}; pre_b[j] = f2{0.0f, 0.0f}; }
    // the lane's constants per component pair, in registers for the whole walk
    QPair qp[PAIRS ? V / 2 : 1];
    if constexpr (PAIRS) {
#pragma unroll
        for (int pr = 0; pr < V / 2; ++pr) qp[pr] = ch.pair(pr);
    }

    // CLEAR = false: the caller's next row ASSIGNS the pre-sums (emit_row_at's `first`), so they need no zeroing
    auto flush_pre = [&](auto clear) {
        if constexpr (PRE32) {
#pragma unroll
            for (int a = 0; a < kAcc / 2; ++a) {
                acc_s[2 * a] += static_cast<double>(pre_s[a].x);
                acc_s[2 * a + 1] += static_cast<double>(pre_s[a].y);
                if (decltype(clear)::value) pre_s[a] = f2{0.0f, 0.0f};
                if (!SYM) {
                    acc_b[2 * a] += static_cast<double>(pre_b[a].x);
                    acc_b[2 * a + 1] += static_cast<double>(pre_b[a].y);
                    if (decltype(clear)::value) pre_b[a] = f2{0.0f, 0.0f};
                }
            }
        }
    };

    // one row of this lane: V elements at element offset e.  `first` (compile time): the row opens a pre-sum group, its
    // terms are assigned instead of added (no zeroing, no add).
    auto emit_row_at = [&](int64_t e, const E (&gi)[V], const E (&xi)[V], bool valid, auto first) {
        E out[V];
        if constexpr (PAIRS) {
#pragma unroll
            for (int pr = 0; pr < V / 2; ++pr) {
                const QPair& q = qp[pr];
                const f2 gv = f2{static_cast<T>(gi[2 * pr]), static_cast<T>(gi[2 * pr + 1])};
                const f2 xv = f2{static_cast<T>(xi[2 * pr]), static_cast<T>(xi[2 * pr + 1])};
                f2 dxv;
                if constexpr (EVAL) {
                    dxv = backward_pair_eval<INIT>(gv, xv, q, r);
                } else {
                    f2 ds_t, db_t;
                    dxv = backward_pair<SYM, INIT>(gv, xv, q, r, ds_t, db_t);
                    if (!valid) { ds_t = f2{0.0f, 0.0f}; db_t = f2{0.0f, 0.0f}; }
                    const int a = (LC::N == 1) ? 0 : pr;           // accumulator pair of this component pair (unrolled: a constant)
                    if constexpr (PRE32) {
                        if (decltype(first)::value && (LC::N != 1 || pr == 0)) {
                            pre_s[a] = ds_t;
                            if (!SYM) pre_b[a] = db_t;
                        } else {
                            pre_s[a] += ds_t;
                            if (!SYM) pre_b[a] += db_t;
                        }
                    } else {
                        ds_t *= grad_scaler;                        // :122, every term individually (reference bits)
                        acc_s[2 * a] += static_cast<double>(ds_t.x);
                        acc_s[2 * a + 1] += static_cast<double>(ds_t.y);
                        if (!SYM) {
                            db_t *= grad_scaler;
                            acc_b[2 * a] += static_cast<double>(db_t.x);
                            acc_b[2 * a + 1] += static_cast<double>(db_t.y);
                        }
                    }
                }
                out[2 * pr] = out_elem<IO, INIT>(dxv.x);                   // (init_mode: dX IS the gradient, :112)
                out[2 * pr + 1] = out_elem<IO, INIT>(dxv.y);
            }
        } else {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const QParams<T> q = ch.params(j);
                const T gv = static_cast<T>(gi[j]), xv = static_cast<T>(xi[j]);
                if (EVAL) {
                    out[j] = out_elem<IO, INIT>(backward_elem_eval<T, INIT>(gv, xv, q, r));
                } else {
                    T ds_t, db_t;
                    out[j] = out_elem<IO, INIT>(backward_elem<T, SYM, INIT>(gv, xv, q, r, grad_scaler, ds_t, db_t));
                    if (!valid) { ds_t = static_cast<T>(0); db_t = static_cast<T>(0); }
                    const double a = static_cast<double>(ds_t), c = static_cast<double>(db_t);
                    acc_s[j < kAcc ? j : 0] += a;
                    if (!SYM) acc_b[j < kAcc ? j : 0] += c;
                }
            }
        }
        // (owner windows: the stand-in lanes past the last row slot computed a row another lane stores)
        if (valid && (!OWN || site.counts)) store_elems<IO, V, NTS>(dx, e, out);
    };
    auto emit_row = [&](int64_t oo, const E (&gi)[V], const E (&xi)[V], bool valid) {
        emit_row_at(oo * g.L + site.p0, gi, xi, valid, std::false_type{});
    };

    auto emit_full = [&](int64_t i0, const E (&gb)[UNROLL][V], const E (&xb)[UNROLL][V]) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) emit_row(walk.row(i0 + u), gb[u], xb[u], true);
    };
    auto emit_ragged = [&](int64_t i0, const E (&gb)[UNROLL][V], const E (&xb)[UNROLL][V]) {
        const int64_t last = walk.n_rows - 1;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) emit_row(walk.row(i0 + u < last ? i0 + u : last), gb[u], xb[u], i0 + u <= last);
    };
    int64_t i = 0;
#ifdef LSQ_TIMELINE
    tl1 = __builtin_readcyclecounter();
#endif
    if constexpr (DMA > 0) {
        // Row i was requested DMA rows ago.  Younger than its two copies are the copies of rows i+1 .. i+DMA-1 (two each)
        // and the dx stores in between; only the copies are counted (a wave without a valid lane skips its stores), so
        // the wait is never too short and at least 2/3 of the ring stays in flight.
        const int lane = threadIdx.x & 63;
        using V4 = __attribute__((ext_vector_type(4))) unsigned int;
        auto consume = [&](int64_t it, bool refill, auto all_valid) {
            const unsigned char* stage = ring + static_cast<uint32_t>(it % DMA) * kDmaStageBytes + lane * 16;
            const V4 graw = *reinterpret_cast<const V4*>(stage);
            const V4 xraw = *reinterpret_cast<const V4*>(stage + 64 * 16);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the stage is in registers: it may be refilled
            if (refill) dma_issue(it + DMA);
            E gi[V], xi[V];
            __builtin_memcpy(&gi[0], &graw, 16);
            __builtin_memcpy(&xi[0], &xraw, 16);
            if constexpr (decltype(all_valid)::value) {
                emit_row(walk.row(it), gi, xi, true);
            } else {
                emit_row(walk.row(it) < g.outer ? walk.row(it) : g.outer - 1, gi, xi, it < walk.n_rows);
            }
        };
        // ragged (owner windows only): the last row tile is short -- the lanes of the row slots past its end have one row
        // fewer.  The blocks then stop one tile early (their refills never reach the short tile) and the rows they leave are
        // walked one at a time with the validity and the row clamp of the generic form.
        auto loop = [&](auto all_valid, auto nt, auto ragged) {
            constexpr bool kTailValid = decltype(all_valid)::value && !decltype(ragged)::value;
            [[maybe_unused]] const std::integral_constant<bool, kTailValid> tail_valid{};
            const int64_t dma_blocks = dma_n - (decltype(ragged)::value ? 1 : 0);
            if constexpr (decltype(all_valid)::value) {
                // Steady state in blocks of DMA rows: the ring stage of a row is a compile-time constant (its LDS addresses
                // are instruction offsets), the lane's row addresses advance by one add (every lane walks every row: no
                // clamping), and -- PRE32 -- the first row of a block assigns the pre-sums, which are flushed un-cleared at
                // its end.
                // The wait is EXACT here.  Vector-memory operations retire in issue order and every row of this loop issues
                // the same ones -- two copies (the refill of its stage), then its dx store -- so the operations younger than
                // row i's two copies are: the copies of rows i+1 .. i+DMA-1 and the dx stores of rows i-DMA .. i-1 (row i's
                // copies were issued as the refill of row i-DMA, before that row's store); in the first block the stores of
                // rows 0 .. u-1 only.  (The generic loops below cannot know whether a wave stored, count the copies alone and
                // so wait for one more row and two store acknowledgements than they need.)
                static_assert(!PRE32 || DMA <= kPreRows, "a pre-sum group is at most kPreRows rows");
                const int64_t step_e = walk.step * g.L;
                int64_t e_cur = walk.row(0) * g.L + site.p0;          // i == 0 here
                // Owner windows: the waves of a SIMD take turns at the higher issue priority, block by block.  An owner
                // workgroup has its CU to itself and ends at a barrier, so it is as slow as its slowest wave -- and the
                // arbiter serves the OLDER wave of a SIMD first: waves 0-3 walked their rows in 20.8 us, waves 4-6 (the second
                // wave of their SIMD) in 27.9 us, and the first four then sat 7 us at the barrier
                // (profiles/r04_owner_timeline.txt).  Alternating s_setprio makes the two finish together.
                [[maybe_unused]] const int own_phase = OWN ? __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 8)) & 1 : 0;
                [[maybe_unused]] int own_blk = 0;
                auto block = [&](auto first_block) {
                    if constexpr (OWN) {
                        if (g.own_prio) {
                            if ((own_blk ^ own_phase) & 1) __builtin_amdgcn_s_setprio(2);
                            else __builtin_amdgcn_s_setprio(0);
                            ++own_blk;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < DMA; ++u) {
                        if (decltype(first_block)::value) LSQ_TL_WAIT(wait_vm_upto(2 * (DMA - 1) + u));
                        else LSQ_TL_WAIT((wait_vm<2 * (DMA - 1) + DMA>()));
                        const unsigned char* stage = ring + u * kDmaStageBytes + lane * 16;
                        const V4 graw = *reinterpret_cast<const V4*>(stage);
                        const V4 xraw = *reinterpret_cast<const V4*>(stage + 64 * 16);
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the stage is in registers: refill it
                        const int64_t e_next = e_cur + DMA * step_e;
                        glds16<decltype(nt)::value>(static_cast<const E*>(grad) + e_next, ring_lds + u * kDmaStageBytes);
                        glds16<decltype(nt)::value>(static_cast<const E*>(x) + e_next, ring_lds + u * kDmaStageBytes + 64 * 16);
                        E gi[V], xi[V];
                        __builtin_memcpy(&gi[0], &graw, 16);
                        __builtin_memcpy(&xi[0], &xraw, 16);
                        if (u == 0) emit_row_at(e_cur, gi, xi, true, std::true_type{});
                        else emit_row_at(e_cur, gi, xi, true, std::false_type{});
                        e_cur += step_e;
                    }
                    flush_pre(std::false_type{});
                    i += DMA;
                };
                static_assert(2 * (DMA - 1) + DMA <= 15, "wait_vm_upto covers counts up to 15");
                if (i + 2 * DMA <= dma_blocks) block(std::true_type{});
                while (i + 2 * DMA <= dma_blocks) block(std::false_type{});
                if constexpr (OWN) __builtin_amdgcn_s_setprio(0);
                if constexpr (PRE32) {
#pragma unroll
                    for (int j = 0; j < kAcc / 2; ++j) { pre_s[j] = f2{0.0f, 0.0f}; pre_b[j] = f2{0.0f, 0.0f}; }
                }
            }
            // the rows the blocks left over (and every row of a wave with dead lanes or a ragged last tile): one at a time,
            // ring stage and validity at run time.  `stores`: dx stores younger than row i's copies -- known only when every
            // row of the wave stores (see above); otherwise they are left out of the count (the wait is then longer).
            auto stores = [&](int64_t row) { return kTailValid ? static_cast<int>(row < DMA ? row : DMA) : 0; };
            for (; i + DMA < dma_n; ++i) {           // the ring is full, one refill per row
                LSQ_TL_WAIT(wait_vm_upto(2 * (DMA - 1) + stores(i)));
                consume(i, true, tail_valid);
                if (PRE32 && (i & (kPreRows - 1)) == kPreRows - 1) flush_pre(std::true_type{});
            }
            for (; i < dma_n; ++i) {                 // the last DMA rows: nothing left to request
                LSQ_TL_WAIT(wait_vm_upto(static_cast<int>(2 * (dma_n - 1 - i)) + stores(i)));
                consume(i, false, tail_valid);
                if (PRE32 && (i & (kPreRows - 1)) == kPreRows - 1) flush_pre(std::true_type{});
            }
            flush_pre(std::true_type{});
            if constexpr (PRE32) {           // the gradient scaler, once per sum
#pragma unroll
                for (int j = 0; j < kAcc; ++j) {
                    acc_s[j] *= static_cast<double>(grad_scaler);
                    acc_b[j] *= static_cast<double>(grad_scaler);
                }
            }
        };
        // every lane of this wave walks all dma_n rows (no dead lane, no ragged last tile): no per-row validity selects
        if (__builtin_amdgcn_readfirstlane(__all(site.live && walk.n_rows == dma_n) ? 1 : 0)) {
            if (g.ring_nt) loop(std::true_type{}, std::true_type{}, std::false_type{});
            else loop(std::true_type{}, std::false_type{}, std::false_type{});
        } else if (OWN && __builtin_amdgcn_readfirstlane(__all(site.live && walk.n_rows + 1 >= dma_n) ? 1 : 0)) {
            if constexpr (OWN) {       // some of this wave's row slots miss the last tile only
                if (g.ring_nt) loop(std::true_type{}, std::true_type{}, std::true_type{});
                else loop(std::true_type{}, std::false_type{}, std::true_type{});
            }
        } else {
            loop(std::false_type{}, std::false_type{}, std::false_type{});
        }
    } else if (PIPE) {
        // Software pipeline, two register buffers: the loads of group k+1 are issued BEFORE the arithmetic of
        // group k, so every wave keeps HBM requests in flight while it computes (for 16-bit storage the VALU time
        // of a group is about its HBM time: without this the two only overlap across waves).
        if (first_full) {
            E other_g[UNROLL][V], other_x[UNROLL][V];
            // sched_barrier: the machine scheduler otherwise sinks each load group below the arithmetic that
            // precedes its first use (to save registers), which undoes the pipeline
            while (i + 2 * UNROLL <= walk.n_rows) {          // groups i and i + UNROLL are both full
                load_group(other_g, other_x, i + UNROLL);
                __builtin_amdgcn_sched_barrier(0);
                emit_full(i, first_g, first_x);
                __builtin_amdgcn_sched_barrier(0);
                load_group(first_g, first_x, i + 2 * UNROLL);   // may be ragged or past the end: clamped re-reads
                __builtin_amdgcn_sched_barrier(0);
                emit_full(i + UNROLL, other_g, other_x);
                __builtin_amdgcn_sched_barrier(0);
                i += 2 * UNROLL;
            }
            if (i + UNROLL <= walk.n_rows) {                  // `first` holds a full group
                load_group(other_g, other_x, i + UNROLL);
                __builtin_amdgcn_sched_barrier(0);
                emit_full(i, first_g, first_x);
                i += UNROLL;
                if (i < walk.n_rows) emit_ragged(i, other_g, other_x);
            } else if (i < walk.n_rows) {
                emit_ragged(i, first_g, first_x);
            }
        }
    } else {
        // plain loop: full groups of UNROLL, then one group of UNROLL/2, ..., one single row -- no padded slots
        auto group = [&](int64_t i0, auto width) {
            constexpr int H = decltype(width)::value;
            E gi[H][V], xi[H][V];
#pragma unroll
            for (int u = 0; u < H; ++u) {
                const int64_t e = walk.row(i0 + u) * g.L + site.p0;
                load_elems<IO, V, NTL>(grad, e, gi[u]);
                load_elems<IO, V, NTL>(x, e, xi[u]);
            }
#pragma unroll
            for (int u = 0; u < H; ++u) emit_row(walk.row(i0 + u), gi[u], xi[u], true);
        };
        if (first_full) {
            emit_full(0, first_g, first_x);
            i = UNROLL;
        }
        for (; i + UNROLL <= walk.n_rows; i += UNROLL) group(i, std::integral_constant<int, UNROLL>{});
        if constexpr (UNROLL >= 8) if (i + 4 <= walk.n_rows) { group(i, std::integral_constant<int, 4>{}); i += 4; }
        if constexpr (UNROLL >= 4) if (i + 2 <= walk.n_rows) { group(i, std::integral_constant<int, 2>{}); i += 2; }
        if constexpr (UNROLL >= 2) if (i < walk.n_rows) group(i, std::integral_constant<int, 1>{});
    }
#ifdef LSQ_TIMELINE
    tl2 = __builtin_readcyclecounter();
    auto tl_record = [&]() {
        if (g.timeline && (threadIdx.x & 63) == 0) {
            const int64_t wave = ((static_cast<int64_t>(blockIdx.y) * g.n_windows + blockIdx.x) * (BLOCK / 64)) + (threadIdx.x >> 6);
            unsigned long long* rec = g.timeline + wave * 8;
            unsigned int hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            unsigned int xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            rec[0] = tl0; rec[1] = tl1; rec[2] = tl2; rec[3] = __builtin_readcyclecounter(); rec[4] = tl_wait;
            rec[5] = static_cast<unsigned long long>(walk.n_tiles_split); rec[6] = hw; rec[7] = xcc;
        }
    };
    if (EVAL) { tl_record(); return; }
#endif
    if (EVAL) return;
    if constexpr (PAIRS && LC::N == 1) {     // one channel per lane: even + odd components
        acc_s[0] += acc_s[1];
        acc_b[0] += acc_b[1];
    }

    if constexpr (WW) {
        // R row groups, R interleaved row sets of the same w x V channels.  Every group parks its sums in LDS
        // ([group][component][lane]: a lane-contiguous 16 bytes each, conflict-free); then EVERY thread takes part in adding them
        // up: slot s = component * w + lane is summed over the groups in group order by one thread (k_slots >= workgroup size:
        // a thread takes several slots) or -- narrow windows, more threads than slots -- by P = threads / k_slots threads that
        // each take the groups part, part + P, ... and whose P results are then added in order.  (Round 4 had row group 0's
        // lanes add all R - 1 parked rows themselves: [64,197,768] bf16, 96 lanes x 56 dependent LDS reads while 672 lanes
        // idled, 2-3 us of a 14 us workgroup.)  The workgroup's partial row is stored slot-major (contiguous per component).
        double2* comb = reinterpret_cast<double2*>(smem);
        const int w = g.ww_lanes, rg = site.row_in_tile;
        const int k_slots = g.k_slots, nthr = static_cast<int>(blockDim.x), t = static_cast<int>(threadIdx.x);
        if constexpr (DMA > 0) __syncthreads();     // the combine buffer takes the ring's place: every wave is done reading
        if (rg < g.R) {
#pragma unroll
            for (int j = 0; j < V; ++j) comb[(rg * V + j) * w + lane_in_group] = make_double2(acc_s[j], acc_b[j]);
        }
        __syncthreads();
        const int64_t block_linear = static_cast<int64_t>(blockIdx.y) * g.n_windows + blockIdx.x;
        double2* out = partials + block_linear * k_slots;
        auto publish = [&](int slot, double ts, double tb) { out[slot] = make_double2(ts, tb); };
        if (k_slots >= nthr) {
            for (int slot = t; slot < k_slots; slot += nthr) {
                double ts = comb[slot].x, tb = comb[slot].y;
                for (int o = 1; o < g.R; ++o) {
                    const double2 v = comb[o * k_slots + slot];
                    ts += v.x;
                    tb += v.y;
                }
                publish(slot, ts, tb);
            }
        } else {
            const int P = nthr / k_slots, slot = t % k_slots, part = t / k_slots;
            double2* comb2 = comb + static_cast<size_t>(g.R) * k_slots;
            if (part < P && part < g.R) {
                double ts = comb[part * k_slots + slot].x, tb = comb[part * k_slots + slot].y;
                for (int o = part + P; o < g.R; o += P) {
                    const double2 v = comb[o * k_slots + slot];
                    ts += v.x;
                    tb += v.y;
                }
                comb2[part * k_slots + slot] = make_double2(ts, tb);
            }
            __syncthreads();
            if (t < k_slots) {
                const int parts = P < g.R ? P : g.R;
                double ts = comb2[t].x, tb = comb2[t].y;
                for (int q = 1; q < parts; ++q) {
                    ts += comb2[q * k_slots + t].x;
                    tb += comb2[q * k_slots + t].y;
                }
                publish(t, ts, tb);
            }
        }
#ifdef LSQ_TIMELINE
        tl_record();
#endif
        return;
    }

    // The window's slots take the waves' run totals with LDS fp64 atomics.  One slot set per workgroup (256-lane windows):
    // the adds are made in WAVE ORDER -- wave w adds between barrier w and barrier w + 1; inside a wave the order is the
    // program's and the LDS unit's lane order -- so the partial row a workgroup publishes, and with it d_scale / d_shift / the
    // un-rounded sums the sharded path all-reduces, are the same bits launch after launch (until round 6 the four waves added in
    // arrival order: two launches could differ by an fp64 rounding).  The shuffles above ran in all waves at once; what is
    // serialised is one or two LDS atomics per lane.  Owner windows keep a slot set per wave (added in wave order below).
    const int my_wave = static_cast<int>(threadIdx.x >> 6);
    auto in_wave_order = [&](auto&& adds_fn) {
        if constexpr (OWN) {
            adds_fn();
            __syncthreads();
        } else {
            for (int w = 0; w < BLOCK / 64; ++w) {
                if (my_wave == w) adds_fn();
                __syncthreads();
            }
        }
    };
    if (CPL == 2) {
        // components below `split` -> first channel, the rest -> second channel (two disjoint sums: a
        // non-finite term of one channel never reaches the other).  The second channel of lane i is
        // the FIRST channel of lane i+1 (their positions are contiguous and inner >= V), so its sums
        // travel one lane up and join that lane's run: ONE segmented reduction instead of two.  Only
        // the last lane of a wave / of a row has no neighbour and adds its second channel itself.
        const int lane = threadIdx.x & 63;
        const bool has_hi = site.counts && ch.split < V;
        double lo_s = 0.0, lo_b = 0.0, hi_s = 0.0, hi_b = 0.0;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const bool hi = j >= ch.split;
            lo_s += hi ? 0.0 : acc_s[j];
            hi_s += hi ? acc_s[j] : 0.0;
            lo_b += hi ? 0.0 : acc_b[j];
            hi_b += hi ? acc_b[j] : 0.0;
        }
        const int key_hi = has_hi ? ch.key[LC::N - 1] : -1;
        const int next_key0 = __shfl_down(site.counts ? ch.key[0] : -2, 1, 64);
        const bool handoff = has_hi && lane < 63 && next_key0 == key_hi;
        const double give_s = handoff ? hi_s : 0.0, give_b = handoff ? hi_b : 0.0;
        const double got_s = shfl_up_f64(give_s, 1), got_b = shfl_up_f64(give_b, 1);
        if (lane > 0) { lo_s += got_s; lo_b += got_b; }
        const int key_lo = site.counts ? ch.key[0] : -1;
        const bool adds_lo = segmented_wave_reduce<SYM>(key_lo, lo_s, lo_b);
        in_wave_order([&]() {
            segmented_wave_commit<SYM>(has_hi && !handoff, key_hi, hi_s, hi_b, lds_s, lds_b);
            segmented_wave_commit<SYM>(adds_lo, key_lo, lo_s, lo_b, lds_s, lds_b);
        });
    } else {
        // lanes -> window slots.  Dead lanes carry key -1 (never written).
        bool adds[LC::N];
#pragma unroll
        for (int j = 0; j < LC::N; ++j) adds[j] = segmented_wave_reduce<SYM>(site.counts ? ch.key[j] : -1, acc_s[j], acc_b[j]);
        in_wave_order([&]() {
#pragma unroll
            for (int j = 0; j < LC::N; ++j) segmented_wave_commit<SYM>(adds[j], ch.key[j], acc_s[j], acc_b[j], lds_s, lds_b);
        });
    }
    if constexpr (OWN) {
        // every element of these channels went through this workgroup: the slots are the channels' totals
        for (int k = threadIdx.x; k < g.k_slots; k += static_cast<int>(blockDim.x)) {
            const int64_t c = site.c_lo + k;
            if (c < g.C) {
                double ts = 0.0, tb = 0.0;
                for (uint32_t w = 0; w < sum_sets; ++w) {          // the waves' sums, in wave order
                    ts += lds_s0[(2u * w) * g.k_slots + k];
                    tb += lds_s0[(2u * w + 1u) * g.k_slots + k];
                }
                if (direct.sym) tb = 0.0 + static_cast<double>(direct.sym_term);
                direct.ds[c] = static_cast<T>(ts);
                direct.db[c] = static_cast<T>(tb);
                if (direct.wide) {
                    direct.wide[c] = ts;
                    direct.wide[g.C + c] = tb;
                }
            }
        }
#ifdef LSQ_TIMELINE
        tl_record();
#endif
        return;
    }
    const int64_t block_linear = static_cast<int64_t>(blockIdx.y) * g.n_windows + blockIdx.x;
    double2* out = partials + block_linear * g.k_slots;
    for (int k = threadIdx.x; k < g.k_slots; k += kBlock) out[k] = make_double2(lds_s[k], lds_b[k]);
#ifdef LSQ_TIMELINE
    tl_record();
#endif
#undef LSQ_TL_WAIT
}

// Finalize (window mode): folds, in a fixed order, every (split, window) partial that can hold a piece
// of a channel (the reference's `ds_buffer.sum(axes != axis)`, lsq_cpu.cpp:287-292).  A workgroup
// handles fin_ch channels x (256 / fin_ch) interleaved slices of a channel's partials, so each lane issues only
// a few INDEPENDENT loads (a one-lane-per-channel loop serialised `splits` dependent HBM latencies); the
// slices are then combined through LDS by a fixed-order tree.
constexpr int kFinCh = 32;   // channels per finalize workgroup when there are at least that many

// Channels per finalize workgroup: a power of two, at most 32, chosen so that the finalize grid still has ~256
// workgroups when the channel count allows it: with few channels (RGB inputs; 768 features x 512 row slabs) the
// lanes of a workgroup share a channel's partials -- there can be thousands -- instead of 24 workgroups walking
// them one lane per channel.
static inline int fin_channels(int64_t C) {
    const int o = knob::get(knob::kFinCh);        // tools build only: a power of two <= kFinCh
    if (o > 0) return o > kFinCh ? kFinCh : (o & (o - 1)) ? 1 : o;
    int ch = 1;
    while (ch < kFinCh && static_cast<int64_t>(ch) * 2 * 256 <= C) ch <<= 1;
    // ... but at least 8 channels (128 contiguous bytes of partials per split) where there are that many: better
    // coalescing beats the extra workgroups (profiles/r02_finalize_channels_sweep.txt: 1-2 us on every shape)
    while (ch < 8 && static_cast<int64_t>(ch) * 2 <= C) ch <<= 1;
    return ch;
}

// Fixed-order combination of the kBlock / fin_ch slices of every channel (fin_ch a power of two <= 32, thread t holds
// channel t % fin_ch): a wave64 butterfly over the lane bits above the channel bits, then the four wave results
// through LDS -- one barrier.  Threads 0 .. fin_ch-1 return their channel's total.
__device__ __forceinline__ double2 combine_parts(double2* wave_part, int fin_ch, double s, double b) {
    for (int m = 32; m >= fin_ch; m >>= 1) {
        s += shfl_xor_f64(s, m);
        b += shfl_xor_f64(b, m);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < fin_ch) wave_part[wave * kFinCh + lane] = make_double2(s, b);
    __syncthreads();
    double2 t = make_double2(0.0, 0.0);
    if (threadIdx.x < fin_ch) {
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) {
            t.x += wave_part[w * kFinCh + threadIdx.x].x;
            t.y += wave_part[w * kFinCh + threadIdx.x].y;
        }
    }
    return t;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void finalize_pc_kernel(const double2* __restrict__ partials, PcGeom g, int fin_ch,
                                                             int eval_mode, int sym, T sym_term, T* __restrict__ ds,
                                                             T* __restrict__ db, double* __restrict__ wide) {
    __shared__ double2 wave_part[(kBlock / 64) * kFinCh];
    const int parts = kBlock / fin_ch;
    const int lane_c = threadIdx.x % fin_ch, part = threadIdx.x / fin_ch;
    const int64_t c = static_cast<int64_t>(blockIdx.x) * fin_ch + lane_c;
    double s = 0.0, b = 0.0;
    if (!eval_mode && c < g.C) {
        const bool f32 = g.fits32 != 0;      // row positions fit 32 bits: every division below is a 32-bit one
        int64_t w_lo = 0, w_hi = 0;
        if (g.R == 1) {
            w_lo = udiv(c * g.inner, g.wpos, f32);
            w_hi = udiv((c + 1) * g.inner - 1, g.wpos, f32);
        }
        // the (window, split) pairs holding a piece of channel c, flattened and dealt out to the `parts` lanes of c
        const int64_t total = (w_hi - w_lo + 1) * g.splits;
        const int64_t stride = g.n_windows * g.k_slots;
        const bool idx32 = total < 0x7fffffffLL;
#pragma unroll 4
        for (int64_t idx = part; idx < total; idx += parts) {
            const int64_t wi = udiv(idx, g.splits, idx32);
            const int64_t w = w_lo + wi;
            const int64_t sy = idx - wi * g.splits;
            const int64_t c_lo = (g.R == 1) ? udiv(w * g.wpos, g.inner, f32) : 0;
            const double2 v = partials[sy * stride + w * g.k_slots + (c - c_lo)];
            s += v.x;
            b += v.y;
        }
    }
    const double2 t = combine_parts(wave_part, fin_ch, s, b);
    if (part == 0 && c < g.C) {
        double ts = t.x, tb = t.y;
        if (!eval_mode && sym) tb = 0.0 + static_cast<double>(sym_term);
        ds[c] = static_cast<T>(ts);
        db[c] = static_cast<T>(tb);
        if (wide) {
            wide[c] = ts;
            wide[g.C + c] = tb;
        }
    }
}

// Finalize (row-group windows): the partials are [splits][n_windows * w V] in slot order (slot = component * w + lane
// inside a window), so consecutive threads read consecutive 16-byte partials; thread -> slot -> channel
// c = window * w V + lane * V + component.  fin_ch slots x (256 / fin_ch) interleaved slices of the splits per
// workgroup, fixed-order combination as in finalize_pc_kernel.
template <typename T>
__global__ __launch_bounds__(kBlock) void finalize_ww_kernel(const double2* __restrict__ partials, PcGeom g, int fin_ch,
                                                             int eval_mode, int sym, T sym_term, T* __restrict__ ds,
                                                             T* __restrict__ db, double* __restrict__ wide) {
    __shared__ double2 wave_part[(kBlock / 64) * kFinCh];
    const int parts = kBlock / fin_ch;
    const int lane_c = threadIdx.x % fin_ch, part = threadIdx.x / fin_ch;
    const uint32_t k_slots = static_cast<uint32_t>(g.k_slots), w = static_cast<uint32_t>(g.ww_lanes);
    const uint32_t total_slots = static_cast<uint32_t>(g.n_windows) * k_slots;     // ~ the channel count: fits 32 bits
    const uint32_t gslot = blockIdx.x * static_cast<uint32_t>(fin_ch) + lane_c;
    const uint32_t win = gslot / k_slots, k = gslot - win * k_slots;
    const uint32_t comp = k / w, lane = k - comp * w;
    const int64_t c = static_cast<int64_t>(win) * k_slots + static_cast<int64_t>(lane) * g.vec + comp;
    const bool valid = gslot < total_slots && c < g.C;
    double s = 0.0, b = 0.0;
    if (!eval_mode && valid) {
        const double2* col = partials + gslot;
#pragma unroll 4
        for (int sy = part; sy < g.splits; sy += parts) {
            const double2 v = col[static_cast<int64_t>(sy) * total_slots];
            s += v.x;
            b += v.y;
        }
    }
    const double2 t = combine_parts(wave_part, fin_ch, s, b);
    if (part == 0 && valid) {
        double ts = t.x, tb = t.y;
        if (!eval_mode && sym) tb = 0.0 + static_cast<double>(sym_term);
        ds[c] = static_cast<T>(ts);
        db[c] = static_cast<T>(tb);
        if (wide) {
            wide[c] = ts;
            wide[g.C + c] = tb;
        }
    }
}

// =================================================================================================
// SEGMENT mode: one channel per workgroup
// =================================================================================================
template <typename IO, int V, bool INIT, bool LEVELS, int UNROLL, bool NTL, bool NTS, int WALK>
__global__ __launch_bounds__(kBlock) void fwd_seg_kernel(const void* __restrict__ x, void* __restrict__ y,
                                                         int8_t* __restrict__ levels, int level_bias, int aux_kind, SegGeom g,
                                                         const typename IO::arith* __restrict__ scale,
                                                         const typename IO::arith* __restrict__ shift,
                                                         Range<typename IO::arith> r) {
    seg_forward<IO, V, INIT, LEVELS, UNROLL, NTL, NTS, WALK>(x, y, levels, level_bias, aux_kind, g, SegWalk(g), scale, shift, r);
}

// (SegDirect, seg_forward, seg_backward: lsq_seg_body.hpp)
template <typename IO, int V, bool SYM, bool INIT, bool EVAL, int UNROLL, bool NTL, bool NTS, int WALK>
__global__ __launch_bounds__(kBlock) void bwd_seg_kernel(const void* __restrict__ grad, const void* __restrict__ x,
                                                         void* __restrict__ dx, SegGeom g,
                                                         const typename IO::arith* __restrict__ scale,
                                                         const typename IO::arith* __restrict__ shift,
                                                         Range<typename IO::arith> r, typename IO::arith grad_scaler,
                                                         double2* __restrict__ partials,
                                                         SegDirect<typename IO::arith> direct) {
    seg_backward<IO, V, SYM, INIT, EVAL, UNROLL, NTL, NTS, WALK>(grad, x, dx, g, SegWalk(g), scale, shift, r, grad_scaler, partials,
                                                           static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x, direct);
}

// Finalize (segment mode): fin_ch channels x (256 / fin_ch) interleaved slices of the (osplit, seg) partials.
template <typename T>
__global__ __launch_bounds__(kBlock) void finalize_seg_kernel(const double2* __restrict__ partials, SegGeom g, int fin_ch,
                                                              int eval_mode, int sym, T sym_term, T* __restrict__ ds,
                                                              T* __restrict__ db, double* __restrict__ wide) {
    __shared__ double2 wave_part[(kBlock / 64) * kFinCh];
    const int parts = kBlock / fin_ch;
    const int lane_c = threadIdx.x % fin_ch, part = threadIdx.x / fin_ch;
    const int64_t c = static_cast<int64_t>(blockIdx.x) * fin_ch + lane_c;
    double s = 0.0, b = 0.0;
    if (!eval_mode && c < g.C) {
        const int64_t gx = g.C * g.segs;
        const int32_t total = g.osplits * g.segs;
#pragma unroll 4
        for (int32_t sl = part; sl < total; sl += parts) {
            const int32_t oy = sl / g.segs, sg = sl - oy * g.segs;
            const double2 v = partials[static_cast<int64_t>(oy) * gx + c * g.segs + sg];
            s += v.x;
            b += v.y;
        }
    }
    const double2 t = combine_parts(wave_part, fin_ch, s, b);
    if (part == 0 && c < g.C) {
        double ts = t.x, tb = t.y;
        if (!eval_mode && sym) tb = 0.0 + static_cast<double>(sym_term);
        ds[c] = static_cast<T>(ts);
        db[c] = static_cast<T>(tb);
        if (wide) {
            wide[c] = ts;
            wide[g.C + c] = tb;
        }
    }
}

// =================================================================================================
// host-side launchers
// =================================================================================================
// Packets in flight per lane in the segment kernels (profiles/r01_segment_sweep.txt, typical weight shapes): 4 for
// 4/8-byte storage; 16-bit storage (twice the arithmetic and registers per packet, half the packets per channel --
// [4096, 4096] in bf16 is two packets per lane) runs best at 1: [32000, 4096] bf16 backward 125 us vs 143 us at 4.
// Elements per lane per row in the window-mode backward.  Half packets (4 elements, 8 bytes per lane) for 16-bit
// storage were measured: 103 instead of 156 VGPRs, but no faster at BASELINE config 5 (best 35.0 us vs 36.3 us, within
// the run-to-run spread: the dx-only kernel shows the access pattern itself tops out near 5.3 TB/s there), so every
// storage type moves full packets; load_elems / store_elems keep the 8-byte path.
template <typename IO>
constexpr int kWindowBwdVec = IO::VEC;
constexpr int kLastAxisBwdBlocksPerCU = 2;
// (tools-build knobs consulted below, all 0 in the production library -- lsq_kernels.hpp `knob`: kWwMinRows = rows a
// row-group-window workgroup walks at least, 0 = kWwMinRows<IO>; kWwSplit64 = rows of 128 / 192 / 256 lanes as 64-lane windows;
// kRingNt = streaming hint on the ring's copies, 0 = policy, 1 = on, 2 = off; kWwBig = 768/1024-lane workgroups, 0 = policy,
// 1 = always, 2 = never)
// Policy: on for the BACKWARD of tensors of more than 32 MB -- the x a backward reads was saved by a forward long ago and
// is not in the 256 MB Infinity Cache any more, whatever the gradient is, and nt copies still hit the lines a producer left
// there.  256-lane windows, cold (profiles/r02_ring_nt_ab.txt): config 5 fp32 64.9 -> 59.8 us, bf16 (51 MB) 37.1 -> 34.3 us.
// Row-group windows with the gradient fresh from a producer kernel and x cold (profiles/r02_ww_nt_ab.txt): [8192,4096] fp32
// 78 -> 68 us, [256,197,768] fp32 91 -> 80 us, [65536,1024] bf16 82 -> 76 us; never slower, cold included.  (An earlier A/B that
// found the hint harmful for row groups and for the forward re-read one set of buffers: the hint kept them out of the cache.)
static inline int ring_nt_for(int64_t tensor_bytes, bool backward, bool row_groups) {
    const int k = knob::get(knob::kRingNt);
    if (k != 0) return k == 1 ? 1 : 0;
    (void)row_groups;
    return backward && tensor_bytes > (int64_t{32} << 20) ? 1 : 0;
}
// (16-bit storage: 768 lanes -- its kernel needs ~140 registers, 1024 lanes would cap it at 128 and spill)
template <int ELEM_BYTES>
constexpr int kBigBlockOf = ELEM_BYTES < 4 ? 768 : 1024;
// Owner windows (plan_own): launch bound of their kernels (the workgroup is R x lanes-per-row threads, at most this: eight
// waves, so the 16-bit kernel keeps its ~120 registers without spilling) and the tensor size up to which the policy takes
// them.  Measured with one owner per CU (plan_own's fattest channel group; profiles/r04_owner_windows_ab2.txt, backward op,
// cold): they win where the finalize launch is a large share of the op -- [64,2048,7,7] bf16 15.6 -> 11.9 us, fp32 22.9 ->
// 18.5; [16,1024,14,14] fp32 17.2 -> 9.9 -- stay ahead in 16-bit storage up to 19 M elements ([192,2048,7,7] 28.2 -> 27.2,
// [96,1024,14,14] 28.7 -> 26.8) and in fp32 up to 12.8 M ([128,2048,7,7] 33.4 -> 32.9, [32,512,28,28] 33.9 -> 31.2), and are
// behind from there: fp32 16 M +2 %, 19 M +5 %, BASELINE config 5 (25.7 M) bf16 33.8 -> 35.6 us, fp32 59.8 -> 65.7 -- every owner
// walks the same rows at the same time and the access pattern tops out at 5.4 TB/s (profiles/r04_owner_pattern_probe.txt),
// where the row slabs of the 256-lane windows spread the chip over the whole tensor.
// Short runs (channel rows of a few positions: 1-D feature maps, 3x3 ... 10x10) are fine -- [128,2048,4,4] fp32 17.9 -> 11.9 us,
// [512,2048,8] 26.2 -> 23.1, [128,2048,5,5] 25.5 -> 19.4, bf16 [512,2048,8] 18.3 -> 13.7 -- unless they are under 512 bytes AND
// not whole 128-byte lines: every row of every owner then shares a partial line with its neighbours ([rows,2048,7] fp32, eight
// channels = 224 bytes: 384 rows 21.2 -> 20.4 us, 512 rows 24.9 -> 25.5, 768 rows 32.2 -> 39.3); those only up to 5 * 2^20
// elements ([256,2048,7] 17.0 -> 11.8 us, [292,2048,7] 18.2 -> 13.8, [192,2048,3,3] 15.5 -> 11.1).  profiles/r04_owner_short_runs.txt, r04_owner_min_run.txt.
constexpr int kOwnBlock = 512;
constexpr size_t kLdsBytesPerWorkgroup = 160 * 1024;      // gfx950: LDS a workgroup may allocate (the Makefile builds for gfx950 only)
template <int ELEM_BYTES>
constexpr int64_t kOwnMaxElemsOf = ELEM_BYTES < 4 ? int64_t{20} << 20 : (ELEM_BYTES == 4 ? int64_t{13} << 20 : int64_t{1} << 23);
constexpr int64_t kOwnMaxElemsShortRun = int64_t{5} << 20;
constexpr int kOwnShortRunBytes = 512;
constexpr int kWwBwdBlocksPerCU = 4;     // row-group windows: one full round for every storage type (3-4 resident per CU)
// Rows a forward workgroup walks at least, per unit of its per-workgroup overhead (make_geom): that overhead is only
// the channel-table build -- VEC channels per lane when the quantized axis is the last one, so twice as heavy per
// streamed byte for 16-bit storage.  ([64,197,768] fp32 forward 17.8 -> 13.7 us, bf16 13.8 -> 11.4 us against the
// backward's bound of 27.)  A variant without the table (every lane computing its own channels' constants, no LDS,
// no barrier) was measured too and is slower: its scale/shift loads are strided by VEC across the lanes.
template <typename IO>
constexpr int kFwdPerSlotRows = sizeof(typename IO::elem) < 4 ? 12 : 4;
template <typename IO>
constexpr int kSegUnroll = sizeof(typename IO::elem) < 4 ? 1 : 4;
static inline int pick_cpl(int vec, int64_t inner) {
    if (vec == 1 || inner % vec == 0) return 1;
    return inner >= vec ? 2 : vec;
}
#ifdef LSQ_TOOLS
// Tools build: a caller may force any launch variant or knob, so the size is the maximum over every geometry the tuning
// range allows (tens of milliseconds of host time; lsq_capi.hip memoises it).
static size_t bwd_pc_workspace_bytes_any(int io_vec, int64_t outer, int64_t channels, int64_t inner) {
    const DeviceInfo& dev = device_info();
    size_t need = 0;
    const int vecs[3] = {io_vec, 1, 4};   // full packets, single elements, half packets (16-bit window backward)
    for (int vi = 0; vi < 3; ++vi) {
        for (int bpc = 1; bpc <= 2 * kMaxBlocksPerCU; ++bpc) {   // pick_splits may go up to twice the requested count
            for (int res = 0; res <= 8; ++res) {   // residency of the instantiation that will run: 0 (not used) .. 8 per CU
                const PcGeom g = make_geom(outer, channels, inner, vecs[vi], dev.cu_count * bpc, 27, dev.cu_count * res);
                need = std::max(need, static_cast<size_t>(g.splits) * g.n_windows * g.k_slots * sizeof(double2));
            }
            if (vi == 0 && inner == 1 && io_vec > 1 && channels % io_vec == 0) {
                const int ovr = knob::get(knob::kWwMinRows);
                const int min_rows = ovr > 0 ? ovr : (io_vec > 4 ? 16 : (107 + (16 / io_vec) - 1) / (16 / io_vec));   // kWwMinRows of the storage type
                for (int res = 0; res <= 8; ++res) {
                    for (int s64 = 0; s64 < 3; ++s64) {     // whole rows, 64-lane windows, 1024-lane workgroups
                        const PcGeom g = make_geom_ww(outer, channels, io_vec, dev.cu_count * bpc, min_rows, dev.cu_count * res, s64 == 1,
                                                      s64 == 2 ? (io_vec > 4 ? kBigBlockOf<2> : kBigBlockOf<4>) : kBlock);
                        need = std::max(need, static_cast<size_t>(g.splits) * g.n_windows * g.k_slots * sizeof(double2));
                    }
                }
            }
            if (vi < 2 && pick_segment_mode(vecs[vi], outer, channels, inner, dev.cu_count)) {
                const SegGeom sgm = make_seg_geom(outer, channels, inner, vecs[vi], dev.cu_count * bpc);
                need = std::max(need, static_cast<size_t>(channels) * sgm.segs * sgm.osplits * sizeof(double2));
            }
        }
    }
    return need + 256;
}


#endif

// Scratch bytes of the backward: exactly what backward_per_channel (the library's own launch policy) asks for -- the same
// code run as a PLAN (BwdPcCall::plan_need: geometry and kernel choice, nothing launched) for the argument properties the
// size does not take: symmetric or not, init mode or not, 16-byte-aligned buffers or not; eval mode needs none.  A few
// microseconds of host time (eight plans).
template <typename IO>
size_t bwd_pc_workspace_bytes(int64_t outer, int64_t channels, int64_t inner) {
    size_t need = 0;
#ifdef LSQ_TOOLS
    need = bwd_pc_workspace_bytes_any(IO::VEC, outer, channels, inner);
#else
    lsq_params p{};
    p.quant_min = 0; p.quant_max = 127; p.type_min = 0; p.type_max = 255;
    p.use_grad_scaling = 1; p.grad_scaler = 1.0; p.numel_for_scaler = 0;
    for (int al = 0; al < 2; ++al) {
        // only the alignment of the (never dereferenced) buffer addresses matters to the plan
        void* const fake = reinterpret_cast<void*>(static_cast<uintptr_t>(al ? 4096 + sizeof(typename IO::elem) : 4096));
        for (int mode = 0; mode < 4; ++mode) {
            p.sym = mode & 1;
            p.init_mode = (mode >> 1) & 1;
            (void)backward_per_channel<IO>(fake, fake, fake, fake, fake, nullptr, outer, channels, inner, fake, fake, p, fake, 0,
                                           nullptr, 0, nullptr, &need);
        }
    }
#endif
    return need + 256;
}

// LDS-DMA ring in the window-mode kernels by default, with the grid it likes: fewer, longer workgroups than the register
// loops (it needs rows to keep its ring full).  A/B on one box, profiles/r02_dma_ab.txt: 8-16 % faster on every large shape
// in both directions when the tensors are cache-resident; on cold buffers (profiles/r02_cold_buffers_pc.txt) it keeps that
// lead for 16-bit storage only, hence the size rules further down (forward_per_channel, ring_nt_for).
template <typename IO>
constexpr bool kDmaDefault = true;
template <typename IO>
constexpr int kDmaBwdBlocksPerCU = sizeof(typename IO::elem) < 4 ? 4 : 8;
template <typename IO>
constexpr int kDmaFwdBlocksPerCU = sizeof(typename IO::elem) < 4 ? 4 : 8;
constexpr int kFwdDmaDepth = 8;      // one 1 KiB stage per row and wave in the forward (x only); the backward rings are 4 deep

// ---- forward --------------------------------------------------------------------------------------
template <typename IO, int V, int CPL, bool INIT, bool LEVELS>
static hipError_t launch_fwd_pc(const void* x, void* y, int8_t* levels, int bias, int aux_kind, const PcGeom& g, const void* scale,
                                const void* shift, const lsq_params& p, const Variant& v, hipStream_t stream) {
    using T = typename IO::arith;
    const Range<T> r = make_range<T>(p);
    const dim3 grid(static_cast<unsigned>(g.n_windows), static_cast<unsigned>(g.splits));
    const size_t table_bytes = static_cast<size_t>(g.k_slots) * sizeof(QSlot<T>);
    const size_t lds = g.direct ? 0 : table_bytes;     // (direct: every lane reads its own channels)
    // LDS-DMA ring (16-byte packets): forward_per_channel decided (v.dma == 2) and sized the grid for it
    constexpr bool kDmaAble = V * sizeof(typename IO::elem) == 16;
    constexpr int kDmaDepth = kFwdDmaDepth;
    if constexpr (kDmaAble) {
        const size_t lds_dma = ((table_bytes + 1023) & ~size_t(1023)) + static_cast<size_t>(kBlock / 64) * kDmaDepth * 1024;
        if (v.dma == 2 && lds_dma <= 64 * 1024) {
            hipLaunchKernelGGL((fwd_pc_kernel<IO, V, CPL, INIT, LEVELS, 1, true, true, kDmaDepth>), grid, dim3(kBlock), lds_dma, stream,
                               x, y, levels, bias, aux_kind, g, static_cast<const T*>(scale), static_cast<const T*>(shift), r);
            return hipGetLastError();
        }
    }
#define LSQ_LAUNCH(U, NTLF, NTSF)                                                                                       \
    hipLaunchKernelGGL((fwd_pc_kernel<IO, V, CPL, INIT, LEVELS, U, NTLF, NTSF>), grid, dim3(kBlock), lds, stream, x, y, levels, \
                       bias, aux_kind, g, static_cast<const T*>(scale), static_cast<const T*>(shift), r)
    [[maybe_unused]] constexpr bool kFull = !INIT && !LEVELS && V > 1 && !std::is_same<IO, io_f64>::value &&
                                            !std::is_same<IO, io_f16>::value;
    LSQ_DISPATCH_VARIANT(kFull, 4, v, LSQ_LAUNCH);
#undef LSQ_LAUNCH
    return hipGetLastError();
}

template <typename IO, bool INIT, bool LEVELS>
static hipError_t launch_fwd_seg(const void* x, void* y, int8_t* levels, int bias, int aux_kind, const SegGeom& g, const void* scale,
                                 const void* shift, const lsq_params& p, const Variant& v, hipStream_t stream) {
    using T = typename IO::arith;
    const Range<T> r = make_range<T>(p);
    const dim3 grid(static_cast<unsigned>(g.C * g.segs), static_cast<unsigned>(g.osplits));
    // the most iterations a workgroup walks: short walks (a weight's channel) and long ones are two kernels (seg_forward)
    // (not for a big grid whose iterations are rows 16 MB apart instead of neighbouring sub-rows: [4,8,1048576] bf16 forward
    // 23.7 us with the loop, 26.4 us with the group -- profiles/r03_seg_up_front_ab.txt)
    const bool short_walk = g.sub_per_seg * g.o_per_split <= kSegUpFront && knob::get(knob::kSegNoUpFront) == 0 &&
                            (g.o_per_split == 1 || g.C * g.segs * g.osplits <= 8 * static_cast<int64_t>(device_info().cu_count));
#define LSQ_LAUNCH(U, NTLF, NTSF)                                                                                              \
    do {                                                                                                                       \
        if (short_walk)                                                                                                        \
            hipLaunchKernelGGL((fwd_seg_kernel<IO, IO::VEC, INIT, LEVELS, U, NTLF, NTSF, 1>), grid, dim3(kBlock), 0, stream, x, y, \
                               levels, bias, aux_kind, g, static_cast<const T*>(scale), static_cast<const T*>(shift), r);     \
        else                                                                                                                   \
            hipLaunchKernelGGL((fwd_seg_kernel<IO, IO::VEC, INIT, LEVELS, U, NTLF, NTSF, 2>), grid, dim3(kBlock), 0, stream, x, y, \
                               levels, bias, aux_kind, g, static_cast<const T*>(scale), static_cast<const T*>(shift), r);     \
    } while (0)
    [[maybe_unused]] constexpr bool kFull = !INIT && !LEVELS && (std::is_same<IO, io_f32>::value || std::is_same<IO, io_bf16>::value);
    LSQ_DISPATCH_VARIANT(kFull, kSegUnroll<IO>, v, LSQ_LAUNCH);
#undef LSQ_LAUNCH
    return hipGetLastError();
}

template <typename IO, int V, int CPL>
static hipError_t fwd_pc_modes(const void* x, void* y, int8_t* levels, int bias, int aux_kind, const PcGeom& g, const void* scale,
                               const void* shift, const lsq_params& p, const Variant& v, hipStream_t stream) {
    if (p.init_mode) {
        return levels ? launch_fwd_pc<IO, V, CPL, true, true>(x, y, levels, bias, aux_kind, g, scale, shift, p, v, stream)
                      : launch_fwd_pc<IO, V, CPL, true, false>(x, y, levels, bias, aux_kind, g, scale, shift, p, v, stream);
    }
    return levels ? launch_fwd_pc<IO, V, CPL, false, true>(x, y, levels, bias, aux_kind, g, scale, shift, p, v, stream)
                  : launch_fwd_pc<IO, V, CPL, false, false>(x, y, levels, bias, aux_kind, g, scale, shift, p, v, stream);
}

template <typename IO>
hipError_t forward_per_channel(const void* x, void* y, int64_t outer, int64_t channels, int64_t inner,
                               const void* scale, const void* shift, const lsq_params& p,
                               const lsq_fwd_extras* ex, int variant, hipStream_t stream) {
    int8_t* levels = ex ? static_cast<int8_t*>(ex->levels) : nullptr;
    const int bias = ex ? ex->level_bias : 0;
    const int aux_kind = ex ? ex->aux_kind : 0;
    const DeviceInfo& dev = device_info();
    // Packets need ELEMENT alignment only (lsq_math.hpp, PacketWord): a sliced view runs the packet kernels' register loops.
    // What wants 16-byte sources is the LDS-DMA ring (ring_ok); the level bytes of a packet are stored as one word.
    const bool aligned = is_elem_aligned<IO>(x) && is_elem_aligned<IO>(y) && (!levels || (reinterpret_cast<uintptr_t>(levels) & 7u) == 0);
    const bool ring_ok = is_aligned16(x) && is_aligned16(y);
    const int vec = pick_vec(IO::VEC, channels * inner, aligned);
    const bool seg = pick_segment_mode(vec, outer, channels, inner, dev.cu_count);
    const Variant v = decode_variant(variant, seg ? (sizeof(typename IO::elem) >= 4 ? kDefaultPcSegVariant : kDefaultPcSegNarrowVariant)
                                                  : kDefaultPcFwdVariant);
    const int target = dev.cu_count * v.blocks_per_cu;
    if (seg) {
        const SegGeom sg = make_seg_geom(outer, channels, inner, vec, target);
        if (!grid_fits(sg)) return hipErrorInvalidConfiguration;
#ifdef LSQ_TOOLS
        last_launch_note() = LaunchNote{static_cast<int>(sg.C * sg.segs), sg.osplits, 0, 0, 3, 0, kBlock, 0};
#endif
        if (p.init_mode) {
            return levels ? launch_fwd_seg<IO, true, true>(x, y, levels, bias, aux_kind, sg, scale, shift, p, v, stream)
                          : launch_fwd_seg<IO, true, false>(x, y, levels, bias, aux_kind, sg, scale, shift, p, v, stream);
        }
        return levels ? launch_fwd_seg<IO, false, true>(x, y, levels, bias, aux_kind, sg, scale, shift, p, v, stream)
                      : launch_fwd_seg<IO, false, false>(x, y, levels, bias, aux_kind, sg, scale, shift, p, v, stream);
    }
    const int cpl = pick_cpl(vec, inner);
    PcGeom g = make_geom(outer, channels, inner, vec, target, kFwdPerSlotRows<IO>);
    // The forward's LDS-DMA ring is NOT a default any more: it wins only when the same buffers are read again and again
    // (profiles/r02_dma_ab.txt: config 5 fp32 35.2 -> 32.2 us).  On input the previous kernel has just written
    // (profiles/r02_producer_consumer.txt: config 5 bf16 14.2 us with the register loops at 16 workgroups per CU, 17.5 us on
    // the ring; [32,256,56,56] bf16 14.3 vs 18.3 us) and on cold input (profiles/r02_cold_buffers_pc.txt: 6-14 % behind for
    // every storage type) the register loops are faster.  Variant bits 12-13 = 2 still select it (A/B runs, tests).
    // What stays from its tuning is the grid for one case: a 16-bit last-axis window has 2048 channels, a 32 KiB table, and
    // building half as many tables pays -- those shapes take the ring's grid (4 workgroups per CU) with the register loop
    // ([8192,4096] bf16 28.8 us against 34.3 us on the usual grid, cold 29.3 vs 35.3 us, after a producer 21.8 vs 24.7 us;
    // profiles/r02_fwd_lastaxis_grid.txt).
    Variant vv = v;
    vv.dma = 1;
    // A lane whose components are different channels (cpl == vec: the quantized axis is the last or nearly the last one)
    // reads its own scale / shift: no LDS table (fwd_pc_kernel, LaneChannels::load_direct) -- profiles/r03_fwd_direct_ab.txt,
    // cold: [12608,768] bf16 11.9 -> 9.5 us, [8192,4096] bf16 29.6 -> 26.8 us, [65536,1024] bf16 51.4 -> 46.6 us, [3152,768]
    // fp32 8.8 -> 7.1 us, the big fp32 tensors -1 .. -4 %.
    // tools builds, lsq_hip_debug_set_fwd_direct: 1 = direct on the usual grid, 2 = the table, 3 = the policy
    const int direct_knob = knob::get(knob::kFwdDirect);
    // tools knob 4: also the lanes of one or two channels (cpl < vec) -- A/B
    const bool direct = direct_knob == 4 || (vec > 2 && cpl == vec && direct_knob != 2);
    g.direct = direct ? 1 : 0;
    // (the grid rule that was found for the 32 KiB table stays on the direct path: [8192,4096] bf16 26.8 us against 27.5 us on
    // the usual grid, [16384,8192] 97.0 vs 99.5 us -- profiles/r03_fwd_direct_ab.txt)
    const bool table_grid_rule = direct_knob != 1;
    if (vec > 1 && vec * sizeof(typename IO::elem) == 16 && v.dma != 1 && (table_grid_rule || v.dma == 2)) {
        const int tgt = variant == 0 ? dev.cu_count * kDmaFwdBlocksPerCU<IO> : target;
        const PcGeom gd = make_geom(outer, channels, inner, vec, tgt, kFwdPerSlotRows<IO>);
        const int64_t tiles_each = gd.n_tiles / std::max(1, gd.splits);
        const size_t lds_ring = ((static_cast<size_t>(gd.k_slots) * sizeof(QSlot<typename IO::arith>) + 1023) & ~size_t(1023)) +
                                static_cast<size_t>(kBlock / 64) * kFwdDmaDepth * 1024;
        const bool table_big = lds_ring > 64 * 1024;
        // (tiles_each >= 8 on the ring's grid with a 2048-slot table already implies >= 2^24 elements: no separate size rule)
        if (v.dma == 2 || (table_big && tiles_each >= kFwdDmaDepth && tiles_each <= 64)) {
            g = gd;
            vv.dma = (table_big || !ring_ok) ? 1 : 2;
            g.direct = (direct && vv.dma == 1) ? 1 : 0;
            g.ring_nt = ring_nt_for(outer * channels * inner * static_cast<int64_t>(sizeof(typename IO::elem)), false, false);
        }
    }
    if (!grid_fits(g)) return hipErrorInvalidConfiguration;
#ifdef LSQ_TOOLS
    last_launch_note() = LaunchNote{static_cast<int>(g.n_windows), g.splits, 0, 0, 1, vv.dma == 2 ? kFwdDmaDepth : 0, kBlock, g.ring_nt};
#endif
    if (vec == 1) return fwd_pc_modes<IO, 1, 1>(x, y, levels, bias, aux_kind, g, scale, shift, p, vv, stream);
    if (cpl == 1) return fwd_pc_modes<IO, IO::VEC, 1>(x, y, levels, bias, aux_kind, g, scale, shift, p, vv, stream);
    if (cpl == 2) return fwd_pc_modes<IO, IO::VEC, 2>(x, y, levels, bias, aux_kind, g, scale, shift, p, vv, stream);
    return fwd_pc_modes<IO, IO::VEC, IO::VEC>(x, y, levels, bias, aux_kind, g, scale, shift, p, vv, stream);
}

// ---- backward -------------------------------------------------------------------------------------
// Everything a window-mode backward launch needs besides the kernel's template arguments.
template <typename T>
struct BwdPcCall {
    const void* grad;
    const void* x;
    void* dx;
    T* ds;
    T* db;
    double* wide;
    int64_t outer, C, inner;
    const void* scale;
    const void* shift;
    const lsq_params* p;
    T gs, sym_term;
    double2* partials;
    size_t workspace_bytes;
    int target_blocks;     // requested workgroups (CUs x workgroups per CU)
    bool default_variant;  // the caller passed variant 0: the launcher may pick the grid of the code path it chooses
    bool whole_rounds;     // size the grid in whole rounds of what the chip holds at once (make_geom)
    bool ring_ok;          // grad / x / dx are 16-byte aligned: the LDS-DMA ring (and the owner windows built on it) may be used
    Variant v;
    hipStream_t stream;
    size_t* plan_need;     // not null: PLAN only -- record the workspace bytes the launch would need, launch nothing
    LaunchNote* plan_note; // PLAN only, may be null: what the launch would look like (lsq_hip_plan_backward_per_channel)
};


// rows a row-group-window workgroup walks at least.  4- and 8-byte storage: enough to keep its 16-byte-per-slot partial
// row under ~5 % of what it streams.  16-bit storage: 16 -- the tensors this floor binds on (fewer rows than workgroups
// wanted x floor) are latency-bound, every row a wave walks is another ~0.6 us on its serial chain, and the extra partial
// bytes cost less than that ([3152,768] bf16 21 -> 13.5 us, [4096,1024] 22 -> 14.6 us, profiles/r02_ww_rows_per_workgroup.txt)
template <typename IO>
constexpr int kWwMinRows = sizeof(typename IO::elem) < 4
                               ? 16
                               : (107 + static_cast<int>(sizeof(typename IO::elem)) - 1) / static_cast<int>(sizeof(typename IO::elem));
template <typename IO>
static inline int ww_min_rows() {
    const int o = knob::get(knob::kWwMinRows);
    return o > 0 ? o : kWwMinRows<IO>;
}

template <typename IO, int V, int CPL, bool SYM, bool INIT, bool EVAL, bool WW = false>
static hipError_t launch_bwd_pc(const BwdPcCall<typename IO::arith>& c) {
    using T = typename IO::arith;
    const Range<T> r = make_range<T>(*c.p);
    const lsq_params& p = *c.p;
    // (tuning builds also compile the variant table of the dx-only EVAL kernel: the streaming rate of the access pattern)
    [[maybe_unused]] constexpr bool kFull = !SYM && !INIT && V > 1 && !std::is_same<IO, io_f64>::value &&
                                            !std::is_same<IO, io_f16>::value;
    // 16-bit storage: unroll 1 + the software-pipelined loop (profiles/r01_pc_pipeline_sweep.txt: 36.3 us against
    // 38.5 us for the best plain variant at BASELINE config 5); 4/8-byte storage gains nothing from it (55.6 vs 55.9 us)
    // and keeps the plain loop at unroll 4.
    // CPL == V (inner < V: the quantized axis is the last or nearly the last one -- [tokens, features], NHWC): 16-bit
    // storage runs the pipelined loop at unroll 2 there (profiles/r01_lastaxis_sweep.txt).
    constexpr bool kNarrow = sizeof(typename IO::elem) < 4;
    constexpr int kDefU = kNarrow ? ((CPL == V && V > 1 && !WW) ? 2 : 1) : 4;
    hipError_t result = hipSuccess;
    // The geometry depends on how many workgroups of the chosen instantiation fit on the chip at once, so it is built
    // here, where the kernel is known, together with the launch and the finalize.
    // returns false (nothing launched) when `min_tiles` is asked for and a workgroup would walk fewer row tiles than that
    auto run = [&](auto kern, int dma_depth, int target_blocks, int64_t min_tiles, int64_t max_tiles = INT64_MAX,
                   int block = kBlock) -> bool {
        const DeviceInfo& dev = device_info();
        auto geom = [&](int resident) {
            // rows of 128 / 192 / 256 lanes: 4- and 8-byte storage cuts them into 64-lane windows of four row groups
            // ([65536,1024] fp32 backward 157 -> 140 us, profiles/r02_ww_split64_ab.txt); 16-bit storage gains nothing
            const bool split64 = sizeof(typename IO::elem) >= 4 ? knob::get(knob::kWwSplit64) != 2 : knob::get(knob::kWwSplit64) == 1;
            return WW ? make_geom_ww(c.outer, c.C, V, target_blocks, ww_min_rows<IO>(), resident, split64, block)
                      : make_geom(c.outer, c.C, c.inner, V, target_blocks, 27, resident);
        };
        auto lds_of = [&](const PcGeom& gg) {
            // row groups: every group parks its sums ([R][k_slots] double2); narrow windows add one row per walk of a slot
            size_t b = WW ? (static_cast<size_t>(gg.R) * gg.k_slots + (gg.k_slots < gg.block_threads ? gg.block_threads : 0)) * sizeof(double2)
                          : static_cast<size_t>(gg.k_slots) * (sizeof(QSlot<T>) + 2 * sizeof(double));
            if (dma_depth > 0) {
                const size_t ring = bwd_lds_front_bytes(gg, sizeof(QSlot<T>)) +
                                    static_cast<size_t>(gg.block_threads / 64) * dma_depth * kDmaStageBytes;
                b = WW ? std::max(b, ring) : ring;      // row groups: the combine buffer reuses the ring's LDS
            }
            return b;
        };
        // the LDS a workgroup needs does not depend on the split count: size it first, then the residency, then the grid
        const PcGeom g0 = geom(0);
        const size_t lds = lds_of(g0);
        // no room for the ring next to a very wide channel table: register loop (a 1024-lane workgroup has the CU to itself)
        if (dma_depth > 0 && lds > (block > kBlock ? 160 : 64) * 1024) return false;
        if (lds > 160 * 1024) { result = hipErrorInvalidConfiguration; return true; }   // (gfx950: 160 KiB of LDS per workgroup)
        int per_cu = c.whole_rounds ? resident_blocks_per_cu(reinterpret_cast<const void*>(kern), lds) : 0;
        if (block > kBlock && per_cu > 0) {      // the register bound counts four-wave workgroups: convert it
            const int by_regs = resident_blocks_by_registers(reinterpret_cast<const void*>(kern)) * 4 / std::max(1, g0.block_threads / 64);
            per_cu = std::max(1, std::min(per_cu, by_regs));
        }
        PcGeom g = geom(per_cu * dev.cu_count);
        g.ring_nt = ring_nt_for(c.outer * c.C * c.inner * static_cast<int64_t>(sizeof(typename IO::elem)), true, WW);
#if defined(LSQ_TOOLS) && defined(LSQ_TIMELINE)
        g.timeline = knob::timeline_buffer().load();
#endif
        const int64_t tiles_each = g.n_tiles / std::max<int64_t>(1, g.splits);
        if (tiles_each < min_tiles || tiles_each > max_tiles) return false;
        if (!grid_fits(g)) { result = hipErrorInvalidConfiguration; return true; }
        const size_t need = static_cast<size_t>(g.splits) * g.n_windows * g.k_slots * sizeof(double2);
        if (c.plan_need) {
            if (!p.eval_mode) *c.plan_need = std::max(*c.plan_need, need);
            if (c.plan_note)
                *c.plan_note = LaunchNote{static_cast<int>(g.n_windows), g.splits, per_cu, registers_of(reinterpret_cast<const void*>(kern)),
                                          WW ? 2 : 1, dma_depth, g.block_threads, g.ring_nt};
            return true;
        }
        if (!p.eval_mode && c.workspace_bytes < need) { result = hipErrorInvalidValue; return true; }
        const dim3 grid(static_cast<unsigned>(g.n_windows), static_cast<unsigned>(g.splits));
#ifdef LSQ_TOOLS
        last_launch_note() = LaunchNote{static_cast<int>(g.n_windows), g.splits, per_cu, registers_of(reinterpret_cast<const void*>(kern)),
                                        WW ? 2 : 1, dma_depth, g.block_threads, g.ring_nt};
#endif
        hipLaunchKernelGGL(kern, grid, dim3(g.block_threads), lds, c.stream, c.grad, c.x, c.dx, g, static_cast<const T*>(c.scale),
                           static_cast<const T*>(c.shift), r, c.gs, c.partials, PcDirect<T>{nullptr, nullptr, nullptr, c.sym_term, 0});
        result = hipGetLastError();
        if (result != hipSuccess) return true;
        const int fin_ch = fin_channels(c.C);
        if (WW) {
            const unsigned fgrid = static_cast<unsigned>((g.n_windows * g.k_slots + fin_ch - 1) / fin_ch);
            hipLaunchKernelGGL((finalize_ww_kernel<T>), dim3(fgrid), dim3(kBlock), 0, c.stream, c.partials, g, fin_ch,
                               p.eval_mode ? 1 : 0, p.sym ? 1 : 0, c.sym_term, c.ds, c.db, c.wide);
        }
        else {
            const unsigned fgrid_w = static_cast<unsigned>((c.C + fin_ch - 1) / fin_ch);
            hipLaunchKernelGGL((finalize_pc_kernel<T>), dim3(fgrid_w), dim3(kBlock), 0, c.stream, c.partials, g, fin_ch,
                               p.eval_mode ? 1 : 0, p.sym ? 1 : 0, c.sym_term, c.ds, c.db, c.wide);
        }
        result = hipGetLastError();
        return true;
    };
    // LDS-DMA ring instead of register buffers (16-byte packets only): the default whenever a workgroup walks at least
    // as many row tiles as the ring is deep -- with the grid the ring likes, kDmaBwdBlocksPerCU workgroups per CU;
    // variant bits 12-13 force either path for A/B runs (1 = registers, 2 = ring).
    constexpr bool kDmaAble = V * sizeof(typename IO::elem) == 16;
#ifndef LSQ_BWD_DMA_DEPTH
#define LSQ_BWD_DMA_DEPTH 4
#endif
    constexpr int kDmaDepth = LSQ_BWD_DMA_DEPTH;
    // OWNER windows first (lsq_pc_geom.hpp, plan_own): activations whose channel rows are short (NCHW with small H x W) and
    // whose tensor is small enough for the finalize launch to matter -- one launch, no workspace.
    if constexpr (kDmaAble && !WW && !EVAL && V > 1 && CPL <= 2) {
        const int own_knob = knob::get(knob::kOwn);     // tools builds: 1 = wherever the shape allows, 2 = never, 3 = 1 without the priority turns
        const int own = own_knob == 3 ? 1 : own_knob;
        const int64_t bytes = c.outer * c.C * c.inner * static_cast<int64_t>(sizeof(typename IO::elem));
        if (c.ring_ok && own != 2 && (own == 1 || (c.default_variant && c.outer * c.C * c.inner <= kOwnMaxElemsOf<static_cast<int>(sizeof(typename IO::elem))>))) {
            const int64_t elems = c.outer * c.C * c.inner;
            auto launch_own = [&](auto block_c) -> bool {
                constexpr int OB = decltype(block_c)::value;
                const int min_run = knob::get(knob::kOwnMinRun);           // tools builds: bytes, 0 = kOwnMinRunBytes
                const int fat = knob::get(knob::kOwnFat);                  // tools builds: 1 = smallest channel group, 2 = fattest
                const OwnPlan op = plan_own(c.outer, c.C, c.inner, V, static_cast<int>(sizeof(typename IO::elem)), kDmaDepth,
                                            device_info().cu_count, OB, min_run > 0 ? min_run : kOwnMinRunBytes,
                                            fat ? fat - 1 : kOwnFatDefault);
                if (op.k == 0) return false;
                if (own != 1 && op.run_bytes < kOwnShortRunBytes && op.run_bytes % 128 != 0 && elems > kOwnMaxElemsShortRun) return false;
                PcGeom g = make_geom_own(c.outer, c.C, c.inner, V, op);
                g.ring_nt = ring_nt_for(bytes, true, false);
                g.own_prio = own_knob == 3 ? 0 : 1;
#if defined(LSQ_TOOLS) && defined(LSQ_TIMELINE)
                g.timeline = knob::timeline_buffer().load();
#endif
                const size_t lds = bwd_lds_front_bytes(g, sizeof(QSlot<T>)) +
                                   static_cast<size_t>(g.block_threads / 64) * kDmaDepth * kDmaStageBytes;
                if (lds > kLdsBytesPerWorkgroup) return false;     // (cannot happen with plan_own's sizing on gfx950: the other families then)
                if (c.plan_need) {                       // no workspace
                    if (c.plan_note)
                        *c.plan_note = LaunchNote{static_cast<int>(g.n_windows), 1, op.per_cu, 0, 4, kDmaDepth, g.block_threads, g.ring_nt};
                    return true;
                }
                constexpr auto kern = bwd_pc_kernel<IO, V, CPL, SYM, INIT, EVAL, 1, true, true, false, false, kDmaDepth, OB>;
#ifdef LSQ_TOOLS
                last_launch_note() = LaunchNote{static_cast<int>(g.n_windows), 1, op.per_cu, registers_of(reinterpret_cast<const void*>(kern)),
                                                4, kDmaDepth, g.block_threads, g.ring_nt};
#endif
                hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(g.n_windows)), dim3(g.block_threads), lds, c.stream, c.grad, c.x,
                                   c.dx, g, static_cast<const T*>(c.scale), static_cast<const T*>(c.shift), r, c.gs, c.partials,
                                   PcDirect<T>{c.ds, c.db, c.wide, c.sym_term, p.sym ? 1 : 0});
                result = hipGetLastError();
                return true;
            };
            if (launch_own(std::integral_constant<int, kOwnBlock>{})) return result;
        }
    }
    if constexpr (kDmaAble) {
        // (4- and 8-byte storage with one channel per lane, CPL == 1, keeps its register loop: it already has eight loads
        // in flight per lane and few registers, the ring only adds its LDS round trip -- measured 3-6 % slower)
        constexpr bool kDefaultHere = kDmaDefault<IO> && (sizeof(typename IO::elem) < 4 || CPL >= 2);
        if (c.ring_ok && (c.v.dma == 2 || (c.v.dma == 0 && kDefaultHere))) {
            const int target = c.default_variant ? device_info().cu_count * kDmaBwdBlocksPerCU<IO> : c.target_blocks;
            // (row-group windows of 4- and 8-byte storage: only tensors up to 160 MB -- [8192,4096] 78 -> 69 us,
            // [64,197,768] 33 -> 28 us; on the bigger ones, four elements a row, the ring's per-row bookkeeping costs more
            // than its loads in flight gain: NHWC [64,56,56,256] 119 -> 127 us, [65536,1024] 157 -> 172 us,
            // profiles/r02_ww_min_rows_sweep.txt)
            const bool big_wide = WW && sizeof(typename IO::elem) >= 4 &&
                                  c.outer * c.C * static_cast<int64_t>(sizeof(typename IO::elem)) > (int64_t{160} << 20);
            // ... and not below 2^24 elements either: register loops are level or ahead there on every row width -- 1 M elements
            // -8 .. -12 % ([1024,1024] 10.1 -> 8.9 us), 2-4 M 0 .. -6 %, 8.4 M -3 .. -11 %, 12.6 M -3 .. -6 %; from 16.8 M on the
            // ring leads ([16384,1024] 51.3 -> 44.7 us).  profiles/r04_rowgroup_ring_small.txt, r04_rowgroup_mid.txt
            const bool small_wide = WW && sizeof(typename IO::elem) >= 4 && c.outer * c.C < (int64_t{1} << 24);
            if constexpr (WW && !EVAL) {
                // Mid-sized tensors whose rows fit one window (8 M .. 80 M elements: [64,197,768], [256,197,768], NHWC
                // [16,56,56,256]): ONE 768/1024-lane workgroup per CU instead of three or four 3-4-wave ones -- the same
                // waves in flight, evenly over the four SIMDs (3-wave workgroups load them 3:2:2:2), a third of the partial
                // rows, constants and epilogues.  6-12 % faster there, slower below (a [16,197,768] wants many short
                // workgroups) and no gain above (profiles/r02_ww_big_ab.txt; upper end, cold buffers:
                // profiles/r03_ww_big_upper_ab.txt -- 16-bit storage -7 % at 48 M elements, -1 .. -4 % at 64 M, +1 .. +5 % at 96 M).
                const int big = knob::get(knob::kWwBig);
                const int64_t elems = c.outer * c.C;
                // (round 2 kept 4- and 8-byte storage up to 64 MB: [256,197,768] fp32 cold 91.5 vs 102.7 us for the usual
                // workgroups, profiles/r02_cold_buffers_pc.txt)
                // Round 4, another box (profiles/r04_rowgroup_mid.txt): 16-bit storage only -- in fp32 the fat workgroup never
                // led (8.4 M: 26.3-29.4 us against 24.1-26.1 for register loops; 12.6 M: level with the usual ring) -- and
                // only for rows of at least 64 lanes: [rows,64] bf16 loses 11-16 % with it (12.6 M elements 33.5 -> 28.0 us),
                // [rows,128] 5-7 %, [rows,256] 2-3 %; from [rows,512] on it is level or ahead up to 67 M elements.  At the lower
                // end, 8.4-11 M elements, it is -1 .. -8 % on nine shapes of twelve and +9 / +14 % on two with power-of-two row
                // counts ([8192,1024], [4096,2048]), with a third of the partial rows (profiles/r04_ww_big_low_end.txt): the
                // lower end stays at 2^23.
                // Round 5, after the epilogue's combine went over all lanes (the fat workgroup's own cost): BELOW 2^23 elements it
                // now leads wherever a row is at most 96 lanes -- every shape of [rows, 128 .. 768] bf16 from 3 M elements on
                // (+4 .. +19 %: [8192,384] 14.2 -> 12.5 us, [10000,768] 20.6 -> 16.8, [12608,256] +8 %), and from 0.8 M on where
                // the row does not tile a 256-lane workgroup (48, 80, 96 lanes: [2048,384] +8 %, [4096,640] +12 %, [3152,768]
                // +14 %; rows of 16 / 32 / 64 lanes are -10 .. +2 % there and keep their four-wave workgroups); rows of 128+ lanes
                // stay as they were ([4096,1024], [2048,2048] -5 %).  One row tile per workgroup is enough down there.
                // profiles/r05_ww_big_small_tensors.txt
                const bool fits = sizeof(typename IO::elem) < 4;
                const int64_t w_lanes = c.C / V;
                const bool low = elems < (int64_t{1} << 23);
                // (floors: what the round-5 sweep covered -- 16 lanes below 2^23 elements, 8 lanes inside the band; narrower rows
                //  mean hundreds of row groups per workgroup and a ~100 KB combine buffer nobody measured: they keep the usual
                //  four-wave workgroups)
                const bool low_ok = w_lanes >= 16 && w_lanes <= 96 && (elems >= (int64_t{3} << 20) ||
                                                      (elems >= (int64_t{3} << 18) && kBlock % static_cast<int>(w_lanes) != 0));
                // ... and INSIDE the band the narrow rows changed sides with it: [rows, 64 / 128 / 256] +4 .. +9 % at 9-17 M
                // elements (round 4: -11 .. -16 % for [rows,64]), -3 % at 25 M, level above; [rows,384] +1 .. +13 % through the
                // whole band (same file, second table)
                const bool band_ok = w_lanes >= 8 && elems < (int64_t{5} << 24) &&
                                     (w_lanes >= 64 || kBlock % static_cast<int>(w_lanes) != 0 || elems < (int64_t{5} << 22));
                const bool use_big = big == 1 || (big == 0 && c.default_variant && w_lanes <= kBlock && fits && (low ? low_ok : band_ok));
                constexpr int kBigBlock = kBigBlockOf<sizeof(typename IO::elem)>;
                if (use_big &&
                    run(bwd_pc_kernel<IO, V, CPL, SYM, INIT, EVAL, 1, true, true, false, WW, kDmaDepth, kBigBlock>, kDmaDepth,
                        device_info().cu_count, (big == 1 || low) ? 0 : 2, INT64_MAX, kBigBlock))
                    return result;
            }
            // The tiles-per-workgroup floor of the ring (as many as it is deep) does not hold for 16-bit row groups: there the
            // ring is ahead with ONE tile per workgroup too -- [1568,512] bf16 13.1 -> 11.0 us, [16384,128] 21.0 -> 15.8,
            // [1365,384] 11.9 -> 10.2, nothing behind by more than 2.5 % (profiles/r04_rowgroup_ring_small.txt)
            const int64_t floor_tiles = (WW && sizeof(typename IO::elem) < 4) ? 1 : kDmaDepth;
            if (!((big_wide || small_wide) && c.v.dma != 2) &&
                run(bwd_pc_kernel<IO, V, CPL, SYM, INIT, EVAL, 1, true, true, false, WW, kDmaDepth>, kDmaDepth, target,
                    c.v.dma == 2 ? 0 : floor_tiles))
                return result;
        }
    }
#define LSQ_LAUNCH_P(U, NTLF, NTSF, PIPEF) \
    run(bwd_pc_kernel<IO, V, CPL, SYM, INIT, EVAL, U, NTLF, NTSF, PIPEF, WW>, 0, c.target_blocks, 0)
#ifdef LSQ_TUNING
    // tuning builds compile both loops for the swept kernels; the switch is the variant's `chunked` bit (unused here)
    const bool pipe = kFull ? c.v.chunked : kNarrow;
#define LSQ_LAUNCH(U, NTLF, NTSF)                                   \
    do {                                                            \
        if constexpr (kFull) {                                      \
            if (pipe) LSQ_LAUNCH_P(U, NTLF, NTSF, true);            \
            else LSQ_LAUNCH_P(U, NTLF, NTSF, false);                \
        } else {                                                    \
            LSQ_LAUNCH_P(U, NTLF, NTSF, kNarrow);                   \
        }                                                           \
    } while (0)
#else
#define LSQ_LAUNCH(U, NTLF, NTSF) LSQ_LAUNCH_P(U, NTLF, NTSF, kNarrow)
#endif
    LSQ_DISPATCH_VARIANT(kFull, kDefU, c.v, LSQ_LAUNCH);
#undef LSQ_LAUNCH
#undef LSQ_LAUNCH_P
    return result;
}

template <typename IO, bool SYM, bool INIT, bool EVAL>
static hipError_t launch_bwd_seg(const void* grad, const void* x, void* dx, const SegGeom& g, const void* scale,
                                 const void* shift, const lsq_params& p, typename IO::arith gs, double2* partials,
                                 const SegDirect<typename IO::arith>& direct, const Variant& v, hipStream_t stream) {
    using T = typename IO::arith;
    const Range<T> r = make_range<T>(p);
    const dim3 grid(static_cast<unsigned>(g.C * g.segs), static_cast<unsigned>(g.osplits));
    const bool short_walk = g.sub_per_seg * g.o_per_split <= kSegUpFrontBwd<IO> && knob::get(knob::kSegNoUpFront) == 0 &&
                            (g.o_per_split == 1 || g.C * g.segs * g.osplits <= 8 * static_cast<int64_t>(device_info().cu_count));
#define LSQ_LAUNCH(U, NTLF, NTSF)                                                                                                 \
    do {                                                                                                                          \
        if (short_walk)                                                                                                           \
            hipLaunchKernelGGL((bwd_seg_kernel<IO, IO::VEC, SYM, INIT, EVAL, U, NTLF, NTSF, 1>), grid, dim3(kBlock), 0, stream, grad, \
                               x, dx, g, static_cast<const T*>(scale), static_cast<const T*>(shift), r, gs, partials, direct);  \
        else                                                                                                                      \
            hipLaunchKernelGGL((bwd_seg_kernel<IO, IO::VEC, SYM, INIT, EVAL, U, NTLF, NTSF, 2>), grid, dim3(kBlock), 0, stream, grad, \
                               x, dx, g, static_cast<const T*>(scale), static_cast<const T*>(shift), r, gs, partials, direct);  \
    } while (0)
    [[maybe_unused]] constexpr bool kFull = !INIT && !EVAL && (std::is_same<IO, io_f32>::value || std::is_same<IO, io_bf16>::value);
    LSQ_DISPATCH_VARIANT(kFull, kSegUnroll<IO>, v, LSQ_LAUNCH);
#undef LSQ_LAUNCH
    return hipGetLastError();
}

#define LSQ_MODE_SWITCH(CALL)                                  \
    do {                                                       \
        const bool sym = p.sym != 0, init = p.init_mode != 0;  \
        if (p.eval_mode) {                                     \
            if (init) return CALL(false, true, true);          \
            return CALL(false, false, true);                   \
        }                                                      \
        if (sym) {                                             \
            if (init) return CALL(true, true, false);          \
            return CALL(true, false, false);                   \
        }                                                      \
        if (init) return CALL(false, true, false);             \
        return CALL(false, false, false);                      \
    } while (0)

template <typename IO, int V, int CPL, bool WW = false>
static hipError_t bwd_pc_modes(const BwdPcCall<typename IO::arith>& c) {
    const lsq_params& p = *c.p;
#define LSQ_CASE(S, I, E) launch_bwd_pc<IO, V, CPL, S, I, E, WW>(c)
    LSQ_MODE_SWITCH(LSQ_CASE);
#undef LSQ_CASE
}

template <typename IO>
static hipError_t bwd_seg_modes(const void* grad, const void* x, void* dx, const SegGeom& g, const void* scale,
                                const void* shift, const lsq_params& p, typename IO::arith gs, double2* partials,
                                const SegDirect<typename IO::arith>& direct, const Variant& v, hipStream_t stream) {
#define LSQ_CASE(S, I, E) launch_bwd_seg<IO, S, I, E>(grad, x, dx, g, scale, shift, p, gs, partials, direct, v, stream)
    LSQ_MODE_SWITCH(LSQ_CASE);
#undef LSQ_CASE
}

// Last-axis tensors under 512 MB take row-group windows in the backward; from there on the 256-lane windows, which read
// 4 KiB contiguous per row and workgroup instead of 1 KiB from each of four rows, are level or ahead
// (profiles/r03_ww_max_ab.txt, cold buffers: bf16 row groups -2 .. -12 % at 256 MB, -4 .. +7 % at 512 MB, level at 1 GB;
// fp32 -3 .. -9 % at 512 MB for rows up to 2048 features, +6 .. +22 % for wider rows -- those decide the 4-byte bound).
template <typename IO>
inline int64_t ww_max_elems() {
    const int k = knob::get(knob::kWwMaxLog2);      // tools builds: lsq_hip_debug_set_ww_max_log2
    return k > 0 ? int64_t{1} << k : (int64_t{512} << 20) / static_cast<int64_t>(sizeof(typename IO::elem));
}

template <typename IO>
hipError_t backward_per_channel(const void* grad, const void* x, void* dx, void* ds, void* db, double* wide,
                                int64_t outer, int64_t channels, int64_t inner, const void* scale,
                                const void* shift, const lsq_params& p, void* workspace, size_t workspace_bytes,
                                uint32_t* ticket, int variant, hipStream_t stream, size_t* plan_need, LaunchNote* plan_note) {
    using T = typename IO::arith;
    (void)ticket;
    const DeviceInfo& dev = device_info();
    // (forward_per_channel: packets on any element-aligned view; the ring and the owner windows built on it want 16 bytes)
    const bool aligned = is_elem_aligned<IO>(grad) && is_elem_aligned<IO>(x) && is_elem_aligned<IO>(dx);
    const bool ring_ok = is_aligned16(grad) && is_aligned16(x) && is_aligned16(dx);
    const int vec = pick_vec(IO::VEC, channels * inner, aligned);
    const bool seg = pick_segment_mode(vec, outer, channels, inner, dev.cu_count);
    const Variant v = decode_variant(variant, seg ? (sizeof(typename IO::elem) >= 4 ? kDefaultPcSegVariant : kDefaultPcSegNarrowVariant)
                                                  : (sizeof(typename IO::elem) >= 4 ? kDefaultPcBwdWideVariant
                                                                                    : kDefaultPcBwdNarrowVariant));
    const int target = dev.cu_count * v.blocks_per_cu;
    const int64_t numel = outer * channels * inner;
    const int64_t n4s = p.numel_for_scaler > 0 ? p.numel_for_scaler : numel;
    const T gs = grad_scaler_per_channel<T>(n4s, p.quant_max, channels, p.use_grad_scaling != 0, p.grad_scaler);
    const T sym_term = static_cast<T>(0) * gs;
    double2* partials = static_cast<double2*>(workspace);
    const int fin_ch = fin_channels(channels);
    const unsigned fgrid_w = static_cast<unsigned>((channels + fin_ch - 1) / fin_ch);

    if (seg) {
        const SegGeom sg = make_seg_geom(outer, channels, inner, vec, target);
        if (!grid_fits(sg)) return hipErrorInvalidConfiguration;
        const size_t need = static_cast<size_t>(channels) * sg.segs * sg.osplits * sizeof(double2);
        if (plan_need) {
            if (!p.eval_mode) *plan_need = std::max(*plan_need, need);
            if (plan_note) *plan_note = LaunchNote{static_cast<int>(sg.C * sg.segs), sg.osplits, 0, 0, 3, 0, kBlock, 0};
            return hipSuccess;
        }
        if (!p.eval_mode && workspace_bytes < need) return hipErrorInvalidValue;
        const bool one_partial = sg.segs == 1 && sg.osplits == 1;
#ifdef LSQ_TOOLS
        last_launch_note() = LaunchNote{static_cast<int>(sg.C * sg.segs), sg.osplits, 0, 0, 3, 0, kBlock, 0};
#endif
        const SegDirect<T> direct{one_partial ? static_cast<T*>(ds) : nullptr, static_cast<T*>(db), wide, sym_term};
        hipError_t e = bwd_seg_modes<IO>(grad, x, dx, sg, scale, shift, p, gs, partials, direct, v, stream);
        if (e != hipSuccess || one_partial) return e;
        hipLaunchKernelGGL((finalize_seg_kernel<T>), dim3(fgrid_w), dim3(kBlock), 0, stream, partials, sg, fin_ch,
                           p.eval_mode ? 1 : 0, p.sym ? 1 : 0, sym_term, static_cast<T*>(ds), static_cast<T*>(db), wide);
        return hipGetLastError();
    }

    constexpr int VB = kWindowBwdVec<IO>;
    const int vecw = pick_vec(VB, channels * inner, aligned);
    const int cpl = pick_cpl(vecw, inner);
    // One channel per packet component (inner < V): a window spans 256 x V channels, so every workgroup ends with a
    // long epilogue and a 16-byte partial per slot, and the finalize has `splits` of them to fold per channel.  Fewer,
    // fatter workgroups win there in every shape swept (profiles/r01_lastaxis_sweep.txt: 2 per CU; [8192, 4096] fp32
    // 82 us against 100 us at 16 per CU, [200704, 256] 133 against 205).
    const bool last_axis = vecw > 1 && cpl == vecw;
    // (under 512 MB only, ww_max_elems; variant bit 11 (tools) forces the 256-lane windows, for A/B runs)
    if (last_axis && inner == 1 && !(variant & (1 << 11)) && (variant != 0 || outer * channels < ww_max_elems<IO>())) {
        // the quantized axis is the last one ([tokens, features], channels-last): row-group windows, one round of what
        // the chip holds (variant: workgroups per CU requested, rounded to whole rounds)
        BwdPcCall<T> call{grad, x, dx, static_cast<T*>(ds), static_cast<T*>(db), wide, outer, channels, inner, scale, shift, &p,
                          gs, sym_term, partials, workspace_bytes, variant == 0 ? dev.cu_count * kWwBwdBlocksPerCU : target,
                          /*default_variant=*/variant == 0, /*whole_rounds=*/true, ring_ok, v, stream, plan_need, plan_note};
        return bwd_pc_modes<IO, VB, VB, true>(call);
    }
    const int target_w = (variant == 0 && last_axis) ? dev.cu_count * kLastAxisBwdBlocksPerCU : target;
    BwdPcCall<T> call{grad, x, dx, static_cast<T*>(ds), static_cast<T*>(db), wide, outer, channels, inner, scale, shift, &p,
                      gs, sym_term, partials, workspace_bytes, target_w, /*default_variant=*/variant == 0,
                      /*whole_rounds=*/!last_axis, ring_ok, v, stream, plan_need, plan_note};
    if (vecw == 1) return bwd_pc_modes<IO, 1, 1>(call);
    if (cpl == 1) return bwd_pc_modes<IO, VB, 1>(call);
    if (cpl == 2) return bwd_pc_modes<IO, VB, 2>(call);
    return bwd_pc_modes<IO, VB, VB>(call);
}

#define LSQ_INSTANTIATE(IO)                                                                                          \
    template hipError_t forward_per_channel<IO>(const void*, void*, int64_t, int64_t, int64_t, const void*,          \
                                                const void*, const lsq_params&, const lsq_fwd_extras*, int,          \
                                                hipStream_t);                                                        \
    template hipError_t backward_per_channel<IO>(const void*, const void*, void*, void*, void*, double*, int64_t,    \
                                                 int64_t, int64_t, const void*, const void*, const lsq_params&,      \
                                                 void*, size_t, uint32_t*, int, hipStream_t, size_t*, LaunchNote*); \
    template size_t bwd_pc_workspace_bytes<IO>(int64_t, int64_t, int64_t);
// One translation unit per storage type (the Makefile compiles this file four times with -DLSQ_PC_IO=io_f32 ... in
// parallel: the window kernels' template space takes minutes in one piece); without the macro, all four.
#ifdef LSQ_PC_IO
LSQ_INSTANTIATE(LSQ_PC_IO)
#else
LSQ_INSTANTIATE(io_f32)
LSQ_INSTANTIATE(io_f64)
LSQ_INSTANTIATE(io_bf16)
LSQ_INSTANTIATE(io_f16)
#endif
#undef LSQ_INSTANTIATE

}  // namespace lsq
