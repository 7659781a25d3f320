// lsq_per_channel.hip -- K3 (forward) and K4 (fused backward + per-channel reduction) on gfx950.
//
// Replaces the reference's per-channel CUDA backend (/root/reference/torchlsq/csrc/ops/cuda/lsq_cuda.cu:147-297):
// TensorIterator-broadcast scale/shift with a division per element (lsq_kernel.h:157-158), three
// elementwise backward kernels, three N-sized temporaries and two `sum(axes != axis)`.
//
// Data view: dense memory as [outer][L], L = C*inner; position p in a row belongs to channel p/inner.
//
// CDNA4 design: "channel-stationary lanes".
//  * A workgroup owns a WINDOW of positions [w*W, (w+1)*W) of the row (W = 256 lanes x V elements,
//    V = one 16-byte packet) and walks down a slab of rows o = o0, o0+R, ... .  A lane keeps the same
//    positions -- hence the same channel(s) -- for its whole life: the per-channel constants
//    {s, 1/s, zp} live in registers, no per-element index arithmetic or division is left in the
//    loop, and every wave instruction still moves 1 KiB of contiguous HBM.
//    Short rows (L < W, e.g. [batch, features] activations) fold R = W/L rows into one tile.
//  * The window's channel table {s, 1/s, zp} is computed ONCE per workgroup into LDS (one IEEE
//    division per channel per workgroup instead of one per element) and fanned out to the lanes.
//  * d_scale / d_shift: fp64 lane accumulators (one per channel the lane touches) ->
//    segmented wave64 shuffle reduction keyed by channel (lanes of a wave hold runs of equal
//    channels) -> LDS fp64 atomics on the window's channel slots (ds_add_f64) -> one 16-byte partial
//    per (workgroup, channel slot) in the workspace -> fixed-order finalize per channel.
//    No global atomics, no zero-initialised buffers.
#include "lsq_kernels.hpp"

namespace lsq {

// Launch geometry, computed on the host and passed by value.
struct PcGeom {
    int64_t outer, C, inner, L;
    int64_t wpos;            // positions per window (R == 1) or L (R > 1)
    int64_t n_windows;       // windows per row
    int64_t rows_per_split;  // rows walked by one workgroup (multiple of R)
    int32_t splits;          // workgroups along the row axis
    int32_t R;               // rows folded into one tile
    int32_t k_slots;         // channel slots per window (LDS table / partial row length)
    int32_t vec;             // elements per lane per row (IO::VEC or 1)
};

static PcGeom make_geom(int64_t outer, int64_t C, int64_t inner, int vec, int target_blocks) {
    PcGeom g;
    g.outer = outer; g.C = C; g.inner = inner; g.L = C * inner; g.vec = vec;
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
    // keep the partial-sum traffic (16 B per slot per workgroup) below ~5 % of the streamed bytes
    const int64_t min_rows = std::max<int64_t>(g.R, (27 * static_cast<int64_t>(g.k_slots) + W - 1) / W * g.R);
    int64_t want_splits = std::max<int64_t>(1, (target_blocks + g.n_windows - 1) / g.n_windows);
    int64_t rows = (outer + want_splits - 1) / want_splits;
    rows = std::max<int64_t>(rows, min_rows);
    rows = (rows + g.R - 1) / g.R * g.R;
    g.rows_per_split = rows;
    g.splits = static_cast<int32_t>((outer + rows - 1) / rows);
    return g;
}

template <typename T>
struct alignas(16) QSlot {  // LDS image of one channel's constants
    T s, inv_s, zp, pad;
};

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

// Where a lane sits: position p0 of its first element, its row inside the tile, and whether it is live.
struct LaneSite {
    int64_t p0;
    int32_t row_in_tile;
    bool live;
    int64_t c_lo;  // first channel of the window
};
__device__ __forceinline__ LaneSite lane_site(const PcGeom& g, int V) {
    LaneSite s;
    const int64_t idx = static_cast<int64_t>(threadIdx.x) * V;
    if (g.R == 1) {
        const int64_t base = static_cast<int64_t>(blockIdx.x) * g.wpos;
        s.p0 = base + idx;
        s.row_in_tile = 0;
        s.live = s.p0 < g.L;
        s.c_lo = base / g.inner;
    } else {
        s.row_in_tile = static_cast<int32_t>(idx / g.L);
        s.p0 = idx - static_cast<int64_t>(s.row_in_tile) * g.L;
        s.live = s.row_in_tile < g.R;
        s.c_lo = 0;
    }
    return s;
}

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
        const int64_t p0 = s.live ? s.p0 : s.c_lo * g.inner;
        const int64_t c0 = p0 / g.inner;
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
            else k = s.live ? static_cast<int32_t>((p0 + j) / g.inner - s.c_lo) : 0;
            key[j] = k;
            const QSlot<T> e = table[k];
            q[j].s = e.s; q[j].inv_s = e.inv_s; q[j].zp = e.zp;
        }
    }
    // 0/1/.. = which of the lane's channels component j belongs to (compile-time for CPL != 2)
    __device__ __forceinline__ int which(int j) const {
        if (N == 1) return 0;
        if (CPL == 2) return j >= split ? 1 : 0;
        return j;
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
};

// ------------------------------------------------------------------------------------------------
// K3: forward
// ------------------------------------------------------------------------------------------------
template <typename IO, int V, int CPL, bool INIT, bool LEVELS, int UNROLL, bool NTL, bool NTS>
__global__ __launch_bounds__(kBlock) void fwd_pc_kernel(const void* __restrict__ x, void* __restrict__ y,
                                                        int8_t* __restrict__ levels, int level_bias, PcGeom g,
                                                        const typename IO::arith* __restrict__ scale,
                                                        const typename IO::arith* __restrict__ shift,
                                                        Range<typename IO::arith> r) {
    using T = typename IO::arith;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    QSlot<T>* table = reinterpret_cast<QSlot<T>*>(smem);

    const LaneSite site = lane_site(g, V);
    build_channel_table<T>(table, g.k_slots, site.c_lo, g.C, scale, shift, r);
    __syncthreads();
    LaneChannels<T, V, CPL> ch;
    ch.init(table, site, g);
    if (!site.live) return;
    const T bias = static_cast<T>(level_bias);

    const int64_t o_begin = static_cast<int64_t>(blockIdx.y) * g.rows_per_split + site.row_in_tile;
    const int64_t o_end = std::min<int64_t>(g.outer, static_cast<int64_t>(blockIdx.y + 1) * g.rows_per_split);
    const int64_t step = g.R;

    auto load_row = [&](int64_t oo, typename IO::elem (&in)[V]) {
        const int64_t e = oo * g.L + site.p0;
        if constexpr (V == 1) {
            in[0] = static_cast<const typename IO::elem*>(x)[e];
        } else {
            const Packet<IO> pk = NTL ? load_packet_nt<IO>(x, e) : load_packet<IO>(x, e);
#pragma unroll
            for (int j = 0; j < V; ++j) in[j] = pk.v[j];
        }
    };
    auto emit_row = [&](int64_t oo, const typename IO::elem (&in)[V]) {
        const int64_t e = oo * g.L + site.p0;
        typename IO::elem out[V];
        LevelPack<V> lv;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const QParams<T> q = ch.params(j);
            const T xv = static_cast<T>(in[j]);
            const T l = level<T>(xv, q, r);
            out[j] = static_cast<typename IO::elem>(INIT ? xv : dequant<T>(l, q));
            if (LEVELS) lv.b[j] = static_cast<int8_t>(static_cast<int>(l - bias));
        }
        if constexpr (V == 1) {
            static_cast<typename IO::elem*>(y)[e] = out[0];
        } else {
            Packet<IO> pk;
#pragma unroll
            for (int j = 0; j < V; ++j) pk.v[j] = out[j];
            if (NTS) store_packet_nt<IO>(y, e, pk); else store_packet<IO>(y, e, pk);
        }
        if (LEVELS) lv.store(levels + e);
    };

    int64_t o = o_begin;
    // full groups of UNROLL rows: unpredicated, all loads issued before the first use
    for (; o + step * (UNROLL - 1) < o_end; o += step * UNROLL) {
        typename IO::elem in[UNROLL][V];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) load_row(o + u * step, in[u]);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) emit_row(o + u * step, in[u]);
    }
    for (; o < o_end; o += step) {
        typename IO::elem in[V];
        load_row(o, in);
        emit_row(o, in);
    }
}

// ------------------------------------------------------------------------------------------------
// K4: backward
// ------------------------------------------------------------------------------------------------
// Segmented wave64 reduction: lanes hold (key, s, b).  A RUN is a maximal group of ADJACENT lanes
// with the same key (equal keys may re-appear further away -- folded rows, inner < V -- so runs are
// numbered with a ballot + popcount and the scan is keyed by run id, not by channel).  After
// log2(64) shuffle steps the first lane of every run owns the run total and adds it to the
// window's LDS slot with an LDS fp64 atomic (ds_add_f64).
template <bool SYM>
__device__ __forceinline__ void segmented_wave_accumulate(int key, double s, double b, double* lds_s, double* lds_b) {
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
    if (head && key >= 0) {
        __hip_atomic_fetch_add(&lds_s[key], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (!SYM) __hip_atomic_fetch_add(&lds_b[key], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

template <typename IO, int V, int CPL, bool SYM, bool INIT, bool EVAL, int UNROLL, bool NTL, bool NTS>
__global__ __launch_bounds__(kBlock) void bwd_pc_kernel(const void* __restrict__ grad, const void* __restrict__ x,
                                                        void* __restrict__ dx, PcGeom g,
                                                        const typename IO::arith* __restrict__ scale,
                                                        const typename IO::arith* __restrict__ shift,
                                                        Range<typename IO::arith> r, typename IO::arith grad_scaler,
                                                        double2* __restrict__ partials) {
    using T = typename IO::arith;
    using LC = LaneChannels<T, V, CPL>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    QSlot<T>* table = reinterpret_cast<QSlot<T>*>(smem);
    double* lds_s = reinterpret_cast<double*>(smem + static_cast<size_t>(g.k_slots) * sizeof(QSlot<T>));
    double* lds_b = lds_s + g.k_slots;

    const LaneSite site = lane_site(g, V);
    build_channel_table<T>(table, g.k_slots, site.c_lo, g.C, scale, shift, r);
    if (!EVAL) {
        for (int k = threadIdx.x; k < g.k_slots; k += kBlock) {
            lds_s[k] = 0.0;
            lds_b[k] = 0.0;
        }
    }
    __syncthreads();
    LC ch;
    ch.init(table, site, g);

    double acc_s[LC::N], acc_b[LC::N];
#pragma unroll
    for (int j = 0; j < LC::N; ++j) { acc_s[j] = 0.0; acc_b[j] = 0.0; }

    if (site.live) {
        const int64_t o_begin = static_cast<int64_t>(blockIdx.y) * g.rows_per_split + site.row_in_tile;
        const int64_t o_end = std::min<int64_t>(g.outer, static_cast<int64_t>(blockIdx.y + 1) * g.rows_per_split);
        const int64_t step = g.R;
        auto load_row = [&](int64_t oo, typename IO::elem (&gi)[V], typename IO::elem (&xi)[V]) {
            const int64_t e = oo * g.L + site.p0;
            if constexpr (V == 1) {
                gi[0] = static_cast<const typename IO::elem*>(grad)[e];
                xi[0] = static_cast<const typename IO::elem*>(x)[e];
            } else {
                const Packet<IO> pg = NTL ? load_packet_nt<IO>(grad, e) : load_packet<IO>(grad, e);
                const Packet<IO> px = NTL ? load_packet_nt<IO>(x, e) : load_packet<IO>(x, e);
#pragma unroll
                for (int j = 0; j < V; ++j) { gi[j] = pg.v[j]; xi[j] = px.v[j]; }
            }
        };
        auto emit_row = [&](int64_t oo, const typename IO::elem (&gi)[V], const typename IO::elem (&xi)[V]) {
            const int64_t e = oo * g.L + site.p0;
            typename IO::elem out[V];
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const QParams<T> q = ch.params(j);
                const T gv = static_cast<T>(gi[j]), xv = static_cast<T>(xi[j]);
                if (EVAL) {
                    out[j] = static_cast<typename IO::elem>(backward_elem_eval<T, INIT>(gv, xv, q, r));
                } else {
                    T ds_t, db_t;
                    out[j] = static_cast<typename IO::elem>(
                        backward_elem<T, SYM, INIT>(gv, xv, q, r, grad_scaler, ds_t, db_t));
                    const double a = static_cast<double>(ds_t), c = static_cast<double>(db_t);
                    if (LC::N == 1) {
                        acc_s[0] += a;
                        if (!SYM) acc_b[0] += c;
                    } else if (CPL == 2) {
                        // branch-free routing between the lane's two channels
                        const bool hi = ch.which(j) != 0;
                        acc_s[0] += hi ? 0.0 : a;
                        acc_s[LC::N - 1] += hi ? a : 0.0;
                        if (!SYM) {
                            acc_b[0] += hi ? 0.0 : c;
                            acc_b[LC::N - 1] += hi ? c : 0.0;
                        }
                    } else {
                        acc_s[j < LC::N ? j : 0] += a;
                        if (!SYM) acc_b[j < LC::N ? j : 0] += c;
                    }
                }
            }
            if constexpr (V == 1) {
                static_cast<typename IO::elem*>(dx)[e] = out[0];
            } else {
                Packet<IO> pk;
#pragma unroll
                for (int j = 0; j < V; ++j) pk.v[j] = out[j];
                if (NTS) store_packet_nt<IO>(dx, e, pk); else store_packet<IO>(dx, e, pk);
            }
        };

        int64_t o = o_begin;
        for (; o + step * (UNROLL - 1) < o_end; o += step * UNROLL) {
            typename IO::elem gi[UNROLL][V], xi[UNROLL][V];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) load_row(o + u * step, gi[u], xi[u]);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) emit_row(o + u * step, gi[u], xi[u]);
        }
        for (; o < o_end; o += step) {
            typename IO::elem gi[V], xi[V];
            load_row(o, gi, xi);
            emit_row(o, gi, xi);
        }
    }
    if (EVAL) return;

    // lanes -> window slots.  Dead lanes carry key -1 (never written).
#pragma unroll
    for (int j = 0; j < LC::N; ++j) {
        int key = site.live ? ch.key[j] : -1;
        if (CPL == 2 && j == 1 && ch.split >= V) key = -1;  // lane touches one channel only
        segmented_wave_accumulate<SYM>(key, acc_s[j], acc_b[j], lds_s, lds_b);
    }
    __syncthreads();
    const int64_t block_linear = static_cast<int64_t>(blockIdx.y) * g.n_windows + blockIdx.x;
    double2* out = partials + block_linear * g.k_slots;
    for (int k = threadIdx.x; k < g.k_slots; k += kBlock) out[k] = make_double2(lds_s[k], lds_b[k]);
}

// Finalize: one lane per channel folds, in a fixed order, every (split, window) partial that can
// hold a piece of that channel (the reference's `ds_buffer.sum(axes != axis)`, lsq_cpu.cpp:287-292).
template <typename T>
__global__ __launch_bounds__(kBlock) void finalize_pc_kernel(const double2* __restrict__ partials, PcGeom g,
                                                             int eval_mode, int sym, T sym_term, T* __restrict__ ds,
                                                             T* __restrict__ db, double* __restrict__ wide) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (c >= g.C) return;
    double s = 0.0, b = 0.0;
    if (!eval_mode) {
        int64_t w_lo = 0, w_hi = 0;
        if (g.R == 1) {
            w_lo = (c * g.inner) / g.wpos;
            w_hi = ((c + 1) * g.inner - 1) / g.wpos;
        }
        for (int32_t sy = 0; sy < g.splits; ++sy) {
            for (int64_t w = w_lo; w <= w_hi; ++w) {
                const int64_t c_lo = (g.R == 1) ? (w * g.wpos) / g.inner : 0;
                const double2 v = partials[(static_cast<int64_t>(sy) * g.n_windows + w) * g.k_slots + (c - c_lo)];
                s += v.x;
                b += v.y;
            }
        }
        if (sym) b = 0.0 + static_cast<double>(sym_term);
    }
    ds[c] = static_cast<T>(s);
    db[c] = static_cast<T>(b);
    if (wide) {
        wide[c] = s;
        wide[g.C + c] = b;
    }
}

// ------------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------------
static inline int pick_vec(int io_vec, int64_t L, bool aligned) { return (aligned && (L % io_vec) == 0) ? io_vec : 1; }
static inline int pick_cpl(int vec, int64_t inner) {
    if (vec == 1 || inner % vec == 0) return 1;
    return inner >= vec ? 2 : vec;
}

size_t bwd_pc_workspace_bytes(int io_vec, int64_t outer, int64_t channels, int64_t inner) {
    const DeviceInfo& dev = device_info();
    size_t need = 0;
    const int vecs[2] = {io_vec, 1};
    for (int vi = 0; vi < 2; ++vi) {
        for (int bpc = 1; bpc <= kMaxBlocksPerCU; bpc <<= 1) {
            const PcGeom g = make_geom(outer, channels, inner, vecs[vi], dev.cu_count * bpc);
            need = std::max(need, static_cast<size_t>(g.splits) * g.n_windows * g.k_slots * sizeof(double2));
        }
    }
    return need + 256;
}

template <typename IO, int V, int CPL, bool INIT, bool LEVELS>
static hipError_t launch_fwd_pc(const void* x, void* y, int8_t* levels, int bias, const PcGeom& g, const void* scale,
                                const void* shift, const lsq_params& p, const Variant& v, hipStream_t stream) {
    using T = typename IO::arith;
    const Range<T> r = make_range<T>(p);
    const dim3 grid(static_cast<unsigned>(g.n_windows), static_cast<unsigned>(g.splits));
    const size_t lds = static_cast<size_t>(g.k_slots) * sizeof(QSlot<T>);
#define LSQ_LAUNCH(U, NTLF, NTSF)                                                                                       \
    hipLaunchKernelGGL((fwd_pc_kernel<IO, V, CPL, INIT, LEVELS, U, NTLF, NTSF>), grid, dim3(kBlock), lds, stream, x, y, levels, \
                       bias, g, static_cast<const T*>(scale), static_cast<const T*>(shift), r)
    LSQ_DISPATCH_VARIANT(false, v, LSQ_LAUNCH);
#undef LSQ_LAUNCH
    return hipGetLastError();
}

template <typename IO, int V, int CPL>
static hipError_t fwd_pc_modes(const void* x, void* y, int8_t* levels, int bias, const PcGeom& g, const void* scale,
                               const void* shift, const lsq_params& p, const Variant& v, hipStream_t stream) {
    if (p.init_mode) {
        return levels ? launch_fwd_pc<IO, V, CPL, true, true>(x, y, levels, bias, g, scale, shift, p, v, stream)
                      : launch_fwd_pc<IO, V, CPL, true, false>(x, y, levels, bias, g, scale, shift, p, v, stream);
    }
    return levels ? launch_fwd_pc<IO, V, CPL, false, true>(x, y, levels, bias, g, scale, shift, p, v, stream)
                  : launch_fwd_pc<IO, V, CPL, false, false>(x, y, levels, bias, g, scale, shift, p, v, stream);
}

static inline bool grid_fits(const PcGeom& g) { return g.n_windows <= 0x7fffffffLL && g.splits <= 65535; }

template <typename IO>
hipError_t forward_per_channel(const void* x, void* y, int64_t outer, int64_t channels, int64_t inner,
                               const void* scale, const void* shift, const lsq_params& p,
                               const lsq_fwd_extras* ex, int variant, hipStream_t stream) {
    int8_t* levels = ex ? static_cast<int8_t*>(ex->levels) : nullptr;
    const int bias = ex ? ex->level_bias : 0;
    const Variant v = decode_variant(variant, kDefaultPcVariant);
    const DeviceInfo& dev = device_info();
    const bool aligned = is_aligned16(x) && is_aligned16(y) && (!levels || (reinterpret_cast<uintptr_t>(levels) & 7u) == 0);
    const int vec = pick_vec(IO::VEC, channels * inner, aligned);
    const int cpl = pick_cpl(vec, inner);
    const PcGeom g = make_geom(outer, channels, inner, vec, dev.cu_count * v.blocks_per_cu);
    if (!grid_fits(g)) return hipErrorInvalidConfiguration;
    if (vec == 1) return fwd_pc_modes<IO, 1, 1>(x, y, levels, bias, g, scale, shift, p, v, stream);
    if (cpl == 1) return fwd_pc_modes<IO, IO::VEC, 1>(x, y, levels, bias, g, scale, shift, p, v, stream);
    if (cpl == 2) return fwd_pc_modes<IO, IO::VEC, 2>(x, y, levels, bias, g, scale, shift, p, v, stream);
    return fwd_pc_modes<IO, IO::VEC, IO::VEC>(x, y, levels, bias, g, scale, shift, p, v, stream);
}

template <typename IO, int V, int CPL, bool SYM, bool INIT, bool EVAL>
static hipError_t launch_bwd_pc(const void* grad, const void* x, void* dx, const PcGeom& g, const void* scale,
                                const void* shift, const lsq_params& p, typename IO::arith gs, double2* partials,
                                const Variant& v, hipStream_t stream) {
    using T = typename IO::arith;
    const Range<T> r = make_range<T>(p);
    const dim3 grid(static_cast<unsigned>(g.n_windows), static_cast<unsigned>(g.splits));
    const size_t lds = static_cast<size_t>(g.k_slots) * (sizeof(QSlot<T>) + 2 * sizeof(double));
#define LSQ_LAUNCH(U, NTLF, NTSF)                                                                                          \
    hipLaunchKernelGGL((bwd_pc_kernel<IO, V, CPL, SYM, INIT, EVAL, U, NTLF, NTSF>), grid, dim3(kBlock), lds, stream, grad, x, dx, \
                       g, static_cast<const T*>(scale), static_cast<const T*>(shift), r, gs, partials)
    LSQ_DISPATCH_VARIANT(false, v, LSQ_LAUNCH);
#undef LSQ_LAUNCH
    return hipGetLastError();
}

template <typename IO, int V, int CPL>
static hipError_t bwd_pc_modes(const void* grad, const void* x, void* dx, const PcGeom& g, const void* scale,
                               const void* shift, const lsq_params& p, typename IO::arith gs, double2* partials,
                               const Variant& v, hipStream_t stream) {
#define LSQ_CASE(S, I, E) \
    return launch_bwd_pc<IO, V, CPL, S, I, E>(grad, x, dx, g, scale, shift, p, gs, partials, v, stream)
    const bool sym = p.sym != 0, init = p.init_mode != 0;
    if (p.eval_mode) {
        if (init) LSQ_CASE(false, true, true);
        LSQ_CASE(false, false, true);
    }
    if (sym) {
        if (init) LSQ_CASE(true, true, false);
        LSQ_CASE(true, false, false);
    }
    if (init) LSQ_CASE(false, true, false);
    LSQ_CASE(false, false, false);
#undef LSQ_CASE
}

template <typename IO>
hipError_t backward_per_channel(const void* grad, const void* x, void* dx, void* ds, void* db, double* wide,
                                int64_t outer, int64_t channels, int64_t inner, const void* scale,
                                const void* shift, const lsq_params& p, void* workspace, size_t workspace_bytes,
                                int variant, hipStream_t stream) {
    using T = typename IO::arith;
    const Variant v = decode_variant(variant, kDefaultPcVariant);
    const DeviceInfo& dev = device_info();
    const bool aligned = is_aligned16(grad) && is_aligned16(x) && is_aligned16(dx);
    const int vec = pick_vec(IO::VEC, channels * inner, aligned);
    const int cpl = pick_cpl(vec, inner);
    const PcGeom g = make_geom(outer, channels, inner, vec, dev.cu_count * v.blocks_per_cu);
    if (!grid_fits(g)) return hipErrorInvalidConfiguration;
    const size_t need = static_cast<size_t>(g.splits) * g.n_windows * g.k_slots * sizeof(double2);
    if (!p.eval_mode && workspace_bytes < need) return hipErrorInvalidValue;
    const int64_t numel = outer * channels * inner;
    const int64_t n4s = p.numel_for_scaler > 0 ? p.numel_for_scaler : numel;
    const T gs = grad_scaler_per_channel<T>(n4s, p.quant_max, channels, p.use_grad_scaling != 0, p.grad_scaler);
    double2* partials = static_cast<double2*>(workspace);
    hipError_t e;
    if (vec == 1) e = bwd_pc_modes<IO, 1, 1>(grad, x, dx, g, scale, shift, p, gs, partials, v, stream);
    else if (cpl == 1) e = bwd_pc_modes<IO, IO::VEC, 1>(grad, x, dx, g, scale, shift, p, gs, partials, v, stream);
    else if (cpl == 2) e = bwd_pc_modes<IO, IO::VEC, 2>(grad, x, dx, g, scale, shift, p, gs, partials, v, stream);
    else e = bwd_pc_modes<IO, IO::VEC, IO::VEC>(grad, x, dx, g, scale, shift, p, gs, partials, v, stream);
    if (e != hipSuccess) return e;
    const T sym_term = static_cast<T>(0) * gs;
    const unsigned fgrid = static_cast<unsigned>((channels + kBlock - 1) / kBlock);
    hipLaunchKernelGGL((finalize_pc_kernel<T>), dim3(fgrid), dim3(kBlock), 0, stream, partials, g, p.eval_mode ? 1 : 0,
                       p.sym ? 1 : 0, sym_term, static_cast<T*>(ds), static_cast<T*>(db), wide);
    return hipGetLastError();
}

#define LSQ_INSTANTIATE(IO)                                                                                          \
    template hipError_t forward_per_channel<IO>(const void*, void*, int64_t, int64_t, int64_t, const void*,          \
                                                const void*, const lsq_params&, const lsq_fwd_extras*, int,          \
                                                hipStream_t);                                                        \
    template hipError_t backward_per_channel<IO>(const void*, const void*, void*, void*, void*, double*, int64_t,    \
                                                 int64_t, int64_t, const void*, const void*, const lsq_params&,      \
                                                 void*, size_t, int, hipStream_t);
LSQ_INSTANTIATE(io_f32)
LSQ_INSTANTIATE(io_f64)
LSQ_INSTANTIATE(io_bf16)
LSQ_INSTANTIATE(io_f16)
#undef LSQ_INSTANTIATE

}  // namespace lsq
