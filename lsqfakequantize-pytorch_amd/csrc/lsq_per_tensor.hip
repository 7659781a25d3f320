// lsq_per_tensor.hip -- K1 (forward) and K2 (fused backward + reduction) for per-tensor LSQ on gfx950.
//
// Replaces the reference's per-tensor CUDA backend (/root/reference/torchlsq/csrc/ops/cuda/lsq_cuda.cu:18-143):
// there the backward is THREE elementwise kernels writing three N-sized temporaries plus two
// at::sum reductions (52 B/element fwd+bwd); here it is one streaming pass (20 B/element, the
// algorithmic minimum) and a one-block finalize.
//
// CDNA4 design
//  * HBM-bound streaming: every lane moves 16-byte packets (global_load/store_dwordx4), a wave
//    instruction covers 1 KiB contiguous; UNROLL independent packets per lane are issued before
//    the first use so ~UNROLL KiB per wave are in flight.
//  * grid = a few workgroups per CU, grid-stride over tiles; 256-thread workgroups (4 wave64).
//  * scale / shift are read from device memory inside the kernel (uniform scalar loads): no
//    host round trip, graph-capturable.
//  * d_scale / d_shift: per-lane fp64 accumulators (the per-element terms are fp32-exact copies of
//    the reference's ds_buffer/db_buffer values; summing them in fp64 keeps the result within a
//    few 1e-8 of the exact sum, inside the 1e-6 parity budget that the reference's own fp32
//    at::sum eats most of) -> wave64 butterfly (DPP/ds_bpermute shuffles) -> 4 partials through LDS
//    -> one 16-byte partial per workgroup in the workspace -> fixed-order fold.  The fold is done by the workgroup
//    that finishes last (a wrapping agent-scope arrival counter in the caller's ticket, lsq_bwd_extras) or, without
//    a ticket, by a one-workgroup finalize launch; either way the partials are added in index order, so
//    the result is bit-deterministic for a given size (the only atomic is the arrival counter).
#include "lsq_kernels.hpp"

namespace lsq {

// Which full tiles a workgroup visits.  strided: tile t belongs to workgroup t % grid, so at any moment the
// whole chip works inside one moving window of grid*tile bytes per stream; chunked: each workgroup owns
// one contiguous run of tiles.
struct TileWalk {
    int64_t first, last, step;
    __device__ __forceinline__ TileWalk(int64_t n_full, bool chunked) {
        if (chunked) {
            const int64_t per = (n_full + gridDim.x - 1) / gridDim.x;
            first = static_cast<int64_t>(blockIdx.x) * per;
            last = first + per < n_full ? first + per : n_full;
            step = 1;
        } else {
            first = blockIdx.x;
            last = n_full;
            step = gridDim.x;
        }
    }
};

// ------------------------------------------------------------------------------------------------
// K1: forward
// ------------------------------------------------------------------------------------------------
template <typename IO, bool INIT, bool LEVELS, int UNROLL, bool NTL, bool NTS>
__global__ __launch_bounds__(kBlock) void fwd_pt_kernel(const void* __restrict__ x, void* __restrict__ y,
                                                        int8_t* __restrict__ levels, int level_bias, int aux_kind,
                                                        int64_t n, const typename IO::arith* __restrict__ scale,
                                                        const typename IO::arith* __restrict__ shift,
                                                        Range<typename IO::arith> r, int chunked) {
    using T = typename IO::arith;
    constexpr int VEC = IO::VEC;
    const QParams<T> q = make_qparams<T>(sanitize_scale_per_tensor<T>(scale[0]), shift[0], r);  // lsq_cpu.cpp:44-47
    const T bias = static_cast<T>(level_bias);

    const int64_t n_packets = n / VEC;
    constexpr int64_t kTile = static_cast<int64_t>(kBlock) * UNROLL;
    const int64_t n_full = n_packets / kTile;

    auto emit = [&](const Packet<IO>& in, int64_t p) {
        Packet<IO> out;
        LevelPack<VEC> lv;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const T xv = static_cast<T>(in.v[j]);
            const T c = clamped<T>(xv, q, r);
            out.v[j] = out_elem<IO, INIT>(INIT ? xv : dequant<T>(rne(c), q));  // lsq_kernel.h:13
            if (LEVELS) lv.b[j] = aux_byte<T>(c, r, bias, aux_kind);
        }
        if (!LEVELS || y != nullptr) {     // y == NULL: only the one-byte output is wanted (include/lsq_hip.h, lsq_fwd_extras)
            if (NTS) store_packet_nt<IO>(y, p * VEC, out); else store_packet<IO>(y, p * VEC, out);
        }
        if (LEVELS) lv.store(levels + p * VEC);
    };

    // full tiles: no predicates, UNROLL independent 16-byte loads per lane in flight
    const TileWalk walk(n_full, chunked != 0);
    for (int64_t tile = walk.first; tile < walk.last; tile += walk.step) {
        const int64_t p0 = tile * kTile + threadIdx.x;
        Packet<IO> in[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t p = p0 + static_cast<int64_t>(u) * kBlock;
            in[u] = NTL ? load_packet_nt<IO>(x, p * VEC) : load_packet<IO>(x, p * VEC);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) emit(in[u], p0 + static_cast<int64_t>(u) * kBlock);
    }
    // the one partial tile, taken by the workgroup whose turn it would be
    if (static_cast<int64_t>(blockIdx.x) == n_full % gridDim.x) {
        for (int64_t p = n_full * kTile + threadIdx.x; p < n_packets; p += kBlock) {
            const Packet<IO> in = load_packet<IO>(x, p * VEC);
            emit(in, p);
        }
    }
    // ragged tail (n % VEC elements): one lane each, first workgroup
    if (blockIdx.x == 0) {
        const int64_t i = n_packets * VEC + threadIdx.x;
        if (i < n) {
            const T xv = IO::load1(x, i);
            const T c = clamped<T>(xv, q, r);
            if (!LEVELS || y != nullptr) store_out<IO, INIT>(y, i, INIT ? xv : dequant<T>(rne(c), q));
            if (LEVELS) levels[i] = aux_byte<T>(c, r, bias, aux_kind);
        }
    }
}

// scalar form, 1 element per lane: only where packets cannot be used at all -- a `levels` buffer that is not 8-byte
// aligned.  Views whose buffers are merely not 16-byte aligned run the packet kernel above (lsq_math.hpp, PacketWord).
template <typename IO, bool INIT, bool LEVELS>
__global__ __launch_bounds__(kBlock) void fwd_pt_scalar_kernel(const void* __restrict__ x, void* __restrict__ y,
                                                               int8_t* __restrict__ levels, int level_bias, int aux_kind,
                                                               int64_t n, const typename IO::arith* __restrict__ scale,
                                                               const typename IO::arith* __restrict__ shift,
                                                               Range<typename IO::arith> r) {
    using T = typename IO::arith;
    const QParams<T> q = make_qparams<T>(sanitize_scale_per_tensor<T>(scale[0]), shift[0], r);
    const T bias = static_cast<T>(level_bias);
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * kBlock) {
        const T xv = IO::load1(x, i);
        const T c = clamped<T>(xv, q, r);
        if (!LEVELS || y != nullptr) store_out<IO, INIT>(y, i, INIT ? xv : dequant<T>(rne(c), q));
        if (LEVELS) levels[i] = aux_byte<T>(c, r, bias, aux_kind);
    }
}

// ------------------------------------------------------------------------------------------------
// K2: backward, one pass: dx + per-workgroup partial sums of the ds / db terms
// ------------------------------------------------------------------------------------------------
template <typename T, bool SYM, bool INIT, bool EVAL>
struct BwdAcc {
    double s = 0.0, b = 0.0;
    __device__ __forceinline__ T step(T g, T x, const QParams<T>& q, const Range<T>& r, T gs) {
        if (EVAL) return backward_elem_eval<T, INIT>(g, x, q, r);
        T ds_t, db_t;
        const T dX = backward_elem<T, SYM, INIT>(g, x, q, r, gs, ds_t, db_t);
        s += static_cast<double>(ds_t);
        if (!SYM) b += static_cast<double>(db_t);
        return dX;
    }
};

// Where d_scale / d_shift go when the kernel finishes them itself (ticket != nullptr), see fold_partials.
template <typename T>
struct PtFold {
    uint32_t* ticket;   // nullptr: leave the partials to finalize_pt_kernel
    T* ds;
    T* db;
    double* wide;
    T sym_term;         // the constant per-element d_shift term of the symmetric case, 0 * grad_scaler (lsq_kernel.h:118,122)
    int sym;
};

// Fold the per-workgroup partials in a FIXED order (lane-strided, wave64 butterfly, the 4 wave totals in order) and
// round once to the parameter type: the reference's `ds_buffer.sum().unsqueeze(0)`, lsq_cpu.cpp:138-139.  One
// workgroup; shared by the last-arriving workgroup of the backward kernel (AGENT = true: the partials were written by
// other workgroups of the same launch) and by finalize_pt_kernel, so both routes give the same bits.
template <typename T, bool AGENT>
__device__ __forceinline__ void fold_partials(const double2* partials, int n_partials, bool eval_mode, const PtFold<T>& f,
                                              double2* wave_tot) {
    double s = 0.0, b = 0.0;
    if (!eval_mode) {
        for (int i = threadIdx.x; i < n_partials; i += kBlock) {
            const double2 v = AGENT ? load_partial_agent(partials + i) : partials[i];
            s += v.x;
            b += v.y;
        }
    }
    s = wave_sum(s);
    b = wave_sum(b);
    if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = make_double2(s, b);
    __syncthreads();
    if (threadIdx.x == 0) {
        double ts = 0.0, tb = 0.0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) {
            ts += wave_tot[w].x;
            tb += wave_tot[w].y;
        }
        if (!eval_mode && f.sym) tb = 0.0 + static_cast<double>(f.sym_term);  // sum of N copies of (0*gs): +0, or NaN
        f.ds[0] = static_cast<T>(ts);
        f.db[0] = static_cast<T>(tb);
        if (f.wide) {
            f.wide[0] = ts;
            f.wide[1] = tb;
        }
    }
}

// workgroup reduction of (s, b): wave64 butterfly, then the 4 wave totals through LDS -> this workgroup's partial.
// With a ticket, the workgroup that arrives last folds all partials and stores d_scale / d_shift.
template <typename T>
__device__ __forceinline__ void block_reduce_store(double s, double b, double2* __restrict__ partials, const PtFold<T>& f) {
    __shared__ double2 wave_tot[kBlock / 64];
    __shared__ int is_last;
    s = wave_sum(s);
    b = wave_sum(b);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_tot[wave] = make_double2(s, b);
    __syncthreads();
    if (threadIdx.x == 0) {
        double ts = 0.0, tb = 0.0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) {
            ts += wave_tot[w].x;
            tb += wave_tot[w].y;
        }
        if (f.ticket) {
            store_partial_agent(partials + blockIdx.x, ts, tb);
            is_last = ticket_arrive_is_last(f.ticket, gridDim.x) ? 1 : 0;
        } else {
            partials[blockIdx.x] = make_double2(ts, tb);
        }
    }
    if (!f.ticket) return;
    __syncthreads();
    if (!is_last) return;
    fold_partials<T, true>(partials, static_cast<int>(gridDim.x), false, f, wave_tot);
}

// eval mode with a ticket: nothing to reduce, d_scale = d_shift = 0 (lsq_kernel.h:142-144)
template <typename T>
__device__ __forceinline__ void eval_store_zero(const PtFold<T>& f) {
    if (f.ticket && blockIdx.x == 0 && threadIdx.x == 0) {
        f.ds[0] = static_cast<T>(0);
        f.db[0] = static_cast<T>(0);
        if (f.wide) {
            f.wide[0] = 0.0;
            f.wide[1] = 0.0;
        }
    }
}

template <typename IO, bool SYM, bool INIT, bool EVAL, int UNROLL, bool NTL, bool NTS>
__global__ __launch_bounds__(kBlock) void bwd_pt_kernel(const void* __restrict__ grad, const void* __restrict__ x,
                                                        void* __restrict__ dx, int64_t n,
                                                        const typename IO::arith* __restrict__ scale,
                                                        const typename IO::arith* __restrict__ shift,
                                                        Range<typename IO::arith> r, typename IO::arith grad_scaler,
                                                        double2* __restrict__ partials, PtFold<typename IO::arith> fold,
                                                        int chunked) {
    using T = typename IO::arith;
    constexpr int VEC = IO::VEC;
    const QParams<T> q = make_qparams<T>(sanitize_scale_per_tensor<T>(scale[0]), shift[0], r);  // lsq_cpu.cpp:99-102
    BwdAcc<T, SYM, INIT, EVAL> acc;

    const int64_t n_packets = n / VEC;
    constexpr int64_t kTile = static_cast<int64_t>(kBlock) * UNROLL;
    const int64_t n_full = n_packets / kTile;

    auto emit = [&](const Packet<IO>& gi, const Packet<IO>& xi, int64_t p) {
        Packet<IO> out;
#pragma unroll
        for (int j = 0; j < VEC; ++j)
            out.v[j] = out_elem<IO, INIT>(          // (init_mode: dX IS the gradient, lsq_kernel.h:112)
                acc.step(static_cast<T>(gi.v[j]), static_cast<T>(xi.v[j]), q, r, grad_scaler));
        if (NTS) store_packet_nt<IO>(dx, p * VEC, out); else store_packet<IO>(dx, p * VEC, out);
    };

    // full tiles: no predicates, 2*UNROLL independent 16-byte loads per lane in flight
    const TileWalk walk(n_full, chunked != 0);
    for (int64_t tile = walk.first; tile < walk.last; tile += walk.step) {
        const int64_t p0 = tile * kTile + threadIdx.x;
        Packet<IO> gi[UNROLL], xi[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t p = p0 + static_cast<int64_t>(u) * kBlock;
            gi[u] = NTL ? load_packet_nt<IO>(grad, p * VEC) : load_packet<IO>(grad, p * VEC);
            xi[u] = NTL ? load_packet_nt<IO>(x, p * VEC) : load_packet<IO>(x, p * VEC);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) emit(gi[u], xi[u], p0 + static_cast<int64_t>(u) * kBlock);
    }
    if (static_cast<int64_t>(blockIdx.x) == n_full % gridDim.x) {
        for (int64_t p = n_full * kTile + threadIdx.x; p < n_packets; p += kBlock) {
            const Packet<IO> gi = load_packet<IO>(grad, p * VEC);
            const Packet<IO> xi = load_packet<IO>(x, p * VEC);
            emit(gi, xi, p);
        }
    }
    if (blockIdx.x == 0) {
        const int64_t i = n_packets * VEC + threadIdx.x;
        if (i < n) store_out<IO, INIT>(dx, i, acc.step(IO::load1(grad, i), IO::load1(x, i), q, r, grad_scaler));
    }
    if (EVAL) eval_store_zero<T>(fold);
    else block_reduce_store<T>(acc.s, acc.b, partials, fold);
}

template <typename IO, bool SYM, bool INIT, bool EVAL>
__global__ __launch_bounds__(kBlock) void bwd_pt_scalar_kernel(const void* __restrict__ grad, const void* __restrict__ x,
                                                               void* __restrict__ dx, int64_t n,
                                                               const typename IO::arith* __restrict__ scale,
                                                               const typename IO::arith* __restrict__ shift,
                                                               Range<typename IO::arith> r,
                                                               typename IO::arith grad_scaler,
                                                               double2* __restrict__ partials,
                                                               PtFold<typename IO::arith> fold) {
    using T = typename IO::arith;
    const QParams<T> q = make_qparams<T>(sanitize_scale_per_tensor<T>(scale[0]), shift[0], r);
    BwdAcc<T, SYM, INIT, EVAL> acc;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * kBlock)
        store_out<IO, INIT>(dx, i, acc.step(IO::load1(grad, i), IO::load1(x, i), q, r, grad_scaler));
    if (EVAL) eval_store_zero<T>(fold);
    else block_reduce_store<T>(acc.s, acc.b, partials, fold);
}

// Finalize (the route without a ticket): ONE workgroup folds the per-workgroup partials, see fold_partials.
// eval_mode: ds = db = 0 (lsq_kernel.h:142-144).
template <typename T>
__global__ __launch_bounds__(kBlock) void finalize_pt_kernel(const double2* __restrict__ partials, int n_partials,
                                                             int eval_mode, PtFold<T> f) {
    __shared__ double2 wave_tot[kBlock / 64];
    fold_partials<T, false>(partials, n_partials, eval_mode != 0, f, wave_tot);
}

// ------------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------------
template <typename IO, bool INIT, bool LEVELS>
static hipError_t launch_fwd_pt(const void* x, void* y, int8_t* levels, int level_bias, int aux_kind, int64_t n,
                                const void* scale, const void* shift, const lsq_params& p, int variant,
                                hipStream_t stream) {
    using T = typename IO::arith;
    const Range<T> r = make_range<T>(p);
    const T* sc = static_cast<const T*>(scale);
    const T* sh = static_cast<const T*>(shift);
    // packets need element alignment only (PacketWord); the 8 / 4 level bytes of a packet are stored as one word
    const bool aligned = is_elem_aligned<IO>(x) && is_elem_aligned<IO>(y) && (!levels || (reinterpret_cast<uintptr_t>(levels) & 7u) == 0);
    const DeviceInfo& dev = device_info();
    if (!aligned) {
        const int64_t want = (n + kBlock - 1) / kBlock;
        const int grid = static_cast<int>(std::min<int64_t>(want, static_cast<int64_t>(dev.cu_count) * 16));
        hipLaunchKernelGGL((fwd_pt_scalar_kernel<IO, INIT, LEVELS>), dim3(grid), dim3(kBlock), 0, stream, x, y, levels,
                           level_bias, aux_kind, n, sc, sh, r);
        return hipGetLastError();
    }
    const Variant v = decode_variant(variant, kDefaultFwdVariant);
    const int64_t n_packets = n / IO::VEC;
    const int64_t tile = static_cast<int64_t>(kBlock) * v.unroll;
    const int64_t n_tiles = std::max<int64_t>(1, (n_packets + tile - 1) / tile);
    const int grid = static_cast<int>(std::min<int64_t>(n_tiles, static_cast<int64_t>(dev.cu_count) * v.blocks_per_cu));
    // (the partial tile is owned by workgroup n_full % grid, which exists because grid <= n_tiles)
#define LSQ_LAUNCH_FWD(U, NTLF, NTSF)                                                                                   \
    hipLaunchKernelGGL((fwd_pt_kernel<IO, INIT, LEVELS, U, NTLF, NTSF>), dim3(grid), dim3(kBlock), 0, stream, x, y, levels, \
                       level_bias, aux_kind, n, sc, sh, r, v.chunked ? 1 : 0)
    [[maybe_unused]] constexpr bool kFull = std::is_same<IO, io_f32>::value && !INIT && !LEVELS;
    LSQ_DISPATCH_VARIANT(kFull, 4, v, LSQ_LAUNCH_FWD);
#undef LSQ_LAUNCH_FWD
    return hipGetLastError();
}

template <typename IO>
hipError_t forward_per_tensor(const void* x, void* y, int64_t n, const void* scale, const void* shift,
                              const lsq_params& p, const lsq_fwd_extras* ex, int variant, hipStream_t stream) {
    int8_t* levels = ex ? static_cast<int8_t*>(ex->levels) : nullptr;
    const int bias = ex ? ex->level_bias : 0;
    const int aux_kind = ex ? ex->aux_kind : 0;
    if (p.init_mode) {
        return levels ? launch_fwd_pt<IO, true, true>(x, y, levels, bias, aux_kind, n, scale, shift, p, variant, stream)
                      : launch_fwd_pt<IO, true, false>(x, y, levels, bias, aux_kind, n, scale, shift, p, variant, stream);
    }
    return levels ? launch_fwd_pt<IO, false, true>(x, y, levels, bias, aux_kind, n, scale, shift, p, variant, stream)
                  : launch_fwd_pt<IO, false, false>(x, y, levels, bias, aux_kind, n, scale, shift, p, variant, stream);
}

int bwd_pt_grid(int64_t n, int vec, const Variant& v, bool aligned) {
    const DeviceInfo& dev = device_info();
    if (!aligned) {
        const int64_t want = std::max<int64_t>(1, (n + kBlock - 1) / kBlock);
        return static_cast<int>(std::min<int64_t>(want, static_cast<int64_t>(dev.cu_count) * 16));
    }
    const int64_t n_packets = n / vec;
    const int64_t tile = static_cast<int64_t>(kBlock) * v.unroll;
    const int64_t n_tiles = std::max<int64_t>(1, (n_packets + tile - 1) / tile);
    return static_cast<int>(std::min<int64_t>(n_tiles, static_cast<int64_t>(dev.cu_count) * v.blocks_per_cu));
}

size_t bwd_pt_workspace_bytes() {
    // one double2 per workgroup, for the largest grid any variant may launch
    return static_cast<size_t>(kMaxCUs) * kMaxBlocksPerCU * sizeof(double2);
}

template <typename IO, bool SYM, bool INIT, bool EVAL>
static hipError_t launch_bwd_pt(const void* grad, const void* x, void* dx, void* ds, void* db, double* wide,
                                int64_t n, const void* scale, const void* shift, const lsq_params& p,
                                void* workspace, uint32_t* ticket, int variant, hipStream_t stream) {
    using T = typename IO::arith;
    const Range<T> r = make_range<T>(p);
    const T* sc = static_cast<const T*>(scale);
    const T* sh = static_cast<const T*>(shift);
    const int64_t n4s = p.numel_for_scaler > 0 ? p.numel_for_scaler : n;
    const T gs = grad_scaler_per_tensor<T>(n4s, p.quant_max, p.use_grad_scaling != 0, p.grad_scaler);
    double2* partials = static_cast<double2*>(workspace);
    const bool aligned = is_elem_aligned<IO>(grad) && is_elem_aligned<IO>(x) && is_elem_aligned<IO>(dx);    // (PacketWord: any view)
    const Variant v = decode_variant(variant, kDefaultBwdVariant);
    const int grid = bwd_pt_grid(n, IO::VEC, v, aligned);
    const T sym_term = static_cast<T>(0) * gs;
    const PtFold<T> fold{ticket, static_cast<T*>(ds), static_cast<T*>(db), wide, sym_term, SYM ? 1 : 0};
    if (!aligned) {
        hipLaunchKernelGGL((bwd_pt_scalar_kernel<IO, SYM, INIT, EVAL>), dim3(grid), dim3(kBlock), 0, stream, grad, x,
                           dx, n, sc, sh, r, gs, partials, fold);
    } else {
#define LSQ_LAUNCH_BWD(U, NTLF, NTSF)                                                                                    \
    hipLaunchKernelGGL((bwd_pt_kernel<IO, SYM, INIT, EVAL, U, NTLF, NTSF>), dim3(grid), dim3(kBlock), 0, stream, grad, x, dx, \
                       n, sc, sh, r, gs, partials, fold, v.chunked ? 1 : 0)
        [[maybe_unused]] constexpr bool kFull = std::is_same<IO, io_f32>::value && !SYM && !INIT && !EVAL;
        LSQ_DISPATCH_VARIANT(kFull, 4, v, LSQ_LAUNCH_BWD);
#undef LSQ_LAUNCH_BWD
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || ticket) return e;   // with a ticket the last workgroup has stored d_scale / d_shift
    hipLaunchKernelGGL((finalize_pt_kernel<T>), dim3(1), dim3(kBlock), 0, stream, partials, grid, EVAL ? 1 : 0, fold);
    return hipGetLastError();
}

template <typename IO>
hipError_t backward_per_tensor(const void* grad, const void* x, void* dx, void* ds, void* db, double* wide,
                               int64_t n, const void* scale, const void* shift, const lsq_params& p,
                               void* workspace, uint32_t* ticket, int variant, hipStream_t stream) {
#define LSQ_BWD_CASE(S, I, E) \
    return launch_bwd_pt<IO, S, I, E>(grad, x, dx, ds, db, wide, n, scale, shift, p, workspace, ticket, variant, stream)
    const bool sym = p.sym != 0, init = p.init_mode != 0;
    if (p.eval_mode) {
        if (init) LSQ_BWD_CASE(false, true, true);
        LSQ_BWD_CASE(false, false, true);
    }
    if (sym) {
        if (init) LSQ_BWD_CASE(true, true, false);
        LSQ_BWD_CASE(true, false, false);
    }
    if (init) LSQ_BWD_CASE(false, true, false);
    LSQ_BWD_CASE(false, false, false);
#undef LSQ_BWD_CASE
}

// ------------------------------------------------------------------------------------------------
// eval-mode backward from the 1-byte inside mask the forward saved: dx = grad * mask
// ------------------------------------------------------------------------------------------------
template <typename IO, int UNROLL>
__global__ __launch_bounds__(kBlock) void bwd_mask_kernel(const void* __restrict__ grad, const int8_t* __restrict__ mask,
                                                          void* __restrict__ dx, int64_t n, int vec_ok) {
    using T = typename IO::arith;
    constexpr int VEC = IO::VEC;
    const int64_t n_packets = vec_ok ? n / VEC : 0;
    constexpr int64_t kTile = static_cast<int64_t>(kBlock) * UNROLL;
    const int64_t n_full = n_packets / kTile;
    auto emit = [&](const Packet<IO>& gi, const LevelPack<VEC>& m, int64_t p) {
        Packet<IO> out;
#pragma unroll
        for (int j = 0; j < VEC; ++j)   // a real multiply, like the reference (inf * 0 = NaN)
            out.v[j] = IO::to_elem(static_cast<T>(gi.v[j]) * static_cast<T>(m.b[j]));
        store_packet_nt<IO>(dx, p * VEC, out);
    };
    for (int64_t tile = blockIdx.x; tile < n_full; tile += gridDim.x) {
        const int64_t p0 = tile * kTile + threadIdx.x;
        Packet<IO> gi[UNROLL];
        LevelPack<VEC> mi[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t p = p0 + static_cast<int64_t>(u) * kBlock;
            gi[u] = load_packet_nt<IO>(grad, p * VEC);
            mi[u].load(mask + p * VEC);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) emit(gi[u], mi[u], p0 + static_cast<int64_t>(u) * kBlock);
    }
    if (static_cast<int64_t>(blockIdx.x) == n_full % gridDim.x) {
        for (int64_t p = n_full * kTile + threadIdx.x; p < n_packets; p += kBlock) {
            LevelPack<VEC> m;
            m.load(mask + p * VEC);
            emit(load_packet<IO>(grad, p * VEC), m, p);
        }
    }
    // element-wise remainder (everything, when the buffers are not aligned for packets)
    for (int64_t i = n_packets * VEC + static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * kBlock)
        IO::store1(dx, i, IO::load1(grad, i) * static_cast<T>(mask[i]));
}

template <typename IO>
hipError_t backward_from_mask(const void* grad, const void* mask, void* dx, int64_t n, hipStream_t stream) {
    const DeviceInfo& dev = device_info();
    const bool aligned = is_elem_aligned<IO>(grad) && is_elem_aligned<IO>(dx) && (reinterpret_cast<uintptr_t>(mask) & 7u) == 0;
    constexpr int kU = 4;
    const int64_t tile = static_cast<int64_t>(kBlock) * kU * IO::VEC;
    const int64_t want = std::max<int64_t>(1, (n + tile - 1) / tile);
    const int grid = static_cast<int>(std::min<int64_t>(want, static_cast<int64_t>(dev.cu_count) * 8));
    hipLaunchKernelGGL((bwd_mask_kernel<IO, kU>), dim3(grid), dim3(kBlock), 0, stream, grad,
                       static_cast<const int8_t*>(mask), dx, n, aligned ? 1 : 0);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// batch-sharded backward, the epilogue after the all-reduce (lsq_hip_sharded_finish): packed = {sum ds terms [C],
// sum db terms [C], element count}, every slot summed over the ranks; the terms are UNSCALED (the local backward ran with
// use_grad_scaling = 0, grad_scaler = 1), so the scaler of lsq_cpu.cpp:103-104 / :250-251 -- same precision chain as
// grad_scaler_per_tensor / _per_channel above, evaluated on the device because the global count only exists there --
// multiplies the fp64 sums once, and the product is rounded once to the parameter type.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void sharded_finish_kernel(const double* __restrict__ packed, int64_t channels,
                                                                int per_channel, T qmax, int use_grad_scaling,
                                                                double grad_scaler, T* __restrict__ ds, T* __restrict__ db) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (c >= channels) return;
    const double count = packed[2 * channels];
    T gs = static_cast<T>(grad_scaler);
    if (use_grad_scaling) {
        T prod = static_cast<T>(count) * qmax;                      // int64 numel -> scalar_t, times scalar_t(quant_max)
        if (per_channel) prod = prod / static_cast<T>(channels);
        gs = static_cast<T>(grad_scaler / static_cast<double>(sqrt(prod)));
    }
    if (!(count > 0.0)) gs = static_cast<T>(0);                      // every shard empty: nothing was summed
    ds[c] = static_cast<T>(packed[c] * static_cast<double>(gs));
    db[c] = static_cast<T>(packed[channels + c] * static_cast<double>(gs));
}

template <typename T>
hipError_t sharded_finish(const double* packed, int64_t channels, bool per_channel, const lsq_params& p, void* ds, void* db,
                          hipStream_t stream) {
    const int grid = static_cast<int>((channels + kBlock - 1) / kBlock);
    hipLaunchKernelGGL((sharded_finish_kernel<T>), dim3(grid), dim3(kBlock), 0, stream, packed, channels, per_channel ? 1 : 0,
                       static_cast<T>(p.quant_max), p.use_grad_scaling ? 1 : 0, p.grad_scaler, static_cast<T*>(ds),
                       static_cast<T*>(db));
    return hipGetLastError();
}
template hipError_t sharded_finish<float>(const double*, int64_t, bool, const lsq_params&, void*, void*, hipStream_t);
template hipError_t sharded_finish<double>(const double*, int64_t, bool, const lsq_params&, void*, void*, hipStream_t);

// explicit instantiations used by the C ABI (lsq_capi.hip)
#define LSQ_INSTANTIATE(IO)                                                                                        \
    template hipError_t forward_per_tensor<IO>(const void*, void*, int64_t, const void*, const void*,              \
                                               const lsq_params&, const lsq_fwd_extras*, int, hipStream_t);        \
    template hipError_t backward_per_tensor<IO>(const void*, const void*, void*, void*, void*, double*, int64_t,   \
                                                const void*, const void*, const lsq_params&, void*, uint32_t*, int, hipStream_t); \
    template hipError_t backward_from_mask<IO>(const void*, const void*, void*, int64_t, hipStream_t);
LSQ_INSTANTIATE(io_f32)
LSQ_INSTANTIATE(io_f64)
LSQ_INSTANTIATE(io_bf16)
LSQ_INSTANTIATE(io_f16)
#undef LSQ_INSTANTIATE

}  // namespace lsq
