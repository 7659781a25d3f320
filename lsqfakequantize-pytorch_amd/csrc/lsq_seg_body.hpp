// lsq_seg_body.hpp -- SEGMENT mode of the per-channel kernels: one workgroup walks (a segment of) ONE channel.
// Shared by the single-tensor kernels (lsq_per_channel.hip: fwd_seg_kernel / bwd_seg_kernel, conv / linear weights on
// axis 0) and the multi-tensor kernels (lsq_multi.hip: many weight quantizers in one launch) -- the same walk, the same
// summation order, hence the same bits.
#pragma once

#include "lsq_kernels.hpp"
#include "lsq_pc_geom.hpp"

namespace lsq {

constexpr int kSegUpFront = 8;     // walks of up to this many iterations are issued as one group (seg_forward)
// ... the backward holds two packets per iteration and unpacks them to fp32: five (4- / 8-byte storage) or three (16-bit)
// iterations up front keep it at three waves per SIMD -- BASELINE config 3 is five / three iterations
template <typename IO>
constexpr int kSegUpFrontBwd = sizeof(typename IO::elem) >= 4 ? 5 : 3;

// Segment mode with ONE workgroup per channel (segs == osplits == 1: every conv / linear weight whose channel row is
// not worth splitting -- the usual weight quantizer): the workgroup's sums ARE the channel's sums, so the kernel
// rounds and stores d_scale / d_shift itself and the finalize launch (3-4 us, a third of the backward of a
// BASELINE-config-3-sized weight) disappears.  ds == nullptr selects the partials + finalize route.
template <typename T>
struct SegDirect {
    T* ds;
    T* db;
    double* wide;
    T sym_term;
    __device__ __forceinline__ void write(int64_t c, int64_t C, double ts, double tb) const {
        ds[c] = static_cast<T>(ts);
        db[c] = static_cast<T>(tb);
        if (wide) {
            wide[c] = ts;
            wide[C + c] = tb;
        }
    }
};

// One channel segment of the forward: the workgroup's walk `w` over channel w.c of the [outer, C, inner] tensor.
// WALK: 0 = any walk (the multi-tensor kernels), 1 = short walks only (at most kSegUpFront iterations: the caller checked),
// 2 = the loop only -- the single-tensor kernels are compiled per kind so that the up-front groups' registers do not cost the
// long walks their occupancy ([4,8,1048576] bf16 forward 24.4 -> 26.9 us when both forms shared one kernel).
template <typename IO, int V, bool INIT, bool LEVELS, int UNROLL, bool NTL, bool NTS, int WALK = 0>
__device__ __forceinline__ void seg_forward(const void* __restrict__ x, void* __restrict__ y, int8_t* __restrict__ levels,
                                            int level_bias, int aux_kind, const SegGeom& g, const SegWalk& w,
                                            const typename IO::arith* __restrict__ scale,
                                            const typename IO::arith* __restrict__ shift,
                                            const Range<typename IO::arith>& r) {
    using T = typename IO::arith;
    using E = typename IO::elem;
    const QParams<T> q = make_qparams<T>(sanitize_scale_per_channel<T>(scale[w.c]), shift[w.c], r);
    const T bias = static_cast<T>(level_bias);
    const int64_t W = static_cast<int64_t>(kBlock) * V;
    const int64_t q0 = static_cast<int64_t>(threadIdx.x) * V;
    const int64_t q_last = (g.inner - V);   // last packet of a channel row (inner % V == 0)

    // iteration it -> element index of the lane's packet (clamped into the row) and its validity
    auto site = [&](int64_t it, bool& valid) {
        const int64_t oi = static_cast<int64_t>(static_cast<uint32_t>(it) / static_cast<uint32_t>(w.n_r));
        const int64_t ri = it - oi * w.n_r;
        const int64_t pos = (w.r_begin + ri) * W + q0;
        valid = pos < g.inner;
        return ((w.o_begin + oi) * g.C + w.c) * g.inner + (valid ? pos : q_last);
    };
    auto emit = [&](int64_t e, const E (&in)[V], bool valid) {
        E out[V];
        LevelPack<V> lv;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const T xv = static_cast<T>(in[j]);
            const T c = clamped<T>(xv, q, r);
            out[j] = out_elem<IO, INIT>(INIT ? xv : dequant<T>(rne(c), q));
            if (LEVELS) lv.b[j] = aux_byte<T>(c, r, bias, aux_kind);
        }
        if (valid) {
            if (!LEVELS || y != nullptr) store_elems<IO, V, NTS>(y, e, out);     // y == NULL: the one-byte output only
            if (LEVELS) lv.store(levels + e);
        }
    };
    // n_it = full groups of UNROLL + (if left) one group of UNROLL/2 + ... + one single iteration: every slot of
    // every group is a real iteration (a padded last group would load and compute for nothing; a weight channel has
    // only a handful of iterations, so that was up to half of the kernel's work)
    auto group = [&](int64_t it, auto width) {
        constexpr int H = decltype(width)::value;
        E in[H][V];
        int64_t e[H];
        bool ok[H];
#pragma unroll
        for (int u = 0; u < H; ++u) {
            e[u] = site(it + u, ok[u]);
            load_elems<IO, V, NTL>(x, e[u], in[u]);
        }
#pragma unroll
        for (int u = 0; u < H; ++u) emit(e[u], in[u], ok[u]);
    };
    // A weight channel is a handful of iterations (BASELINE config 3: five in fp32, three in bf16): up to kSegUpFront of them go
    // out as ONE group of exactly that size -- every load of the walk in flight before the first use, one memory round trip
    // instead of one per UNROLL-sized group (the order of the arithmetic, hence every sum, is unchanged).
    if constexpr (WALK != 2) {
        if (WALK == 1 || w.n_it <= kSegUpFront) {
            switch (static_cast<int>(w.n_it)) {
                case 1: group(0, std::integral_constant<int, 1>{}); break;
                case 2: group(0, std::integral_constant<int, 2>{}); break;
                case 3: group(0, std::integral_constant<int, 3>{}); break;
                case 4: group(0, std::integral_constant<int, 4>{}); break;
                case 5: group(0, std::integral_constant<int, 5>{}); break;
                case 6: group(0, std::integral_constant<int, 6>{}); break;
                case 7: group(0, std::integral_constant<int, 7>{}); break;
                case 8: group(0, std::integral_constant<int, 8>{}); break;
                default: break;
            }
            return;
        }
    }
    if constexpr (WALK == 1) return;
    int64_t it = 0;
    for (; it + UNROLL <= w.n_it; it += UNROLL) group(it, std::integral_constant<int, UNROLL>{});
    if constexpr (UNROLL >= 8) if (it + 4 <= w.n_it) { group(it, std::integral_constant<int, 4>{}); it += 4; }
    if constexpr (UNROLL >= 4) if (it + 2 <= w.n_it) { group(it, std::integral_constant<int, 2>{}); it += 2; }
    if constexpr (UNROLL >= 2) if (it < w.n_it) group(it, std::integral_constant<int, 1>{});
}

// One channel segment of the backward (dx + the segment's d_scale / d_shift sums).  direct.ds != nullptr: the segment is
// the whole channel, its sums are rounded and stored here; otherwise they go to partials[partial_index].
template <typename IO, int V, bool SYM, bool INIT, bool EVAL, int UNROLL, bool NTL, bool NTS, int WALK = 0>
__device__ __forceinline__ void seg_backward(const void* __restrict__ grad, const void* __restrict__ x, void* __restrict__ dx,
                                             const SegGeom& g, const SegWalk& w,
                                             const typename IO::arith* __restrict__ scale,
                                             const typename IO::arith* __restrict__ shift,
                                             const Range<typename IO::arith>& r, typename IO::arith grad_scaler,
                                             double2* __restrict__ partials, int64_t partial_index,
                                             const SegDirect<typename IO::arith>& direct) {
    using T = typename IO::arith;
    using E = typename IO::elem;
    __shared__ double2 wave_tot[kBlock / 64];
    const QParams<T> q = make_qparams<T>(sanitize_scale_per_channel<T>(scale[w.c]), shift[w.c], r);
    const int64_t W = static_cast<int64_t>(kBlock) * V;
    const int64_t q0 = static_cast<int64_t>(threadIdx.x) * V;
    const int64_t q_last = (g.inner - V);
    double acc_s = 0.0, acc_b = 0.0;

    auto site = [&](int64_t it, bool& valid) {
        const int64_t oi = static_cast<int64_t>(static_cast<uint32_t>(it) / static_cast<uint32_t>(w.n_r));
        const int64_t ri = it - oi * w.n_r;
        const int64_t pos = (w.r_begin + ri) * W + q0;
        valid = pos < g.inner;
        return ((w.o_begin + oi) * g.C + w.c) * g.inner + (valid ? pos : q_last);
    };
    auto emit = [&](int64_t e, const E (&gi)[V], const E (&xi)[V], bool valid) {
        E out[V];
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const T gv = static_cast<T>(gi[j]), xv = static_cast<T>(xi[j]);
            if (EVAL) {
                out[j] = out_elem<IO, INIT>(backward_elem_eval<T, INIT>(gv, xv, q, r));
            } else {
                T ds_t, db_t;
                out[j] = out_elem<IO, INIT>(backward_elem<T, SYM, INIT>(gv, xv, q, r, grad_scaler, ds_t, db_t));
                if (!valid) { ds_t = static_cast<T>(0); db_t = static_cast<T>(0); }
                acc_s += static_cast<double>(ds_t);
                if (!SYM) acc_b += static_cast<double>(db_t);
            }
        }
        if (valid) store_elems<IO, V, NTS>(dx, e, out);
    };
    auto group = [&](int64_t it, auto width) {     // see fwd_seg_kernel: groups of UNROLL, then UNROLL/2, ..., 1
        constexpr int H = decltype(width)::value;
        E gi[H][V], xi[H][V];
        int64_t e[H];
        bool ok[H];
#pragma unroll
        for (int u = 0; u < H; ++u) {
            e[u] = site(it + u, ok[u]);
            load_elems<IO, V, NTL>(grad, e[u], gi[u]);
            load_elems<IO, V, NTL>(x, e[u], xi[u]);
        }
#pragma unroll
        for (int u = 0; u < H; ++u) emit(e[u], gi[u], xi[u], ok[u]);
    };
    int64_t it = 0;
    if constexpr (WALK != 2) {
        if (WALK == 1 || w.n_it <= kSegUpFrontBwd<IO>) {       // see seg_forward: the whole walk as one group
            switch (static_cast<int>(w.n_it)) {
                case 1: group(0, std::integral_constant<int, 1>{}); break;
                case 2: group(0, std::integral_constant<int, 2>{}); break;
                case 3: group(0, std::integral_constant<int, 3>{}); break;
                case 4: if constexpr (kSegUpFrontBwd<IO> >= 4) group(0, std::integral_constant<int, 4>{}); break;
                case 5: if constexpr (kSegUpFrontBwd<IO> >= 5) group(0, std::integral_constant<int, 5>{}); break;
                default: break;
            }
            it = w.n_it;
        }
    }
    if constexpr (WALK != 1) {
        for (; it + UNROLL <= w.n_it; it += UNROLL) group(it, std::integral_constant<int, UNROLL>{});
        if constexpr (UNROLL >= 8) if (it + 4 <= w.n_it) { group(it, std::integral_constant<int, 4>{}); it += 4; }
        if constexpr (UNROLL >= 4) if (it + 2 <= w.n_it) { group(it, std::integral_constant<int, 2>{}); it += 2; }
        if constexpr (UNROLL >= 2) if (it < w.n_it) group(it, std::integral_constant<int, 1>{});
    }
    if (EVAL) {
        if (direct.ds && threadIdx.x == 0) direct.write(w.c, g.C, 0.0, 0.0);   // d_scale = d_shift = 0 (lsq_kernel.h:142-144)
        return;
    }
    acc_s = wave_sum(acc_s);
    acc_b = wave_sum(acc_b);
    if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = make_double2(acc_s, acc_b);
    __syncthreads();
    if (threadIdx.x == 0) {
        double ts = 0.0, tb = 0.0;
#pragma unroll
        for (int k = 0; k < kBlock / 64; ++k) { ts += wave_tot[k].x; tb += wave_tot[k].y; }
        if (direct.ds) {   // this workgroup holds the channel's only partial: finish here, no finalize launch
            if (SYM) tb = 0.0 + static_cast<double>(direct.sym_term);
            direct.write(w.c, g.C, ts, tb);
        } else {
            partials[partial_index] = make_double2(ts, tb);
        }
    }
}

}  // namespace lsq
