// lsq_observe.hip -- observer statistics (running min / max; mean / std) for the init phase of LSQFakeQuantizer.
//
// during its initialisation batches the reference module runs a
// torch MinMax observer over the input right before the fake-quantize op
// (/root/reference/torchlsq/quantized/modules/observers.py:446-449 ->
// torch.ao.quantization.observer.*MinMaxObserver.forward -> torch.aminmax).  For the per-channel
// observers that is permute + flatten (a full copy of x) + aminmax: three passes over HBM.  These
// kernels do it in ONE read-only pass at the HBM read roofline, with the same machinery as the
// fake-quantize kernels: 16-byte packets per lane, unpredicated unrolled loads, wave64 shuffle
// reduction, channel-stationary lanes + LDS atomics (ds_min_u32 / ds_max_u32 on order-preserving
// integer keys) for the per-channel case, one partial per workgroup, fixed-order finalize.
//
// Semantics = torch.aminmax: exact min and max; NaN anywhere (in a channel) makes both results NaN.
#include "lsq_kernels.hpp"
#include "lsq_pc_geom.hpp"

namespace lsq {

template <typename T>
struct alignas(8) MinMaxPartial {
    T mn, mx;
    int32_t nan;
    int32_t pad;
};

template <typename T> struct inf_of;
template <> struct inf_of<float> { __device__ static float value() { return __builtin_huge_valf(); } };
template <> struct inf_of<double> { __device__ static double value() { return __builtin_huge_val(); } };

// order-preserving integer keys: k(a) < k(b)  <=>  a < b  (with -0 < +0), for LDS integer atomics
__device__ __forceinline__ uint32_t order_key(float v) {
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_key(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ unsigned long long order_key(double v) {
    const unsigned long long u = static_cast<unsigned long long>(__double_as_longlong(v));
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double from_key(unsigned long long k) {
    return __longlong_as_double(static_cast<long long>((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k));
}
template <typename T> struct key_of;
template <> struct key_of<float> { using type = uint32_t; };
template <> struct key_of<double> { using type = unsigned long long; };

template <typename T>
struct RunningMinMax {
    T mn, mx;
    bool nan;
    __device__ __forceinline__ RunningMinMax() : mn(inf_of<T>::value()), mx(-inf_of<T>::value()), nan(false) {}
    __device__ __forceinline__ void push(T v) {   // fmin/fmax drop NaNs; they are tracked separately
        mn = fmin_(mn, v);
        mx = fmax_(mx, v);
        nan = nan || (v != v);
    }
};

__device__ __forceinline__ float shfl_xor_t(float v, int m) { return __shfl_xor(v, m, 64); }
__device__ __forceinline__ double shfl_xor_t(double v, int m) { return shfl_xor_f64(v, m); }

// wave64 butterfly, then the 4 wave results through LDS; thread 0 returns the workgroup's result
template <typename T>
__device__ __forceinline__ MinMaxPartial<T> block_minmax(RunningMinMax<T> r) {
    __shared__ MinMaxPartial<T> wave_res[kBlock / 64];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        r.mn = fmin_(r.mn, shfl_xor_t(r.mn, m));
        r.mx = fmax_(r.mx, shfl_xor_t(r.mx, m));
    }
    const bool any_nan = __any(r.nan ? 1 : 0) != 0;
    if ((threadIdx.x & 63) == 0) wave_res[threadIdx.x >> 6] = MinMaxPartial<T>{r.mn, r.mx, any_nan ? 1 : 0, 0};
    __syncthreads();
    MinMaxPartial<T> out = wave_res[0];
#pragma unroll
    for (int w = 1; w < kBlock / 64; ++w) {
        out.mn = fmin_(out.mn, wave_res[w].mn);
        out.mx = fmax_(out.mx, wave_res[w].mx);
        out.nan |= wave_res[w].nan;
    }
    return out;
}

// ------------------------------------------------------------------------------------------------
// per-tensor
// ------------------------------------------------------------------------------------------------
template <typename IO, int UNROLL>
__global__ __launch_bounds__(kBlock) void minmax_pt_kernel(const void* __restrict__ x, int64_t n,
                                                           MinMaxPartial<typename IO::arith>* __restrict__ partials) {
    using T = typename IO::arith;
    constexpr int VEC = IO::VEC;
    RunningMinMax<T> r;
    const int64_t n_packets = n / VEC;
    constexpr int64_t kTile = static_cast<int64_t>(kBlock) * UNROLL;
    const int64_t n_full = n_packets / kTile;
    for (int64_t tile = blockIdx.x; tile < n_full; tile += gridDim.x) {
        const int64_t p0 = tile * kTile + threadIdx.x;
        Packet<IO> in[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) in[u] = load_packet_nt<IO>(x, (p0 + static_cast<int64_t>(u) * kBlock) * VEC);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int j = 0; j < VEC; ++j) r.push(static_cast<T>(in[u].v[j]));
    }
    if (static_cast<int64_t>(blockIdx.x) == n_full % gridDim.x) {
        for (int64_t p = n_full * kTile + threadIdx.x; p < n_packets; p += kBlock) {
            const Packet<IO> in = load_packet<IO>(x, p * VEC);
#pragma unroll
            for (int j = 0; j < VEC; ++j) r.push(static_cast<T>(in.v[j]));
        }
    }
    if (blockIdx.x == 0) {
        const int64_t i = n_packets * VEC + threadIdx.x;
        if (i < n) r.push(IO::load1(x, i));
    }
    const MinMaxPartial<T> res = block_minmax<T>(r);
    if (threadIdx.x == 0) partials[blockIdx.x] = res;
}

template <typename IO>
__global__ __launch_bounds__(kBlock) void minmax_pt_scalar_kernel(const void* __restrict__ x, int64_t n,
                                                                  MinMaxPartial<typename IO::arith>* __restrict__ partials) {
    using T = typename IO::arith;
    RunningMinMax<T> r;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * kBlock)
        r.push(IO::load1(x, i));
    const MinMaxPartial<T> res = block_minmax<T>(r);
    if (threadIdx.x == 0) partials[blockIdx.x] = res;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void minmax_pt_finalize_kernel(const MinMaxPartial<T>* __restrict__ partials,
                                                                    int n_partials, T* __restrict__ out_min,
                                                                    T* __restrict__ out_max) {
    RunningMinMax<T> r;
    for (int i = threadIdx.x; i < n_partials; i += kBlock) {
        const MinMaxPartial<T> p = partials[i];
        r.mn = fmin_(r.mn, p.mn);
        r.mx = fmax_(r.mx, p.mx);
        r.nan = r.nan || (p.nan != 0);
    }
    const MinMaxPartial<T> res = block_minmax<T>(r);
    if (threadIdx.x == 0) {
        const T qnan = inf_of<T>::value() - inf_of<T>::value();
        out_min[0] = res.nan ? qnan : res.mn;
        out_max[0] = res.nan ? qnan : res.mx;
    }
}

// ------------------------------------------------------------------------------------------------
// per-channel (window mode of lsq_pc_geom.hpp; every component of a lane keeps its own channel)
// ------------------------------------------------------------------------------------------------
template <typename IO, int V, int UNROLL>
__global__ __launch_bounds__(kBlock) void minmax_pc_kernel(const void* __restrict__ x, PcGeom g,
                                                           MinMaxPartial<typename IO::arith>* __restrict__ partials) {
    using T = typename IO::arith;
    using E = typename IO::elem;
    using K = typename key_of<T>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    K* key_min = reinterpret_cast<K*>(smem);
    K* key_max = key_min + g.k_slots;
    uint32_t* nan_flag = reinterpret_cast<uint32_t*>(key_max + g.k_slots);

    const LaneSite site = lane_site(g, V);
    const RowWalk walk(g, site);
    E first[UNROLL][V];                       // in flight while LDS is initialised
    const bool first_full = walk.n_rows >= UNROLL;
    if (first_full) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) load_elems<IO, V, true>(x, walk.row(u) * g.L + site.p0, first[u]);
    }
    for (int k = threadIdx.x; k < g.k_slots; k += kBlock) {
        key_min[k] = ~static_cast<K>(0);
        key_max[k] = static_cast<K>(0);
        nan_flag[k] = 0u;
    }
    int32_t slot[V];
#pragma unroll
    for (int j = 0; j < V; ++j)
        slot[j] = site.live ? static_cast<int32_t>(udiv(site.p0 + j, g.inner, g.fits32 != 0) - site.c_lo) : 0;
    __syncthreads();

    RunningMinMax<T> r[V];
    auto absorb = [&](const E (&in)[V], bool valid) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const T v = static_cast<T>(in[j]);
            if (valid) r[j].push(v);
        }
    };
    // rows = full groups of UNROLL + one group of UNROLL/2 + ... + one single row (no padded slots, see lsq_per_channel.hip)
    auto group = [&](int64_t i0, auto width) {
        constexpr int H = decltype(width)::value;
        E in[H][V];
#pragma unroll
        for (int u = 0; u < H; ++u) load_elems<IO, V, true>(x, walk.row(i0 + u) * g.L + site.p0, in[u]);
#pragma unroll
        for (int u = 0; u < H; ++u) absorb(in[u], true);
    };
    int64_t i = 0;
    if (first_full) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) absorb(first[u], true);
        i = UNROLL;
    }
    for (; i + UNROLL <= walk.n_rows; i += UNROLL) group(i, std::integral_constant<int, UNROLL>{});
    if constexpr (UNROLL >= 8) if (i + 4 <= walk.n_rows) { group(i, std::integral_constant<int, 4>{}); i += 4; }
    if constexpr (UNROLL >= 4) if (i + 2 <= walk.n_rows) { group(i, std::integral_constant<int, 2>{}); i += 2; }
    if constexpr (UNROLL >= 2) if (i < walk.n_rows) group(i, std::integral_constant<int, 1>{});
    if (site.live && walk.n_rows > 0) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
            atomicMin(&key_min[slot[j]], order_key(r[j].mn));
            atomicMax(&key_max[slot[j]], order_key(r[j].mx));
            if (r[j].nan) atomicOr(&nan_flag[slot[j]], 1u);
        }
    }
    __syncthreads();
    const int64_t block_linear = static_cast<int64_t>(blockIdx.y) * g.n_windows + blockIdx.x;
    MinMaxPartial<T>* out = partials + block_linear * g.k_slots;
    for (int k = threadIdx.x; k < g.k_slots; k += kBlock)
        out[k] = MinMaxPartial<T>{from_key(key_min[k]), from_key(key_max[k]), static_cast<int32_t>(nan_flag[k]), 0};
}

constexpr int kMmFinCh = 32;
constexpr int kMmFinParts = kBlock / kMmFinCh;

// kMmFinCh channels x kMmFinParts interleaved slices of the split axis per workgroup: independent loads
// instead of one serial chain per channel; slices combined through LDS.
template <typename T>
__global__ __launch_bounds__(kBlock) void minmax_pc_finalize_kernel(const MinMaxPartial<T>* __restrict__ partials,
                                                                    PcGeom g, T* __restrict__ out_min,
                                                                    T* __restrict__ out_max) {
    __shared__ MinMaxPartial<T> part_res[kMmFinParts][kMmFinCh];
    const int lane_c = threadIdx.x % kMmFinCh, part = threadIdx.x / kMmFinCh;
    const int64_t c = static_cast<int64_t>(blockIdx.x) * kMmFinCh + lane_c;
    RunningMinMax<T> r;
    if (c < g.C) {
        int64_t w_lo = 0, w_hi = 0;
        if (g.R == 1) {
            w_lo = (c * g.inner) / g.wpos;
            w_hi = ((c + 1) * g.inner - 1) / g.wpos;
        }
        for (int64_t w = w_lo; w <= w_hi; ++w) {
            const int64_t c_lo = (g.R == 1) ? (w * g.wpos) / g.inner : 0;
            const MinMaxPartial<T>* col = partials + w * g.k_slots + (c - c_lo);
            const int64_t stride = g.n_windows * g.k_slots;
#pragma unroll 4
            for (int32_t sy = part; sy < g.splits; sy += kMmFinParts) {
                const MinMaxPartial<T> p = col[static_cast<int64_t>(sy) * stride];
                r.mn = fmin_(r.mn, p.mn);
                r.mx = fmax_(r.mx, p.mx);
                r.nan = r.nan || (p.nan != 0);
            }
        }
    }
    part_res[part][lane_c] = MinMaxPartial<T>{r.mn, r.mx, r.nan ? 1 : 0, 0};
    __syncthreads();
    if (part == 0 && c < g.C) {
        MinMaxPartial<T> t = part_res[0][lane_c];
#pragma unroll
        for (int k = 1; k < kMmFinParts; ++k) {
            t.mn = fmin_(t.mn, part_res[k][lane_c].mn);
            t.mx = fmax_(t.mx, part_res[k][lane_c].mx);
            t.nan |= part_res[k][lane_c].nan;
        }
        const T qnan = inf_of<T>::value() - inf_of<T>::value();
        out_min[c] = t.nan ? qnan : t.mn;
        out_max[c] = t.nan ? qnan : t.mx;
    }
}

// ------------------------------------------------------------------------------------------------
// per-channel, segment mode (few rows, long channels: weights on axis 0): one channel per workgroup
// ------------------------------------------------------------------------------------------------
template <typename IO, int V, int UNROLL>
__global__ __launch_bounds__(kBlock) void minmax_seg_kernel(const void* __restrict__ x, SegGeom g,
                                                            MinMaxPartial<typename IO::arith>* __restrict__ partials) {
    using T = typename IO::arith;
    using E = typename IO::elem;
    const SegWalk w(g);
    const int64_t W = static_cast<int64_t>(kBlock) * V;
    const int64_t q0 = static_cast<int64_t>(threadIdx.x) * V;
    const int64_t q_last = g.inner - V;
    RunningMinMax<T> r;
    auto site = [&](int64_t it, bool& valid) {
        const int64_t oi = static_cast<int64_t>(static_cast<uint32_t>(it) / static_cast<uint32_t>(w.n_r));
        const int64_t ri = it - oi * w.n_r;
        const int64_t pos = (w.r_begin + ri) * W + q0;
        valid = pos < g.inner;
        return ((w.o_begin + oi) * g.C + w.c) * g.inner + (valid ? pos : q_last);
    };
    auto group = [&](int64_t it, auto width) {     // groups of UNROLL, then UNROLL/2, ..., 1: no padded slots
        constexpr int H = decltype(width)::value;
        E in[H][V];
        bool ok[H];
#pragma unroll
        for (int u = 0; u < H; ++u) load_elems<IO, V, true>(x, site(it + u, ok[u]), in[u]);
#pragma unroll
        for (int u = 0; u < H; ++u)
#pragma unroll
            for (int j = 0; j < V; ++j)
                if (ok[u]) r.push(static_cast<T>(in[u][j]));
    };
    int64_t it = 0;
    for (; it + UNROLL <= w.n_it; it += UNROLL) group(it, std::integral_constant<int, UNROLL>{});
    if constexpr (UNROLL >= 8) if (it + 4 <= w.n_it) { group(it, std::integral_constant<int, 4>{}); it += 4; }
    if constexpr (UNROLL >= 4) if (it + 2 <= w.n_it) { group(it, std::integral_constant<int, 2>{}); it += 2; }
    if constexpr (UNROLL >= 2) if (it < w.n_it) group(it, std::integral_constant<int, 1>{});
    const MinMaxPartial<T> res = block_minmax<T>(r);
    if (threadIdx.x == 0) partials[static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x] = res;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void minmax_seg_finalize_kernel(const MinMaxPartial<T>* __restrict__ partials,
                                                                     SegGeom g, T* __restrict__ out_min,
                                                                     T* __restrict__ out_max) {
    __shared__ MinMaxPartial<T> part_res[kMmFinParts][kMmFinCh];
    const int lane_c = threadIdx.x % kMmFinCh, part = threadIdx.x / kMmFinCh;
    const int64_t c = static_cast<int64_t>(blockIdx.x) * kMmFinCh + lane_c;
    RunningMinMax<T> r;
    if (c < g.C) {
        const int64_t gx = g.C * g.segs;
        const int32_t total = g.osplits * g.segs;
#pragma unroll 4
        for (int32_t sl = part; sl < total; sl += kMmFinParts) {
            const int32_t oy = sl / g.segs, sg = sl - oy * g.segs;
            const MinMaxPartial<T> p = partials[static_cast<int64_t>(oy) * gx + c * g.segs + sg];
            r.mn = fmin_(r.mn, p.mn);
            r.mx = fmax_(r.mx, p.mx);
            r.nan = r.nan || (p.nan != 0);
        }
    }
    part_res[part][lane_c] = MinMaxPartial<T>{r.mn, r.mx, r.nan ? 1 : 0, 0};
    __syncthreads();
    if (part == 0 && c < g.C) {
        MinMaxPartial<T> t = part_res[0][lane_c];
#pragma unroll
        for (int k = 1; k < kMmFinParts; ++k) {
            t.mn = fmin_(t.mn, part_res[k][lane_c].mn);
            t.mx = fmax_(t.mx, part_res[k][lane_c].mx);
            t.nan |= part_res[k][lane_c].nan;
        }
        const T qnan = inf_of<T>::value() - inf_of<T>::value();
        out_min[c] = t.nan ? qnan : t.mn;
        out_max[c] = t.nan ? qnan : t.mx;
    }
}

// =================================================================================================
// mean / standard deviation (the 3-sigma initialisation of weight quantizers)
// =================================================================================================
// The reference module creates the scale of a weight quantizer from the weight itself on its first call:
// scale = max(|mu - 3 sigma|, |mu + 3 sigma|) / 2^bits with mu / sigma = torch.mean / torch.std (unbiased) over
// everything, or per channel over the other axes (/root/reference/torchlsq/quantized/modules/observers.py:329-337):
// two library reductions, each several passes for the per-channel case.  Here: ONE read-only pass, same walk as
// the min/max kernels.  Numerics: shifted-data sums in fp64 -- S1 = sum(x - K), S2 = sum((x - K)^2) with the pivot
// K = the channel's first element (a sample of the data, so |K - mu| is O(sigma) and S2 - S1^2/n does not cancel);
// mean = K + S1/n, var = (S2 - S1^2/n) / (n - 1).  A non-finite pivot is replaced by 0 so that inf / NaN
// propagate the way they do through torch (mean inf or NaN, std NaN); n = 1 gives std = NaN like torch.
template <typename T>
__device__ __forceinline__ double pivot_of(T v) {
    const double d = static_cast<double>(v);
    return (d - d == 0.0) ? d : 0.0;   // inf - inf and NaN - NaN are NaN
}

struct RunningMoments {
    double s1 = 0.0, s2 = 0.0;
    __device__ __forceinline__ void push(double v, double pivot) {
        const double d = v - pivot;
        s1 += d;
        s2 = __builtin_fma(d, d, s2);
    }
};

// wave64 butterfly, then the 4 wave results through LDS in a fixed order; thread 0 returns the workgroup's sums
__device__ __forceinline__ double2 block_sum2(double a, double b) {
    __shared__ double2 wave_res[kBlock / 64];
    a = wave_sum(a);
    b = wave_sum(b);
    if ((threadIdx.x & 63) == 0) wave_res[threadIdx.x >> 6] = make_double2(a, b);
    __syncthreads();
    double2 out = wave_res[0];
#pragma unroll
    for (int w = 1; w < kBlock / 64; ++w) { out.x += wave_res[w].x; out.y += wave_res[w].y; }
    return out;
}

template <typename T>
__device__ __forceinline__ void write_mean_std(double pivot, double s1, double s2, double n, T* mean_out, T* std_out, int64_t c) {
    const double var = (s2 - s1 * s1 / n) / (n - 1.0);          // n == 1: 0/0 = NaN, like torch.std
    mean_out[c] = static_cast<T>(pivot + s1 / n);
    std_out[c] = static_cast<T>(__builtin_sqrt(var < 0.0 ? 0.0 : var));   // NaN stays NaN
}

template <typename IO, int UNROLL>
__global__ __launch_bounds__(kBlock) void moments_pt_kernel(const void* __restrict__ x, int64_t n, double2* __restrict__ partials) {
    using T = typename IO::arith;
    constexpr int VEC = IO::VEC;
    const double pivot = pivot_of<T>(IO::load1(x, 0));
    RunningMoments r;
    const int64_t n_packets = n / VEC;
    constexpr int64_t kTile = static_cast<int64_t>(kBlock) * UNROLL;
    const int64_t n_full = n_packets / kTile;
    for (int64_t tile = blockIdx.x; tile < n_full; tile += gridDim.x) {
        const int64_t p0 = tile * kTile + threadIdx.x;
        Packet<IO> in[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) in[u] = load_packet_nt<IO>(x, (p0 + static_cast<int64_t>(u) * kBlock) * VEC);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int j = 0; j < VEC; ++j) r.push(static_cast<double>(static_cast<T>(in[u].v[j])), pivot);
    }
    if (static_cast<int64_t>(blockIdx.x) == n_full % gridDim.x) {
        for (int64_t p = n_full * kTile + threadIdx.x; p < n_packets; p += kBlock) {
            const Packet<IO> in = load_packet<IO>(x, p * VEC);
#pragma unroll
            for (int j = 0; j < VEC; ++j) r.push(static_cast<double>(static_cast<T>(in.v[j])), pivot);
        }
    }
    if (blockIdx.x == 0) {
        const int64_t i = n_packets * VEC + threadIdx.x;
        if (i < n) r.push(static_cast<double>(IO::load1(x, i)), pivot);
    }
    const double2 res = block_sum2(r.s1, r.s2);
    if (threadIdx.x == 0) partials[blockIdx.x] = res;
}

template <typename IO>
__global__ __launch_bounds__(kBlock) void moments_pt_scalar_kernel(const void* __restrict__ x, int64_t n, double2* __restrict__ partials) {
    using T = typename IO::arith;
    const double pivot = pivot_of<T>(IO::load1(x, 0));
    RunningMoments r;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kBlock)
        r.push(static_cast<double>(IO::load1(x, i)), pivot);
    const double2 res = block_sum2(r.s1, r.s2);
    if (threadIdx.x == 0) partials[blockIdx.x] = res;
}

template <typename IO>
__global__ __launch_bounds__(kBlock) void moments_pt_finalize_kernel(const double2* __restrict__ partials, int n_partials,
                                                                     const void* __restrict__ x, int64_t n,
                                                                     typename IO::arith* __restrict__ mean_out,
                                                                     typename IO::arith* __restrict__ std_out) {
    using T = typename IO::arith;
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < n_partials; i += kBlock) { a += partials[i].x; b += partials[i].y; }
    const double2 t = block_sum2(a, b);
    if (threadIdx.x == 0) write_mean_std<T>(pivot_of<T>(IO::load1(x, 0)), t.x, t.y, static_cast<double>(n), mean_out, std_out, 0);
}

// per-channel, window mode: every component of a lane keeps its own channel, pivot and pair of sums
template <typename IO, int V, int UNROLL>
__global__ __launch_bounds__(kBlock) void moments_pc_kernel(const void* __restrict__ x, PcGeom g, double2* __restrict__ partials) {
    using T = typename IO::arith;
    using E = typename IO::elem;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* lds_pivot = reinterpret_cast<double*>(smem);
    double* lds_s1 = lds_pivot + g.k_slots;
    double* lds_s2 = lds_s1 + g.k_slots;

    const LaneSite site = lane_site(g, V);
    const RowWalk walk(g, site);
    E first[UNROLL][V];
    const bool first_full = walk.n_rows >= UNROLL;
    if (first_full) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) load_elems<IO, V, true>(x, walk.row(u) * g.L + site.p0, first[u]);
    }
    for (int k = threadIdx.x; k < g.k_slots; k += kBlock) {
        const int64_t c = site.c_lo + k;
        lds_pivot[k] = c < g.C ? pivot_of<T>(IO::load1(x, c * g.inner)) : 0.0;   // row 0, first element of the channel
        lds_s1[k] = 0.0;
        lds_s2[k] = 0.0;
    }
    int32_t slot[V];
#pragma unroll
    for (int j = 0; j < V; ++j)
        slot[j] = site.live ? static_cast<int32_t>(udiv(site.p0 + j, g.inner, g.fits32 != 0) - site.c_lo) : 0;
    __syncthreads();
    double pivot[V];
#pragma unroll
    for (int j = 0; j < V; ++j) pivot[j] = lds_pivot[slot[j]];

    RunningMoments r[V];
    auto absorb = [&](const E (&in)[V], bool valid) {
#pragma unroll
        for (int j = 0; j < V; ++j)
            if (valid) r[j].push(static_cast<double>(static_cast<T>(in[j])), pivot[j]);
    };
    // rows = full groups of UNROLL + one group of UNROLL/2 + ... + one single row (no padded slots, see lsq_per_channel.hip)
    auto group = [&](int64_t i0, auto width) {
        constexpr int H = decltype(width)::value;
        E in[H][V];
#pragma unroll
        for (int u = 0; u < H; ++u) load_elems<IO, V, true>(x, walk.row(i0 + u) * g.L + site.p0, in[u]);
#pragma unroll
        for (int u = 0; u < H; ++u) absorb(in[u], true);
    };
    int64_t i = 0;
    if (first_full) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) absorb(first[u], true);
        i = UNROLL;
    }
    for (; i + UNROLL <= walk.n_rows; i += UNROLL) group(i, std::integral_constant<int, UNROLL>{});
    if constexpr (UNROLL >= 8) if (i + 4 <= walk.n_rows) { group(i, std::integral_constant<int, 4>{}); i += 4; }
    if constexpr (UNROLL >= 4) if (i + 2 <= walk.n_rows) { group(i, std::integral_constant<int, 2>{}); i += 2; }
    if constexpr (UNROLL >= 2) if (i < walk.n_rows) group(i, std::integral_constant<int, 1>{});
    if (site.live && walk.n_rows > 0) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
            __hip_atomic_fetch_add(&lds_s1[slot[j]], r[j].s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&lds_s2[slot[j]], r[j].s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    const int64_t block_linear = static_cast<int64_t>(blockIdx.y) * g.n_windows + blockIdx.x;
    double2* out = partials + block_linear * g.k_slots;
    for (int k = threadIdx.x; k < g.k_slots; k += kBlock) out[k] = make_double2(lds_s1[k], lds_s2[k]);
}

template <typename IO>
__global__ __launch_bounds__(kBlock) void moments_pc_finalize_kernel(const double2* __restrict__ partials, PcGeom g,
                                                                     const void* __restrict__ x,
                                                                     typename IO::arith* __restrict__ mean_out,
                                                                     typename IO::arith* __restrict__ std_out) {
    using T = typename IO::arith;
    __shared__ double2 part_res[kMmFinParts][kMmFinCh];
    const int lane_c = threadIdx.x % kMmFinCh, part = threadIdx.x / kMmFinCh;
    const int64_t c = static_cast<int64_t>(blockIdx.x) * kMmFinCh + lane_c;
    double a = 0.0, b = 0.0;
    if (c < g.C) {
        int64_t w_lo = 0, w_hi = 0;
        if (g.R == 1) {
            w_lo = (c * g.inner) / g.wpos;
            w_hi = ((c + 1) * g.inner - 1) / g.wpos;
        }
        for (int64_t w = w_lo; w <= w_hi; ++w) {
            const int64_t c_lo = (g.R == 1) ? (w * g.wpos) / g.inner : 0;
            const double2* col = partials + w * g.k_slots + (c - c_lo);
            const int64_t stride = g.n_windows * g.k_slots;
#pragma unroll 4
            for (int32_t sy = part; sy < g.splits; sy += kMmFinParts) {
                const double2 p = col[static_cast<int64_t>(sy) * stride];
                a += p.x;
                b += p.y;
            }
        }
    }
    part_res[part][lane_c] = make_double2(a, b);
    __syncthreads();
    if (part == 0 && c < g.C) {
        double2 t = part_res[0][lane_c];
#pragma unroll
        for (int k = 1; k < kMmFinParts; ++k) { t.x += part_res[k][lane_c].x; t.y += part_res[k][lane_c].y; }
        write_mean_std<T>(pivot_of<T>(IO::load1(x, c * g.inner)), t.x, t.y, static_cast<double>(g.outer) * static_cast<double>(g.inner),
                          mean_out, std_out, c);
    }
}

// per-channel, segment mode: one channel per workgroup
template <typename IO, int V, int UNROLL>
__global__ __launch_bounds__(kBlock) void moments_seg_kernel(const void* __restrict__ x, SegGeom g, double2* __restrict__ partials) {
    using T = typename IO::arith;
    using E = typename IO::elem;
    const SegWalk w(g);
    const int64_t W = static_cast<int64_t>(kBlock) * V;
    const int64_t q0 = static_cast<int64_t>(threadIdx.x) * V;
    const int64_t q_last = g.inner - V;
    const double pivot = pivot_of<T>(IO::load1(x, w.c * g.inner));
    RunningMoments r;
    auto site = [&](int64_t it, bool& valid) {
        const int64_t oi = static_cast<int64_t>(static_cast<uint32_t>(it) / static_cast<uint32_t>(w.n_r));
        const int64_t ri = it - oi * w.n_r;
        const int64_t pos = (w.r_begin + ri) * W + q0;
        valid = pos < g.inner;
        return ((w.o_begin + oi) * g.C + w.c) * g.inner + (valid ? pos : q_last);
    };
    auto group = [&](int64_t it, auto width) {     // groups of UNROLL, then UNROLL/2, ..., 1: no padded slots
        constexpr int H = decltype(width)::value;
        E in[H][V];
        bool ok[H];
#pragma unroll
        for (int u = 0; u < H; ++u) load_elems<IO, V, true>(x, site(it + u, ok[u]), in[u]);
#pragma unroll
        for (int u = 0; u < H; ++u)
#pragma unroll
            for (int j = 0; j < V; ++j)
                if (ok[u]) r.push(static_cast<double>(static_cast<T>(in[u][j])), pivot);
    };
    int64_t it = 0;
    for (; it + UNROLL <= w.n_it; it += UNROLL) group(it, std::integral_constant<int, UNROLL>{});
    if constexpr (UNROLL >= 8) if (it + 4 <= w.n_it) { group(it, std::integral_constant<int, 4>{}); it += 4; }
    if constexpr (UNROLL >= 4) if (it + 2 <= w.n_it) { group(it, std::integral_constant<int, 2>{}); it += 2; }
    if constexpr (UNROLL >= 2) if (it < w.n_it) group(it, std::integral_constant<int, 1>{});
    const double2 res = block_sum2(r.s1, r.s2);
    if (threadIdx.x == 0) partials[static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x] = res;
}

template <typename IO>
__global__ __launch_bounds__(kBlock) void moments_seg_finalize_kernel(const double2* __restrict__ partials, SegGeom g,
                                                                      const void* __restrict__ x,
                                                                      typename IO::arith* __restrict__ mean_out,
                                                                      typename IO::arith* __restrict__ std_out) {
    using T = typename IO::arith;
    __shared__ double2 part_res[kMmFinParts][kMmFinCh];
    const int lane_c = threadIdx.x % kMmFinCh, part = threadIdx.x / kMmFinCh;
    const int64_t c = static_cast<int64_t>(blockIdx.x) * kMmFinCh + lane_c;
    double a = 0.0, b = 0.0;
    if (c < g.C) {
        const int64_t gx = g.C * g.segs;
        const int32_t total = g.osplits * g.segs;
#pragma unroll 4
        for (int32_t sl = part; sl < total; sl += kMmFinParts) {
            const int32_t oy = sl / g.segs, sg = sl - oy * g.segs;
            const double2 p = partials[static_cast<int64_t>(oy) * gx + c * g.segs + sg];
            a += p.x;
            b += p.y;
        }
    }
    part_res[part][lane_c] = make_double2(a, b);
    __syncthreads();
    if (part == 0 && c < g.C) {
        double2 t = part_res[0][lane_c];
#pragma unroll
        for (int k = 1; k < kMmFinParts; ++k) { t.x += part_res[k][lane_c].x; t.y += part_res[k][lane_c].y; }
        write_mean_std<T>(pivot_of<T>(IO::load1(x, c * g.inner)), t.x, t.y, static_cast<double>(g.outer) * static_cast<double>(g.inner),
                          mean_out, std_out, c);
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
constexpr int kObserveUnroll = 4;
// workgroups per CU of the statistics kernels (the tools build can override it: knob::kObserveWgPerCu, 0 in production)
static inline int observe_wg_per_cu(int dflt) {
    const int v = knob::get(knob::kObserveWgPerCu);
    return v > 0 ? std::min(v, kMaxBlocksPerCU) : dflt;
}
constexpr int kObserveWgPerTensor = 8;
constexpr int kObserveWgWindow = 4;
constexpr int kObserveWgSegment = 16;

static inline PcGeom observe_geom(int64_t outer, int64_t C, int64_t inner, int vec, int cu_count) {
    // few, fat workgroups along the rows when rows are plentiful; windows give the parallelism otherwise
    return make_geom(outer, C, inner, vec, cu_count * observe_wg_per_cu(kObserveWgWindow));
}

size_t minmax_workspace_bytes(int io_vec, int elem_arith_bytes, int64_t outer, int64_t channels, int64_t inner) {
    const size_t psz = elem_arith_bytes == 8 ? sizeof(MinMaxPartial<double>) : sizeof(MinMaxPartial<float>);
    size_t need = static_cast<size_t>(kMaxCUs) * kMaxBlocksPerCU * psz;   // per-tensor
    const DeviceInfo& dev = device_info();
    const int vecs[2] = {io_vec, 1};
    for (int vi = 0; vi < 2; ++vi) {
        for (int wg = 1; wg <= kMaxBlocksPerCU; ++wg) {
            const PcGeom g = make_geom(outer, channels, inner, vecs[vi], dev.cu_count * wg);
            need = std::max(need, static_cast<size_t>(g.splits) * g.n_windows * g.k_slots * psz);
            if (pick_segment_mode(vecs[vi], outer, channels, inner)) {
                const SegGeom sg = make_seg_geom(outer, channels, inner, vecs[vi], dev.cu_count * wg);
                need = std::max(need, static_cast<size_t>(channels) * sg.segs * sg.osplits * psz);
            }
        }
    }
    return need + 256;
}

template <typename IO>
hipError_t minmax_per_tensor(const void* x, int64_t n, void* out_min, void* out_max, void* workspace,
                             hipStream_t stream) {
    using T = typename IO::arith;
    const DeviceInfo& dev = device_info();
    auto* partials = static_cast<MinMaxPartial<T>*>(workspace);
    int grid;
    if (!is_elem_aligned<IO>(x)) {      // (packets take any element-aligned view: lsq_math.hpp, PacketWord)
        const int64_t want = std::max<int64_t>(1, (n + kBlock - 1) / kBlock);
        grid = static_cast<int>(std::min<int64_t>(want, static_cast<int64_t>(dev.cu_count) * observe_wg_per_cu(kObserveWgPerTensor)));
        hipLaunchKernelGGL((minmax_pt_scalar_kernel<IO>), dim3(grid), dim3(kBlock), 0, stream, x, n, partials);
    } else {
        const int64_t tile = static_cast<int64_t>(kBlock) * kObserveUnroll;
        const int64_t n_tiles = std::max<int64_t>(1, (n / IO::VEC + tile - 1) / tile);
        grid = static_cast<int>(std::min<int64_t>(n_tiles, static_cast<int64_t>(dev.cu_count) * observe_wg_per_cu(kObserveWgPerTensor)));
        hipLaunchKernelGGL((minmax_pt_kernel<IO, kObserveUnroll>), dim3(grid), dim3(kBlock), 0, stream, x, n, partials);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((minmax_pt_finalize_kernel<T>), dim3(1), dim3(kBlock), 0, stream, partials, grid,
                       static_cast<T*>(out_min), static_cast<T*>(out_max));
    return hipGetLastError();
}

template <typename IO>
hipError_t minmax_per_channel(const void* x, int64_t outer, int64_t channels, int64_t inner, void* out_min,
                              void* out_max, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    using T = typename IO::arith;
    using K = typename key_of<T>::type;
    const DeviceInfo& dev = device_info();
    const int vec = pick_vec(IO::VEC, channels * inner, is_elem_aligned<IO>(x));
    const unsigned fgrid_c = static_cast<unsigned>((channels + kMmFinCh - 1) / kMmFinCh);
    if (pick_segment_mode(vec, outer, channels, inner)) {
        const SegGeom sg = make_seg_geom(outer, channels, inner, vec, dev.cu_count * observe_wg_per_cu(kObserveWgSegment));
        if (!grid_fits(sg)) return hipErrorInvalidConfiguration;
        if (workspace_bytes < static_cast<size_t>(channels) * sg.segs * sg.osplits * sizeof(MinMaxPartial<T>))
            return hipErrorInvalidValue;
        auto* sp = static_cast<MinMaxPartial<T>*>(workspace);
        hipLaunchKernelGGL((minmax_seg_kernel<IO, IO::VEC, kObserveUnroll>),
                           dim3(static_cast<unsigned>(sg.C * sg.segs), static_cast<unsigned>(sg.osplits)), dim3(kBlock), 0,
                           stream, x, sg, sp);
        hipError_t es = hipGetLastError();
        if (es != hipSuccess) return es;
        hipLaunchKernelGGL((minmax_seg_finalize_kernel<T>), dim3(fgrid_c), dim3(kBlock), 0, stream, sp, sg,
                           static_cast<T*>(out_min), static_cast<T*>(out_max));
        return hipGetLastError();
    }
    const PcGeom g = observe_geom(outer, channels, inner, vec, dev.cu_count);
    if (!grid_fits(g)) return hipErrorInvalidConfiguration;
    if (workspace_bytes < static_cast<size_t>(g.splits) * g.n_windows * g.k_slots * sizeof(MinMaxPartial<T>))
        return hipErrorInvalidValue;
    auto* partials = static_cast<MinMaxPartial<T>*>(workspace);
    const dim3 grid(static_cast<unsigned>(g.n_windows), static_cast<unsigned>(g.splits));
    const size_t lds = static_cast<size_t>(g.k_slots) * (2 * sizeof(K) + sizeof(uint32_t));
    if (vec == 1)
        hipLaunchKernelGGL((minmax_pc_kernel<IO, 1, kObserveUnroll>), grid, dim3(kBlock), lds, stream, x, g, partials);
    else
        hipLaunchKernelGGL((minmax_pc_kernel<IO, IO::VEC, kObserveUnroll>), grid, dim3(kBlock), lds, stream, x, g, partials);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const unsigned fgrid = static_cast<unsigned>((channels + kMmFinCh - 1) / kMmFinCh);
    hipLaunchKernelGGL((minmax_pc_finalize_kernel<T>), dim3(fgrid), dim3(kBlock), 0, stream, partials, g,
                       static_cast<T*>(out_min), static_cast<T*>(out_max));
    return hipGetLastError();
}

size_t meanstd_workspace_bytes(int io_vec, int64_t outer, int64_t channels, int64_t inner) {
    // same launch geometry as the min/max kernels, 16-byte {S1, S2} partials
    return minmax_workspace_bytes(io_vec, 8, outer, channels, inner);   // sizeof(MinMaxPartial<double>) = 24 >= 16
}

template <typename IO>
hipError_t meanstd_per_tensor(const void* x, int64_t n, void* out_mean, void* out_std, void* workspace, hipStream_t stream) {
    using T = typename IO::arith;
    const DeviceInfo& dev = device_info();
    auto* partials = static_cast<double2*>(workspace);
    int grid;
    if (!is_elem_aligned<IO>(x)) {      // (packets take any element-aligned view: lsq_math.hpp, PacketWord)
        const int64_t want = std::max<int64_t>(1, (n + kBlock - 1) / kBlock);
        grid = static_cast<int>(std::min<int64_t>(want, static_cast<int64_t>(dev.cu_count) * observe_wg_per_cu(kObserveWgPerTensor)));
        hipLaunchKernelGGL((moments_pt_scalar_kernel<IO>), dim3(grid), dim3(kBlock), 0, stream, x, n, partials);
    } else {
        const int64_t tile = static_cast<int64_t>(kBlock) * kObserveUnroll;
        const int64_t n_tiles = std::max<int64_t>(1, (n / IO::VEC + tile - 1) / tile);
        grid = static_cast<int>(std::min<int64_t>(n_tiles, static_cast<int64_t>(dev.cu_count) * observe_wg_per_cu(kObserveWgPerTensor)));
        hipLaunchKernelGGL((moments_pt_kernel<IO, kObserveUnroll>), dim3(grid), dim3(kBlock), 0, stream, x, n, partials);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((moments_pt_finalize_kernel<IO>), dim3(1), dim3(kBlock), 0, stream, partials, grid, x, n,
                       static_cast<T*>(out_mean), static_cast<T*>(out_std));
    return hipGetLastError();
}

template <typename IO>
hipError_t meanstd_per_channel(const void* x, int64_t outer, int64_t channels, int64_t inner, void* out_mean,
                               void* out_std, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    using T = typename IO::arith;
    const DeviceInfo& dev = device_info();
    const int vec = pick_vec(IO::VEC, channels * inner, is_elem_aligned<IO>(x));
    const unsigned fgrid = static_cast<unsigned>((channels + kMmFinCh - 1) / kMmFinCh);
    auto* partials = static_cast<double2*>(workspace);
    if (pick_segment_mode(vec, outer, channels, inner)) {
        const SegGeom sg = make_seg_geom(outer, channels, inner, vec, dev.cu_count * observe_wg_per_cu(kObserveWgSegment));
        if (!grid_fits(sg)) return hipErrorInvalidConfiguration;
        if (workspace_bytes < static_cast<size_t>(channels) * sg.segs * sg.osplits * sizeof(double2)) return hipErrorInvalidValue;
        hipLaunchKernelGGL((moments_seg_kernel<IO, IO::VEC, kObserveUnroll>),
                           dim3(static_cast<unsigned>(sg.C * sg.segs), static_cast<unsigned>(sg.osplits)), dim3(kBlock), 0,
                           stream, x, sg, partials);
        hipError_t es = hipGetLastError();
        if (es != hipSuccess) return es;
        hipLaunchKernelGGL((moments_seg_finalize_kernel<IO>), dim3(fgrid), dim3(kBlock), 0, stream, partials, sg, x,
                           static_cast<T*>(out_mean), static_cast<T*>(out_std));
        return hipGetLastError();
    }
    const PcGeom g = observe_geom(outer, channels, inner, vec, dev.cu_count);
    if (!grid_fits(g)) return hipErrorInvalidConfiguration;
    if (workspace_bytes < static_cast<size_t>(g.splits) * g.n_windows * g.k_slots * sizeof(double2)) return hipErrorInvalidValue;
    const dim3 grid(static_cast<unsigned>(g.n_windows), static_cast<unsigned>(g.splits));
    const size_t lds = static_cast<size_t>(g.k_slots) * 3 * sizeof(double);
    if (vec == 1)
        hipLaunchKernelGGL((moments_pc_kernel<IO, 1, kObserveUnroll>), grid, dim3(kBlock), lds, stream, x, g, partials);
    else
        hipLaunchKernelGGL((moments_pc_kernel<IO, IO::VEC, kObserveUnroll>), grid, dim3(kBlock), lds, stream, x, g, partials);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((moments_pc_finalize_kernel<IO>), dim3(fgrid), dim3(kBlock), 0, stream, partials, g, x,
                       static_cast<T*>(out_mean), static_cast<T*>(out_std));
    return hipGetLastError();
}

#define LSQ_INSTANTIATE(IO)                                                                                   \
    template hipError_t meanstd_per_tensor<IO>(const void*, int64_t, void*, void*, void*, hipStream_t);       \
    template hipError_t meanstd_per_channel<IO>(const void*, int64_t, int64_t, int64_t, void*, void*, void*, \
                                                size_t, hipStream_t);                                        \
    template hipError_t minmax_per_tensor<IO>(const void*, int64_t, void*, void*, void*, hipStream_t);        \
    template hipError_t minmax_per_channel<IO>(const void*, int64_t, int64_t, int64_t, void*, void*, void*,  \
                                               size_t, hipStream_t);
LSQ_INSTANTIATE(io_f32)
LSQ_INSTANTIATE(io_f64)
LSQ_INSTANTIATE(io_bf16)
LSQ_INSTANTIATE(io_f16)
#undef LSQ_INSTANTIATE

// ------------------------------------------------------------------------------------------------
// observer update + qparams + LSQ parameter store in one launch (lsq_hip_observer_update)
// ------------------------------------------------------------------------------------------------
// torch.min / torch.max of two tensors propagate NaN (unlike fmin / fmax)
__device__ __forceinline__ float torch_min(float a, float b) { return (a != a || b != b) ? __builtin_nanf("") : (a < b ? a : b); }
__device__ __forceinline__ float torch_max(float a, float b) { return (a != a || b != b) ? __builtin_nanf("") : (a > b ? a : b); }

__global__ __launch_bounds__(kBlock) void observer_update_kernel(int64_t channels, const float* __restrict__ cur_min,
                                                                 const float* __restrict__ cur_max, float* __restrict__ min_state,
                                                                 float* __restrict__ max_state, lsq_observer_update u,
                                                                 float inv_range, float* __restrict__ scale_out,
                                                                 float* __restrict__ shift_out) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (c >= channels) return;
    float mn = min_state[c], mx = max_state[c];
    const float cmn = cur_min[c], cmx = cur_max[c];
    const bool first = u.first == 1 || (u.first < 0 && mn == __builtin_inff() && mx == -__builtin_inff());
    if (first) {
        mn = cmn;
        mx = cmx;
    } else if (u.mode == 1) {
        mn = torch_min(cmn, mn);
        mx = torch_max(cmx, mx);
    } else {      // min_val + averaging_constant * (min_val_cur - min_val): three tensor operations, three roundings
        mn = mn + u.averaging_constant * (cmn - mn);
        mx = mx + u.averaging_constant * (cmx - mx);
    }
    min_state[c] = mn;
    max_state[c] = mx;
    const float min_neg = torch_min(mn, 0.0f), max_pos = torch_max(mx, 0.0f);
    float scale;
    int zp;
    if (u.symmetric) {
        scale = torch_max(torch_max(-min_neg, max_pos) * inv_range, u.eps);      // tensor / python scalar = tensor * (1 / scalar)
        zp = u.zero_point_symmetric;
    } else {
        scale = torch_max((max_pos - min_neg) * inv_range, u.eps);
        const int q = u.quant_min - static_cast<int>(__builtin_rintf(min_neg / scale));
        zp = q < u.quant_min ? u.quant_min : (q > u.quant_max ? u.quant_max : q);
    }
    scale_out[c] = scale;
    shift_out[c] = static_cast<float>(-zp) * scale;
}

hipError_t observer_update(int64_t channels, const float* cur_min, const float* cur_max, float* min_state, float* max_state,
                           const lsq_observer_update& u, float* scale_out, float* shift_out, hipStream_t stream) {
    // ATen divides a tensor by a host scalar as a multiplication by its fp32 reciprocal (BinaryDivTrueKernel): same here
    const float range = u.symmetric ? static_cast<float>(static_cast<double>(u.quant_max - u.quant_min) / 2.0)
                                    : static_cast<float>(u.quant_max - u.quant_min);
    const float inv_range = 1.0f / range;
    const unsigned grid = static_cast<unsigned>((channels + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(observer_update_kernel, dim3(grid), dim3(kBlock), 0, stream, channels, cur_min, cur_max, min_state,
                       max_state, u, inv_range, scale_out, shift_out);
    return hipGetLastError();
}

}  // namespace lsq
