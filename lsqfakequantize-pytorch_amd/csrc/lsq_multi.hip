// lsq_multi.hip -- MANY per-channel quantizers in one launch (lsq_hip_*_per_channel_multi, include/lsq_hip.h).
//
// A QAT model runs one weight quantizer per conv / linear layer -- reference quantized/modules/observers.py:458-461 called once
// per layer, i.e. dozens to hundreds of lsq_forward_per_channel / lsq_backward_per_channel calls per step on tensors of a
// few MB each.  One such call is launch-latency-bound on MI355X (BASELINE config 3, [512,512,3,3]: 7 us forward + 10 us
// backward of GPU time for 9.4 MB, and about as much host time again), so the path is accelerated horizontally: the work
// items of up to kMultiItems tensors go into ONE grid.
//
// CDNA4 design
//  * work unit = one channel of one tensor = one 256-lane workgroup, exactly the segment-mode walk of the single-tensor
//    kernels (lsq_seg_body.hpp: the same packets, groups and summation order, so y / dx / d_scale / d_shift carry the same
//    bits as the single calls); workgroup b of the grid serves channel b - first[i] of item i;
//  * the item table travels in the KERNEL ARGUMENTS (3.2 KB of the 4 KB kernarg segment for 32 items): no device table to
//    build, upload or keep alive, nothing to synchronise, HIP-graph capturable; the workgroup finds its item with a
//    32-step scalar scan of `first[]` (SGPR compares, no memory beyond the kernarg lines every workgroup reads anyway);
//  * d_scale / d_shift are finished by the workgroup itself (SegDirect): no partials, no workspace, no finalize launch.
// A tensor takes part if the single-tensor policy would give it one workgroup per channel (multi_eligible: pick_segment_mode's
// rule with its `fused` allowance for short rows on many channels); the host
// layer sends the others through the single-tensor entry points.
#include "lsq_kernels.hpp"
#include "lsq_pc_geom.hpp"
#include "lsq_seg_body.hpp"

namespace lsq {

constexpr int kMultiItems = 32;      // tensors per launch: 32 x 96 bytes of item + 33 x 4 of prefix fits the kernarg segment

template <typename T>
struct MultiItem {      // one tensor of a launch (kernel-argument image)
    const void* x;
    const void* grad;
    void* y;
    void* dx;
    const T* scale;
    const T* shift;
    T* ds;
    T* db;
    int64_t outer, C, inner;
    T gs;               // the tensor's gradient scaler (lsq_cpu.cpp:250: its own numel and channel count)
    T sym_term;
};

template <typename T>
struct MultiArgs {
    int32_t count;
    int32_t first[kMultiItems + 1];      // first[i] = workgroups before item i; first[count] = grid size
    MultiItem<T> item[kMultiItems];
};

// item of workgroup b: the number of k >= 1 with first[k] <= b (entries past the last item hold the grid size, which no
// workgroup index reaches).  Branch-free over the whole table: two wide scalar loads and 31 scalar compares, all wave-uniform.
template <typename T>
__device__ __forceinline__ int multi_item_of(const MultiArgs<T>& a, int32_t b) {
    int i = 0;
#pragma unroll
    for (int k = 1; k < kMultiItems; ++k) i += (a.first[k] <= b) ? 1 : 0;
    return i;
}

template <typename T>
__device__ __forceinline__ SegGeom multi_geom(const MultiItem<T>& it, int V) {
    SegGeom g;
    g.outer = it.outer; g.C = it.C; g.inner = it.inner;
    const int64_t W = static_cast<int64_t>(kBlock) * V;
    g.n_sub = (it.inner + W - 1) / W;
    g.sub_per_seg = g.n_sub;
    g.o_per_split = it.outer;
    g.segs = 1;
    g.osplits = 1;
    return g;
}

template <typename IO, bool INIT, int UNROLL>
__global__ __launch_bounds__(kBlock) void fwd_multi_kernel(const MultiArgs<typename IO::arith> a, Range<typename IO::arith> r) {
    const int32_t b = static_cast<int32_t>(blockIdx.x);
    const int i = __builtin_amdgcn_readfirstlane(multi_item_of(a, b));
    const MultiItem<typename IO::arith>& it = a.item[i];
    const SegGeom g = multi_geom(it, IO::VEC);
    seg_forward<IO, IO::VEC, INIT, false, UNROLL, true, true>(it.x, it.y, nullptr, 0, 0, g, SegWalk(g, b - a.first[i]), it.scale,
                                                             it.shift, r);
}

template <typename IO, bool SYM, bool INIT, bool EVAL, int UNROLL>
__global__ __launch_bounds__(kBlock) void bwd_multi_kernel(const MultiArgs<typename IO::arith> a, Range<typename IO::arith> r) {
    using T = typename IO::arith;
    const int32_t b = static_cast<int32_t>(blockIdx.x);
    const int i = __builtin_amdgcn_readfirstlane(multi_item_of(a, b));
    const MultiItem<T>& it = a.item[i];
    const SegGeom g = multi_geom(it, IO::VEC);
    const SegDirect<T> direct{it.ds, it.db, nullptr, it.sym_term};
    seg_backward<IO, IO::VEC, SYM, INIT, EVAL, UNROLL, true, true>(it.grad, it.x, it.dx, g, SegWalk(g, b - a.first[i]), it.scale,
                                                                   it.shift, r, it.gs, nullptr, 0, direct);
}

// packets in flight per lane: as the single-tensor segment kernels (lsq_per_channel.hip kSegUnroll)
template <typename IO>
constexpr int kMultiUnroll = sizeof(typename IO::elem) < 4 ? 1 : 4;

// Does the single-tensor launch policy give this tensor ONE workgroup per channel (segment mode, segs == osplits == 1)?
// Then the multi-tensor kernels walk it exactly like the single-tensor ones.  `aligned16`: every buffer 16-byte aligned.
template <typename IO>
bool multi_eligible(int64_t outer, int64_t channels, int64_t inner, bool aligned16) {
    if (outer <= 0 || channels <= 0 || inner <= 0 || channels > 0x3fffffffLL) return false;
    const int vec = pick_vec(IO::VEC, channels * inner, aligned16);
    if (vec != IO::VEC || !pick_segment_mode(vec, outer, channels, inner, device_info().cu_count, /*fused=*/true)) return false;
    const SegGeom sg = make_seg_geom(outer, channels, inner, vec, device_info().cu_count * 16);     // the segment kernels' default grid
    return sg.segs == 1 && sg.osplits == 1;
}

template <typename IO>
static hipError_t launch_multi(bool backward, const lsq_pc_item* items, int32_t count, const lsq_params& p, hipStream_t stream) {
    using T = typename IO::arith;
    const Range<T> r = make_range<T>(p);
    for (int32_t base = 0; base < count; base += kMultiItems) {
        MultiArgs<T> a;
        a.count = std::min<int32_t>(kMultiItems, count - base);
        int64_t blocks = 0;
        for (int k = 0; k < a.count; ++k) {
            const lsq_pc_item& s = items[base + k];
            a.first[k] = static_cast<int32_t>(blocks);
            MultiItem<T>& d = a.item[k];
            d.x = s.x; d.grad = s.grad; d.y = s.y; d.dx = s.dx;
            d.scale = static_cast<const T*>(s.scale); d.shift = static_cast<const T*>(s.shift);
            d.ds = static_cast<T*>(s.ds); d.db = static_cast<T*>(s.db);
            d.outer = s.outer; d.C = s.channels; d.inner = s.inner;
            d.gs = grad_scaler_per_channel<T>(s.outer * s.channels * s.inner, p.quant_max, s.channels, p.use_grad_scaling != 0, p.grad_scaler);
            d.sym_term = static_cast<T>(0) * d.gs;
            blocks += s.channels;
        }
        for (int k = a.count; k <= kMultiItems; ++k) a.first[k] = static_cast<int32_t>(blocks);
        for (int k = a.count; k < kMultiItems; ++k) a.item[k] = a.item[0];
        if (blocks > 0x7fffffffLL) return hipErrorInvalidConfiguration;
        const dim3 grid(static_cast<unsigned>(blocks));
        constexpr int U = kMultiUnroll<IO>;
        if (!backward) {
            if (p.init_mode) hipLaunchKernelGGL((fwd_multi_kernel<IO, true, U>), grid, dim3(kBlock), 0, stream, a, r);
            else hipLaunchKernelGGL((fwd_multi_kernel<IO, false, U>), grid, dim3(kBlock), 0, stream, a, r);
        } else {
            const bool sym = p.sym != 0, init = p.init_mode != 0;
#define LSQ_MULTI(S, I, E) hipLaunchKernelGGL((bwd_multi_kernel<IO, S, I, E, U>), grid, dim3(kBlock), 0, stream, a, r)
            if (p.eval_mode) { if (init) LSQ_MULTI(false, true, true); else LSQ_MULTI(false, false, true); }
            else if (sym) { if (init) LSQ_MULTI(true, true, false); else LSQ_MULTI(true, false, false); }
            else { if (init) LSQ_MULTI(false, true, false); else LSQ_MULTI(false, false, false); }
#undef LSQ_MULTI
        }
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

template <typename IO>
hipError_t forward_per_channel_multi(const lsq_pc_item* items, int32_t count, const lsq_params& p, hipStream_t stream) {
    return launch_multi<IO>(false, items, count, p, stream);
}
template <typename IO>
hipError_t backward_per_channel_multi(const lsq_pc_item* items, int32_t count, const lsq_params& p, hipStream_t stream) {
    return launch_multi<IO>(true, items, count, p, stream);
}

#define LSQ_INSTANTIATE(IO)                                                                                        \
    template bool multi_eligible<IO>(int64_t, int64_t, int64_t, bool);                                             \
    template hipError_t forward_per_channel_multi<IO>(const lsq_pc_item*, int32_t, const lsq_params&, hipStream_t); \
    template hipError_t backward_per_channel_multi<IO>(const lsq_pc_item*, int32_t, const lsq_params&, hipStream_t);
LSQ_INSTANTIATE(io_f32)
LSQ_INSTANTIATE(io_f64)
LSQ_INSTANTIATE(io_bf16)
LSQ_INSTANTIATE(io_f16)
#undef LSQ_INSTANTIATE

}  // namespace lsq
