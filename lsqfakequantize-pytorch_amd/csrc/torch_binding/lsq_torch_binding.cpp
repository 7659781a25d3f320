// lsq_torch_binding.cpp -- the host-only torch binding of liblsq_hip.so: INTEGRATION.md section 1 as code.
//
// What a maintainer of the reference would put in place of csrc/ops/cuda/lsq_cuda.cu plus
// csrc/ops/autograd/lsq_autograd.cpp: no device code, only tensor bookkeeping (checks, dense layout,
// output / workspace allocation from the caching allocator, current stream) around the C ABI of
// include/lsq_hip.h.  Compiled with g++ (no hipcc) into torchlsq/_lsq_torch.so and registered under the
// namespace `torchlsq_native`, so it lives next to the torch.library registration of torchlsq/extension.py
// (namespace `torchlsq`, ctypes) instead of fighting it for the same operator names:
//
//   torchlsq_native::lsq_forward_per_tensor / lsq_backward_per_tensor      <- lsq_cuda.cu:18-61 / 64-143
//   torchlsq_native::lsq_forward_per_channel / lsq_backward_per_channel    <- lsq_cuda.cu:147-199 / 202-297
//   torchlsq_native::lsq_backward_from_mask                                 (eval-mode backward, 1-byte mask)
//   torchlsq_native::lsq                                                    <- lsq.cpp:104-134 + lsq_autograd.cpp
//
// torchlsq.functional.lsq uses torchlsq_native::lsq for GPU tensors when this library is present: one
// dispatcher call and a C++ autograd node instead of a Python autograd.Function (host cost per
// forward+backward of a small layer: see DESIGN_HISTORY.md section 7).
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPCachingAllocator.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/csrc/autograd/custom_function.h>
#include <torch/library.h>

#include <hip/hip_runtime_api.h>

#include <array>
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <limits>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "lsq_hip.h"

namespace {

using at::Tensor;

struct Scalars {
    int64_t qmin, qmax, tmin, tmax;
    bool use_gs;
    double gs;
    bool sym, eval_mode, init_mode;
};

int dtype_code(at::ScalarType t, const char* what) {
    switch (t) {
        case at::kFloat: return LSQ_F32;
        case at::kDouble: return LSQ_F64;
        case at::kBFloat16: return LSQ_BF16;
        case at::kHalf: return LSQ_F16;
        default: TORCH_CHECK(false, "\"", what, "\" not implemented for '", c10::toString(t), "'");
    }
}

// scale/shift type for input type t: the same (lsq_cpu.cpp:28-29); 16-bit storage takes fp32 parameters.
at::ScalarType param_type(at::ScalarType t) { return (t == at::kBFloat16 || t == at::kHalf) ? at::kFloat : t; }

int32_t narrow(int64_t v, const char* name) {
    TORCH_CHECK(v >= std::numeric_limits<int32_t>::min() && v <= std::numeric_limits<int32_t>::max(), name, "=", v,
                " does not fit a 32-bit integer");
    return static_cast<int32_t>(v);
}

lsq_params pack(const Scalars& s, int64_t numel_for_scaler = 0) {
    lsq_params p;
    p.quant_min = narrow(s.qmin, "quant_min");
    p.quant_max = narrow(s.qmax, "quant_max");
    p.type_min = narrow(s.tmin, "type_min");
    p.type_max = narrow(s.tmax, "type_max");
    p.use_grad_scaling = s.use_gs;
    p.sym = s.sym;
    p.eval_mode = s.eval_mode;
    p.init_mode = s.init_mode;
    p.grad_scaler = s.gs;
    p.numel_for_scaler = numel_for_scaler;   // <= 0: this call's own element count (the reference behaviour)
    return p;
}

void status(int rc, const char* what) { TORCH_CHECK(rc == 0, what, " failed (", rc, "): ", lsq_hip_last_error()); }

// every tensor of a call lives on the GPU the kernel is launched on (the first tensor's): raw pointers of another
// device would only work by accident of peer access
void require_gpu(const char* what, std::initializer_list<const Tensor*> ts) {
    const Tensor* first = *ts.begin();
    for (const Tensor* t : ts) {
        TORCH_CHECK(t->is_cuda(), what, ": expected a tensor on the GPU (HIP device) but got device ", t->device());
        TORCH_CHECK(t->device() == first->device(), what, ": expected all tensors on ", first->device(), " but got one on ",
                    t->device());
    }
}

void require_param(const char* what, const Tensor& scale, const Tensor& shift) {
    TORCH_CHECK(scale.numel() >= 1 && shift.numel() >= 1, what, ": scale and shift need at least one element");
}

void check_forward_types(const Tensor& x, const Tensor& scale, const Tensor& shift) {
    dtype_code(x.scalar_type(), "lsq_forward");
    const auto pt = param_type(x.scalar_type());
    TORCH_CHECK(scale.scalar_type() == pt, "`input` and `scale` must have the same floating-point type");
    TORCH_CHECK(shift.scalar_type() == pt, "`input` and `shift` must have the same floating-point type");
}

void check_backward_types(const Tensor& grad, const Tensor& x, const Tensor& scale, const Tensor& shift) {
    dtype_code(x.scalar_type(), "lsq_backward");
    const auto pt = param_type(x.scalar_type());
    TORCH_CHECK(grad.scalar_type() == x.scalar_type(), "`grad` and `input` must have the same floating-point type");
    TORCH_CHECK(scale.scalar_type() == pt, "`grad` and `scale` must have the same floating-point type");
    TORCH_CHECK(shift.scalar_type() == pt, "`grad` and `shift` must have the same floating-point type");
    TORCH_CHECK(x.numel() == grad.numel(), "`x` and `grad` are not the same size");
}

void check_channel_args(const Tensor& x, const Tensor& scale, const Tensor& shift, int64_t axis) {
    TORCH_CHECK(scale.numel() == shift.numel(), "scale and shift need to have the same dimensions");
    TORCH_CHECK(axis >= 0 && axis < x.dim(), "`axis` must be between 0 and number of dimensions of input");
    TORCH_CHECK(scale.numel() == x.size(axis), "dimensions of scale and shift are not consistent with input tensor");
}

// The kernels see dense memory.  Channel geometry of dense `t` along `axis` in MEMORY order -- the
// [outer, C, inner] view of lsq_hip.h -- for any permutation of a contiguous tensor (channels-last
// included); a tensor that is not dense is replaced by a contiguous copy first.
struct Geometry {
    int64_t outer, channels, inner;
};

Tensor dense(const Tensor& t) { return t.is_non_overlapping_and_dense() ? t : t.contiguous(); }

Geometry geometry(const Tensor& t, int64_t axis) {
    const int64_t c = t.size(axis);
    if (c == 1) return {1, 1, t.numel()};  // one channel covering the whole tensor
    // dims slower than `axis` in memory have a larger stride (ties cannot happen between non-unit dims of a dense tensor)
    int64_t outer = 1, inner = 1;
    const int64_t sa = t.stride(axis);
    for (int64_t d = 0; d < t.dim(); ++d) {
        if (d == axis || t.size(d) == 1) continue;
        if (t.stride(d) > sa) outer *= t.size(d);
        else inner *= t.size(d);
    }
    return {outer, c, inner};
}

void* stream_of(const Tensor& t) { return c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

// memory order of a dense tensor: its non-unit dims, slowest first (empty: not dense)
std::vector<int64_t> physical_order(const Tensor& t) {
    std::vector<int64_t> dims;
    for (int64_t d = 0; d < t.dim(); ++d)
        if (t.size(d) != 1) dims.push_back(d);
    std::stable_sort(dims.begin(), dims.end(), [&](int64_t a, int64_t b) { return t.stride(a) > t.stride(b); });
    int64_t expect = 1;
    for (auto it = dims.rbegin(); it != dims.rend(); ++it) {
        if (t.stride(*it) != expect) return {};
        expect *= t.size(*it);
    }
    return dims;
}

// grad laid out exactly like the dense xd.  Two dense orders that differ by ONE swap of adjacent dimension groups -- g's memory
// [A][B][C], xd's [A][C][B]: a contiguous NCHW gradient for a channels-last input, or the reverse -- go through the library's
// tiled pass (lsq_hip_relayout); anything else through Tensor.copy_ (torchlsq/_hip_host.py: _like_layout, the same rule).
Tensor like_layout(const Tensor& g, const Tensor& xd) {
    if (g.sizes() == xd.sizes() && g.strides() == xd.strides()) return g;
    Tensor out = at::empty_like(xd);  // preserve_format keeps the strides of the dense xd
    if (g.is_cuda() && g.sizes() == xd.sizes() && g.scalar_type() == xd.scalar_type() && g.device() == xd.device()) {
        const std::vector<int64_t> og = physical_order(g), ox = physical_order(xd);
        if (!og.empty() && og.size() == ox.size() && og != ox) {
            size_t k0 = 0;
            while (og[k0] == ox[k0]) ++k0;
            const size_t n = og.size() - k0;
            for (size_t k = 1; k < n; ++k) {
                bool rotated = true;
                for (size_t i = 0; i < n && rotated; ++i) rotated = og[k0 + (k + i) % n] == ox[k0 + i];
                if (!rotated) continue;
                int64_t A = 1, B = 1, C = 1;
                for (size_t i = 0; i < k0; ++i) A *= g.size(og[i]);
                for (size_t i = 0; i < k; ++i) B *= g.size(og[k0 + i]);
                for (size_t i = k; i < n; ++i) C *= g.size(og[k0 + i]);
                c10::DeviceGuard guard(xd.device());
                status(lsq_hip_relayout(dtype_code(g.scalar_type(), "lsq_backward"), g.data_ptr(), out.data_ptr(), A, B, C, stream_of(xd)),
                       "lsq_hip_relayout");
                return out;
            }
        }
    }
    out.copy_(g.sizes() == xd.sizes() ? g : g.reshape(xd.sizes()));
    return out;
}

Tensor byte_workspace(const Tensor& like, size_t nbytes) {
    return at::empty({static_cast<int64_t>(nbytes < 256 ? 256 : nbytes)}, like.options().dtype(at::kByte));
}

// ---- tickets (lsq_bwd_extras): persistent per-stream arrival counters that make the backward ONE launch ----
// The C ABI wants LSQ_TICKET_BYTES of zero-initialised device memory that outlives the call and is never shared by
// launches that can run concurrently.  One slab of kTicketSlots tickets per device is allocated (and zeroed) at the
// first eager backward on that device -- with hipMalloc and never freed: a static holding an at::Tensor would be destroyed
// after the HIP runtime at process exit -- and streams get a slot each on first use.  A launch that is being captured
// into a HIP graph gets no ticket (two-launch route): the graph may be replayed on any stream, next to eager work on the
// capture stream, and two concurrent launches must never share an arrival counter.
// Mode 2 ("auto", the default): per-tensor tensors of at most 8 MB -- host-bound in eager mode, where one launch less is
// 11-15 % of the forward + backward wall time; on the GPU the single-launch route is not faster, and 2-3 us slower where the
// kernel is busy (profiles/r03_ticket_sizes.txt).  TORCHLSQ_SINGLE_LAUNCH_BACKWARD=1: always, =0: never.
std::atomic<int> g_ticket_mode{[] {
    const char* e = std::getenv("TORCHLSQ_SINGLE_LAUNCH_BACKWARD");
    return !e ? 2 : (e[0] == '1' ? 1 : (e[0] == '0' ? 0 : 2));
}()};
constexpr int kTicketSlots = 64;
// Behind every ticket sits the per-tensor backward's workspace (its block partial sums: a constant 256 KiB): a launch that has
// a ticket -- eager, not being captured, alone on its stream slot -- has a private scratch area with the same lifetime rules, so
// the host-bound small tensors the ticket exists for also skip one allocator round trip per backward (~0.9 us of ~9).
inline size_t ticket_workspace_bytes() {
    static const size_t b = (lsq_hip_backward_per_tensor_workspace(LSQ_F32, 1) + 4095) & ~size_t(4095);
    return b;
}
inline size_t ticket_slot_stride() { return LSQ_TICKET_BYTES + ticket_workspace_bytes(); }
struct TicketSlab {
    char* base = nullptr;
    int next = 0;
};
std::mutex g_ticket_mutex;
std::map<int, TicketSlab> g_ticket_slabs;                    // device index -> slab
std::map<std::pair<int, void*>, void*> g_tickets;            // (device index, stream) -> ticket

void* ticket_for(const Tensor& x, void* stream, bool per_channel) {
    // the decision is the library's (include/lsq_hip.h: one rule for both host layers); the mode is this layer's state
    if (!lsq_hip_policy_ticket(g_ticket_mode.load(std::memory_order_relaxed), per_channel ? 1 : 0,
                               x.numel() * static_cast<int64_t>(x.element_size())))
        return nullptr;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(static_cast<hipStream_t>(stream), &st) != hipSuccess || st != hipStreamCaptureStatusNone)
        return nullptr;                                       // captured launches: two-launch route
    const int dev = x.device().index();
    std::lock_guard<std::mutex> lock(g_ticket_mutex);
    const auto key = std::make_pair(dev, stream);
    const auto hit = g_tickets.find(key);
    if (hit != g_tickets.end()) return hit->second;
    TicketSlab& slab = g_ticket_slabs[dev];
    if (!slab.base) {
        // From PyTorch's caching allocator (raw_alloc: normally a cached block, no hipMalloc) and zeroed on the CALLING stream,
        // which was just seen not to be capturing, with a stream-level wait: a hipMalloc / device-wide synchronisation here
        // would be illegal -- and would invalidate the capture -- while ANOTHER stream captures in global mode.  Never
        // returned to the allocator (a raw pointer, not a Tensor: nothing to destroy after the runtime at process exit).
        const size_t bytes = static_cast<size_t>(kTicketSlots) * ticket_slot_stride();
        void* p = nullptr;
        try {
            c10::DeviceGuard guard(x.device());
            p = c10::hip::HIPCachingAllocator::raw_alloc(bytes);
        } catch (...) {
            return nullptr;                                   // two-launch route; asked again at the next backward
        }
        if (hipMemsetAsync(p, 0, bytes, static_cast<hipStream_t>(stream)) != hipSuccess ||
            hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess) {   // zeroed before any other stream uses a slot (one-off)
            (void)hipGetLastError();
            return nullptr;
        }
        slab.base = static_cast<char*>(p);
    }
    if (slab.next >= kTicketSlots) return nullptr;            // more streams than slots: two-launch route for the rest
    void* t = slab.base + static_cast<size_t>(slab.next++) * ticket_slot_stride();
    g_tickets.emplace(key, t);
    return t;
}

// ---- the four kernels of the reference + the masked eval backward --------------------------------------

std::tuple<Tensor, Tensor> forward_impl(const Tensor& x, const Tensor& scale, const Tensor& shift, bool per_channel,
                                        int64_t axis, const Scalars& s, bool want_mask) {
    check_forward_types(x, scale, shift);
    if (per_channel) check_channel_args(x, scale, shift, axis);
    const char* what = per_channel ? "lsq_forward_per_channel" : "lsq_forward_per_tensor";
    require_gpu(what, {&x, &scale, &shift});
    const Tensor xd = dense(x);
    Tensor y = at::empty_like(xd);
    Tensor mask;
    if (want_mask) mask = at::empty_strided(xd.sizes(), xd.strides(), xd.options().dtype(at::kChar));
    if (xd.numel() == 0) return {y, mask};
    require_param(what, scale, shift);
    const lsq_params p = pack(s);
    const lsq_fwd_extras ex{want_mask ? mask.data_ptr() : nullptr, 0, 1};
    const Tensor sc = scale.contiguous(), sh = shift.contiguous();
    const int code = dtype_code(x.scalar_type(), "lsq_forward");
    c10::DeviceGuard guard(x.device());
    if (per_channel) {
        const Geometry g = geometry(xd, axis);
        status(lsq_hip_forward_per_channel(code, xd.data_ptr(), y.data_ptr(), g.outer, g.channels, g.inner, sc.data_ptr(),
                                           sh.data_ptr(), &p, want_mask ? &ex : nullptr, stream_of(x)),
               "lsq_hip_forward_per_channel");
    } else {
        status(lsq_hip_forward_per_tensor(code, xd.data_ptr(), y.data_ptr(), xd.numel(), sc.data_ptr(), sh.data_ptr(), &p,
                                          want_mask ? &ex : nullptr, stream_of(x)),
               "lsq_hip_forward_per_tensor");
    }
    return {y, mask};
}

// `want_wide`: also return the un-rounded fp64 sums ([2] per-tensor, [2, C] per-channel; d_scale sums first) -- what
// the batch-sharded path all-reduces -- and use `numel_for_scaler` (the GLOBAL element count) in the gradient scaler.
struct BackwardOut {
    Tensor dx, ds, db, wide;
};

BackwardOut backward_impl(const Tensor& grad, const Tensor& x, const Tensor& scale, const Tensor& shift, bool per_channel,
                          int64_t axis, const Scalars& s, int64_t numel_for_scaler = 0, bool want_wide = false,
                          const Tensor* wide_out = nullptr /* caller's fp64 buffer of at least 2 C (or 2) elements */) {
    check_backward_types(grad, x, scale, shift);
    if (per_channel) check_channel_args(x, scale, shift, axis);
    if (x.numel() <= 0) {   // lsq_cpu.cpp:76-78,221-223 return (x, scale, shift) themselves
        Tensor wide;
        if (want_wide) {
            const auto dopt = x.options().dtype(at::kDouble);
            wide = per_channel ? at::zeros({2, scale.numel()}, dopt) : at::zeros({2}, dopt);
        }
        return {x.clone(), scale.clone(), shift.clone(), wide};
    }
    const char* what = per_channel ? "lsq_backward_per_channel" : "lsq_backward_per_tensor";
    require_gpu(what, {&x, &grad, &scale, &shift});
    require_param(what, scale, shift);
    const Tensor xd = dense(x);
    const Tensor gd = like_layout(grad, xd);
    Tensor dx = at::empty_like(xd);
    const lsq_params p = pack(s, numel_for_scaler);
    const Tensor sc = scale.contiguous(), sh = shift.contiguous();
    const int code = dtype_code(x.scalar_type(), "lsq_backward");
    const auto popt = x.options().dtype(param_type(x.scalar_type()));
    c10::DeviceGuard guard(x.device());
    void* const stream = stream_of(x);
    const lsq_bwd_extras extras{ticket_for(x, stream, per_channel)};
    Tensor wide;
    if (per_channel) {
        const Geometry g = geometry(xd, axis);
        Tensor ds = at::empty({g.channels}, popt), db = at::empty({g.channels}, popt);
        if (want_wide) wide = wide_out ? *wide_out : at::empty({2, g.channels}, x.options().dtype(at::kDouble));
        const Tensor ws = byte_workspace(x, lsq_hip_backward_per_channel_workspace(code, g.outer, g.channels, g.inner));
        status(lsq_hip_backward_per_channel(code, gd.data_ptr(), xd.data_ptr(), dx.data_ptr(), ds.data_ptr(), db.data_ptr(),
                                            want_wide ? wide.data_ptr<double>() : nullptr, g.outer, g.channels, g.inner,
                                            sc.data_ptr(), sh.data_ptr(), &p, &extras, ws.data_ptr(),
                                            static_cast<size_t>(ws.numel()), stream),
               "lsq_hip_backward_per_channel");
        return {dx, ds, db, wide};
    }
    Tensor ds = at::empty({1}, popt), db = at::empty({1}, popt);
    if (want_wide) wide = wide_out ? *wide_out : at::empty({2}, x.options().dtype(at::kDouble));
    Tensor ws;                                   // with a ticket: the slot's own workspace, no allocation
    void* ws_ptr = nullptr;
    size_t ws_bytes = 0;
    if (extras.ticket) {
        ws_ptr = static_cast<char*>(extras.ticket) + LSQ_TICKET_BYTES;
        ws_bytes = ticket_workspace_bytes();
    } else {
        ws = byte_workspace(x, lsq_hip_backward_per_tensor_workspace(code, xd.numel()));
        ws_ptr = ws.data_ptr();
        ws_bytes = static_cast<size_t>(ws.numel());
    }
    status(lsq_hip_backward_per_tensor(code, gd.data_ptr(), xd.data_ptr(), dx.data_ptr(), ds.data_ptr(), db.data_ptr(),
                                       want_wide ? wide.data_ptr<double>() : nullptr, xd.numel(), sc.data_ptr(), sh.data_ptr(), &p,
                                       &extras, ws_ptr, ws_bytes, stream),
           "lsq_hip_backward_per_tensor");
    return {dx, ds, db, wide};
}

// ---- the batch-sharded backward in ONE host call (torchlsq.distributed.sharded_backward(async_op=True), native route) ------------
// backward (`*_wide`: the global element count in the scaler, the un-rounded fp64 sums out) + lsq_hip_comm_all_reduce_begin of
// those sums on the communicator's own stream + their rounding to the parameter type BEHIND the reduction on that stream: the
// compute stream never waits for a single reduction (the caller joins once per step, lsq_hip_comm_all_reduce_end on the
// returned ticket).  `comm` is the lsq_comm* the Python layer created (torchlsq._hip_host.HipComm.handle).
// Returns (dx, wide, rounded, ticket): wide [2] / [2, C] fp64 (all-reduced in place once the side stream gets there),
// rounded = wide in the parameter type (valid on the caller's stream after the join).
std::tuple<Tensor, Tensor, Tensor, int64_t> backward_sharded(const Tensor& grad, const Tensor& x, const Tensor& scale, const Tensor& shift,
                                                             bool per_channel, int64_t axis, const Scalars& s, int64_t numel_for_scaler,
                                                             int64_t comm) {
    TORCH_CHECK(comm != 0, "lsq_backward_*_sharded: no communicator");
    BackwardOut o = backward_impl(grad, x, scale, shift, per_channel, axis, s, numel_for_scaler, /*want_wide=*/true);
    lsq_comm* const c = reinterpret_cast<lsq_comm*>(static_cast<intptr_t>(comm));
    const c10::Device dev = o.wide.device();
    c10::DeviceGuard guard(dev);
    int32_t ticket = -1;
    status(lsq_hip_comm_all_reduce_begin(c, o.wide.data_ptr(), o.wide.data_ptr(), o.wide.numel(), LSQ_F64, LSQ_COMM_SUM, stream_of(o.wide), &ticket),
           "lsq_hip_comm_all_reduce_begin");
    Tensor rounded;
    {
        const c10::hip::HIPStream side = c10::hip::getStreamFromExternal(static_cast<hipStream_t>(lsq_hip_comm_side_stream(c)), dev.index());
        c10::hip::HIPStreamGuard on_side(side);
        rounded = o.wide.to(param_type(x.scalar_type()));
        c10::hip::HIPCachingAllocator::recordStream(o.wide.storage().data_ptr(), side);   // `wide` is read over there
    }
    return {o.dx, o.wide, rounded, static_cast<int64_t>(ticket)};
}

std::tuple<Tensor, Tensor, Tensor, int64_t> backward_per_tensor_sharded(const Tensor& grad, const Tensor& x, const Tensor& scale,
                                                                        const Tensor& shift, int64_t qmin, int64_t qmax, int64_t tmin,
                                                                        int64_t tmax, bool use_gs, double gs, bool sym, bool eval_mode,
                                                                        bool init_mode, int64_t numel_for_scaler, int64_t comm) {
    return backward_sharded(grad, x, scale, shift, false, 0, {qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode}, numel_for_scaler, comm);
}

std::tuple<Tensor, Tensor, Tensor, int64_t> backward_per_channel_sharded(const Tensor& grad, const Tensor& x, const Tensor& scale,
                                                                         const Tensor& shift, int64_t axis, int64_t qmin, int64_t qmax,
                                                                         int64_t tmin, int64_t tmax, bool use_gs, double gs, bool sym,
                                                                         bool eval_mode, bool init_mode, int64_t numel_for_scaler, int64_t comm) {
    return backward_sharded(grad, x, scale, shift, true, axis, {qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode}, numel_for_scaler, comm);
}

// ---- the sharded backward when no rank knows the global element count (torchlsq.distributed, global_numel=COLLECTIVE) ----------
// lsq_backward_packed: the local backward with UNSCALED terms into packed = [sum ds terms (C), sum db terms (C), this shard's
// element count] -- the buffer of the ONE all-reduce; lsq_sharded_finish: d_scale / d_shift from the all-reduced buffer, the
// gradient scaler derived from the summed count on the device (lsq_hip_sharded_finish).  The collective in between is the
// caller's (the library's communicator or torch.distributed).
std::tuple<Tensor, Tensor> backward_packed(const Tensor& grad, const Tensor& x, const Tensor& scale, const Tensor& shift, bool per_channel,
                                           int64_t axis, int64_t qmin, int64_t qmax, int64_t tmin, int64_t tmax, bool sym, bool init_mode) {
    const int64_t C = per_channel ? scale.numel() : 1;
    Tensor packed = at::full({2 * C + 1}, static_cast<double>(x.numel()), x.options().dtype(at::kDouble));
    if (x.numel() <= 0) {
        packed.narrow(0, 0, 2 * C).zero_();
        return {x.clone(), packed};
    }
    BackwardOut o = backward_impl(grad, x, scale, shift, per_channel, axis, {qmin, qmax, tmin, tmax, false, 1.0, sym, false, init_mode}, 0,
                                  /*want_wide=*/true, &packed);
    return {o.dx, packed};
}

std::tuple<Tensor, Tensor> sharded_finish(const Tensor& packed, int64_t channels, bool per_channel, int64_t x_dtype_code, int64_t qmax,
                                          bool use_gs, double gs) {
    require_gpu("lsq_sharded_finish", {&packed});
    TORCH_CHECK(packed.scalar_type() == at::kDouble && packed.is_contiguous() && packed.numel() == 2 * channels + 1,
                "lsq_sharded_finish: packed must be a contiguous float64 tensor of 2 * channels + 1 elements");
    const auto ptype = x_dtype_code == LSQ_F64 ? at::kDouble : at::kFloat;
    Tensor ds = at::empty({channels}, packed.options().dtype(ptype)), db = at::empty({channels}, packed.options().dtype(ptype));
    const lsq_params p = pack({0, qmax, 0, qmax, use_gs, gs, false, false, false});
    c10::DeviceGuard guard(packed.device());
    status(lsq_hip_sharded_finish(static_cast<int>(x_dtype_code), packed.data_ptr<double>(), channels, per_channel ? 1 : 0, &p,
                                  ds.data_ptr(), db.data_ptr(), stream_of(packed)),
           "lsq_hip_sharded_finish");
    return {ds, db};
}

Tensor backward_from_mask(const Tensor& grad, const Tensor& mask) {
    const int code = dtype_code(grad.scalar_type(), "lsq_backward");
    TORCH_CHECK(mask.scalar_type() == at::kChar && mask.numel() == grad.numel(),
                "`mask` must be the int8 inside mask of the forward");
    require_gpu("lsq_backward_from_mask", {&grad, &mask});
    Tensor gd = grad;
    if (!(grad.sizes() == mask.sizes() && grad.strides() == mask.strides())) {
        gd = at::empty_strided(mask.sizes(), mask.strides(), grad.options());
        gd.copy_(grad.sizes() == mask.sizes() ? grad : grad.reshape(mask.sizes()));
    }
    Tensor dx = at::empty_strided(mask.sizes(), mask.strides(), grad.options());
    if (grad.numel() == 0) return dx;
    c10::DeviceGuard guard(grad.device());
    status(lsq_hip_backward_from_mask(code, gd.data_ptr(), mask.data_ptr(), dx.data_ptr(), gd.numel(), stream_of(grad)),
           "lsq_hip_backward_from_mask");
    return dx;
}

// ---- operator-level wrappers (schemas of lsq.cpp:138-145) -----------------------------------------------

Tensor forward_per_tensor(const Tensor& x, const Tensor& scale, const Tensor& shift, int64_t qmin, int64_t qmax,
                          int64_t tmin, int64_t tmax, bool use_gs, double gs, bool sym, bool eval_mode, bool init_mode) {
    return std::get<0>(forward_impl(x, scale, shift, false, 0, {qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode}, false));
}

std::tuple<Tensor, Tensor, Tensor> backward_per_tensor(const Tensor& grad, const Tensor& x, const Tensor& scale,
                                                       const Tensor& shift, int64_t qmin, int64_t qmax, int64_t tmin,
                                                       int64_t tmax, bool use_gs, double gs, bool sym, bool eval_mode,
                                                       bool init_mode) {
    BackwardOut o = backward_impl(grad, x, scale, shift, false, 0, {qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode});
    return {o.dx, o.ds, o.db};
}

Tensor forward_per_channel(const Tensor& x, const Tensor& scale, const Tensor& shift, int64_t axis, int64_t qmin,
                           int64_t qmax, int64_t tmin, int64_t tmax, bool use_gs, double gs, bool sym, bool eval_mode,
                           bool init_mode) {
    return std::get<0>(forward_impl(x, scale, shift, true, axis, {qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode}, false));
}

std::tuple<Tensor, Tensor, Tensor> backward_per_channel(const Tensor& grad, const Tensor& x, const Tensor& scale,
                                                        const Tensor& shift, int64_t axis, int64_t qmin, int64_t qmax,
                                                        int64_t tmin, int64_t tmax, bool use_gs, double gs, bool sym,
                                                        bool eval_mode, bool init_mode) {
    BackwardOut o = backward_impl(grad, x, scale, shift, true, axis, {qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode});
    return {o.dx, o.ds, o.db};
}

// The batch-sharded backward (torchlsq/distributed.py): dx and the un-rounded fp64 sums, with the GLOBAL element count
// in the gradient scaler.  No reference counterpart.
std::tuple<Tensor, Tensor> backward_per_tensor_wide(const Tensor& grad, const Tensor& x, const Tensor& scale, const Tensor& shift,
                                                    int64_t qmin, int64_t qmax, int64_t tmin, int64_t tmax, bool use_gs, double gs,
                                                    bool sym, bool eval_mode, bool init_mode, int64_t numel_for_scaler) {
    BackwardOut o = backward_impl(grad, x, scale, shift, false, 0, {qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode},
                                  numel_for_scaler, true);
    return {o.dx, o.wide};
}

std::tuple<Tensor, Tensor> backward_per_channel_wide(const Tensor& grad, const Tensor& x, const Tensor& scale, const Tensor& shift,
                                                     int64_t axis, int64_t qmin, int64_t qmax, int64_t tmin, int64_t tmax, bool use_gs,
                                                     double gs, bool sym, bool eval_mode, bool init_mode, int64_t numel_for_scaler) {
    BackwardOut o = backward_impl(grad, x, scale, shift, true, axis, {qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode},
                                  numel_for_scaler, true);
    return {o.dx, o.wide};
}

// ---- autograd node: LSQPerTensorFunction / LSQPerChannelFunction of lsq_autograd.cpp:16-74,111-173 in one ----
// saves {input, scale, shift} (or {mask, scale, shift} in eval mode: the backward then only needs "was the
// element strictly inside the range", lsq_kernel.h:126-145), returns gradients for those three only.
class LsqNode : public torch::autograd::Function<LsqNode> {
   public:
    static Tensor forward(torch::autograd::AutogradContext* ctx, const Tensor& x, const Tensor& scale,
                          const Tensor& shift, int64_t qmin, int64_t qmax, int64_t tmin, int64_t tmax, int64_t axis,
                          bool use_gs, double gs, bool sym, bool per_channel, bool eval_mode, bool init_mode, bool mask_backward) {
        at::AutoDispatchBelowADInplaceOrView below;
        // mask_backward == false: the reference's eval backward, from x and the parameters as they are at backward time
        // (lsq_autograd.cpp:46-73) -- what LSQFakeQuantizer asks for while its observer rewrites them on every call
        const bool masked = lsq_hip_policy_saves_mask(eval_mode, init_mode, x.requires_grad(), mask_backward) != 0;
        const Scalars s{qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode};
        auto [y, mask] = forward_impl(x, scale, shift, per_channel, axis, s, masked);
        ctx->save_for_backward({masked ? mask : x, scale, shift});
        // two map entries instead of twelve: the integers and flags as one list, the scaler as a double
        const int64_t flags = (use_gs ? 1 : 0) | (sym ? 2 : 0) | (per_channel ? 4 : 0) | (eval_mode ? 8 : 0) |
                              (init_mode ? 16 : 0) | (masked ? 32 : 0);
        ctx->saved_data["cfg"] = c10::IValue(std::vector<int64_t>{qmin, qmax, tmin, tmax, axis, flags});
        ctx->saved_data["gs"] = gs;
        return y;
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx,
                                                   const torch::autograd::variable_list& grads) {
        const auto saved = ctx->get_saved_variables();
        const std::vector<int64_t> cfg = ctx->saved_data["cfg"].toIntVector();
        const int64_t flags = cfg[5];
        Tensor dx, ds, db;
        if (flags & 32) {   // masked: saved[0] is the inside mask, d_scale = d_shift = 0
            dx = backward_from_mask(grads[0], saved[0]);
            ds = at::zeros_like(saved[1]);
            db = at::zeros_like(saved[2]);
        } else {
            const Scalars s{cfg[0], cfg[1], cfg[2], cfg[3], (flags & 1) != 0, ctx->saved_data["gs"].toDouble(),
                            (flags & 2) != 0, (flags & 8) != 0, (flags & 16) != 0};
            BackwardOut o = backward_impl(grads[0], saved[0], saved[1], saved[2], (flags & 4) != 0, cfg[4], s);
            dx = o.dx; ds = o.ds; db = o.db;
        }
        torch::autograd::variable_list out(15);
        out[0] = dx; out[1] = ds; out[2] = db;
        return out;
    }
};

// quantops::ops::lsq (lsq.cpp:104-134): checks, per-channel broadcast of single-element parameters, routing.
Tensor lsq_impl(const Tensor& x, const Tensor& scale, const Tensor& shift, int64_t qmin, int64_t qmax, int64_t tmin,
           int64_t tmax, int64_t axis, bool use_gs, double gs, bool is_affine, bool is_perchannel, bool eval_mode,
           bool init_mode, bool mask_backward) {
    // (registered twice below with the reference's argument list: `lsq` = mask_backward true, `lsq_keep_input` = false)
    TORCH_CHECK(scale.dim() == 1, "scale should be a 1-D tensor, even in per tensor case(please, avoid torch.Scalar too)");
    TORCH_CHECK(shift.dim() == 1, "shift should be a 1-D tensor, even in per tensor case(please, avoid torch.Scalar too)");
    Tensor sc = scale, sh = shift;
    if (is_perchannel) {
        const int64_t size = std::max(sc.size(0), sh.size(0));
        if (sc.size(0) != size) sc = sc.repeat({size});  // differentiable: a size-1 leaf receives the summed gradient
        if (sh.size(0) != size) sh = sh.repeat({size});
    }
    return LsqNode::apply(x, sc, sh, qmin, qmax, tmin, tmax, axis, use_gs, gs, !is_affine, is_perchannel, eval_mode, init_mode,
                          mask_backward);
}

// ---- many per-channel quantizers in one launch: lsq_hip_*_per_channel_multi behind one autograd node --------------------
// (torchlsq.functional.lsq_foreach picks the tensors that qualify -- lsq_hip_per_channel_multi_ok -- and hands them over here;
// the host cost per tensor is an output allocation and a table row)
struct MultiCfg {
    Scalars s;
    std::vector<int64_t> axes;
};

void multi_rows(const std::vector<Tensor>& xs, const std::vector<Tensor>& scales, const std::vector<Tensor>& shifts,
                const std::vector<int64_t>& axes, std::vector<lsq_pc_item>& rows, std::vector<Tensor>& keep, bool backward,
                const char* what) {
    const size_t n = xs.size();
    rows.resize(n);
    for (size_t i = 0; i < n; ++i) {
        const Tensor& x = xs[i];
        TORCH_CHECK(x.scalar_type() == xs[0].scalar_type(), what, ": all tensors must have the same floating-point type");
        if (backward) dtype_code(x.scalar_type(), "lsq_backward"); else check_forward_types(x, scales[i], shifts[i]);
        check_channel_args(x, scales[i], shifts[i], axes[i]);
        require_gpu(what, {&xs[0], &x, &scales[i], &shifts[i]});
        TORCH_CHECK(x.is_contiguous(), what, ": tensor ", i, " must be contiguous");
        const Geometry g = geometry(x, axes[i]);
        keep.push_back(scales[i].contiguous());
        keep.push_back(shifts[i].contiguous());
        lsq_pc_item& r = rows[i];
        r = lsq_pc_item{};
        r.x = x.data_ptr();
        r.scale = keep[keep.size() - 2].data_ptr();
        r.shift = keep[keep.size() - 1].data_ptr();
        r.outer = g.outer; r.channels = g.channels; r.inner = g.inner;
    }
}

// the two backend ops of the multi-tensor path (no autograd): what LsqForeachNode below calls from its forward / backward
std::vector<Tensor> forward_per_channel_multi(at::TensorList xs, at::TensorList scales, at::TensorList shifts, at::IntArrayRef axes,
                                              int64_t qmin, int64_t qmax, int64_t tmin, int64_t tmax, bool use_gs, double gs, bool sym,
                                              bool eval_mode, bool init_mode) {
    const size_t n = xs.size();
    TORCH_CHECK(n > 0 && scales.size() == n && shifts.size() == n && axes.size() == n, "lsq_forward_per_channel_multi: list lengths differ");
    const Scalars s{qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode};
    std::vector<lsq_pc_item> rows;
    std::vector<Tensor> keep;
    multi_rows(xs.vec(), scales.vec(), shifts.vec(), axes.vec(), rows, keep, false, "lsq_forward_per_channel_multi");
    std::vector<Tensor> ys(n);
    for (size_t i = 0; i < n; ++i) {
        ys[i] = at::empty_like(xs[i]);
        rows[i].y = ys[i].data_ptr();
    }
    const lsq_params p = pack(s);
    c10::DeviceGuard guard(xs[0].device());
    status(lsq_hip_forward_per_channel_multi(dtype_code(xs[0].scalar_type(), "lsq_forward"), rows.data(), static_cast<int32_t>(n), &p,
                                             stream_of(xs[0])),
           "lsq_hip_forward_per_channel_multi");
    return ys;
}

// returns [dx_0 .. dx_{n-1}, ds_0 .. ds_{n-1}, db_0 .. db_{n-1}]
std::vector<Tensor> backward_per_channel_multi(at::TensorList grads, at::TensorList xs, at::TensorList scales, at::TensorList shifts,
                                               at::IntArrayRef axes, int64_t qmin, int64_t qmax, int64_t tmin, int64_t tmax, bool use_gs,
                                               double gs, bool sym, bool eval_mode, bool init_mode) {
    const size_t n = xs.size();
    TORCH_CHECK(n > 0 && grads.size() == n && scales.size() == n && shifts.size() == n && axes.size() == n,
                "lsq_backward_per_channel_multi: list lengths differ");
    const Scalars s{qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode};
    std::vector<lsq_pc_item> rows;
    std::vector<Tensor> keep;
    multi_rows(xs.vec(), scales.vec(), shifts.vec(), axes.vec(), rows, keep, true, "lsq_backward_per_channel_multi");
    std::vector<Tensor> out(3 * n);
    for (size_t i = 0; i < n; ++i) {
        Tensor g = like_layout(grads[i], xs[i]);
        check_backward_types(g, xs[i], scales[i], shifts[i]);
        if (reinterpret_cast<uintptr_t>(g.data_ptr()) & 15u) g = g.clone();
        keep.push_back(g);
        const auto popt = xs[i].options().dtype(param_type(xs[i].scalar_type()));
        out[i] = at::empty_like(xs[i]);
        out[n + i] = at::empty({rows[i].channels}, popt);
        out[2 * n + i] = at::empty({rows[i].channels}, popt);
        rows[i].grad = g.data_ptr();
        rows[i].dx = out[i].data_ptr();
        rows[i].ds = out[n + i].data_ptr();
        rows[i].db = out[2 * n + i].data_ptr();
    }
    const lsq_params p = pack(s);
    c10::DeviceGuard guard(xs[0].device());
    status(lsq_hip_backward_per_channel_multi(dtype_code(xs[0].scalar_type(), "lsq_backward"), rows.data(), static_cast<int32_t>(n), &p,
                                              stream_of(xs[0])),
           "lsq_hip_backward_per_channel_multi");
    return out;
}

class LsqForeachNode : public torch::autograd::Function<LsqForeachNode> {
   public:
    // (the tensors arrive as at::TensorList: that is the list type custom Functions recognise as differentiable inputs)
    static torch::autograd::variable_list forward(torch::autograd::AutogradContext* ctx, at::TensorList tensors,
                                                  std::vector<int64_t> axes, int64_t qmin, int64_t qmax, int64_t tmin,
                                                  int64_t tmax, bool use_gs, double gs, bool sym, bool eval_mode, bool init_mode) {
        at::AutoDispatchBelowADInplaceOrView below;
        const size_t n = tensors.size() / 3;
        torch::autograd::variable_list ys = forward_per_channel_multi(tensors.slice(0, n), tensors.slice(n, n), tensors.slice(2 * n, n), axes,
                                                                      qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode);
        ctx->save_for_backward(tensors.vec());
        ctx->set_materialize_grads(false);      // an unused output arrives undefined in backward, not as a zero tensor
        const int64_t flags = (use_gs ? 1 : 0) | (sym ? 2 : 0) | (eval_mode ? 8 : 0) | (init_mode ? 16 : 0);
        ctx->saved_data["cfg"] = c10::IValue(std::vector<int64_t>{qmin, qmax, tmin, tmax, flags});
        ctx->saved_data["axes"] = c10::IValue(axes);
        ctx->saved_data["gs"] = gs;
        return ys;
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
        const auto saved = ctx->get_saved_variables();
        const size_t n = saved.size() / 3;
        const std::vector<int64_t> cfg = ctx->saved_data["cfg"].toIntVector();
        const std::vector<int64_t> axes = ctx->saved_data["axes"].toIntVector();
        const int64_t flags = cfg[4];
        const Scalars s{cfg[0], cfg[1], cfg[2], cfg[3], (flags & 1) != 0, ctx->saved_data["gs"].toDouble(), (flags & 2) != 0,
                        (flags & 8) != 0, (flags & 16) != 0};
        // An output nobody used has no gradient: its tensor takes no part in the launch and gets NO gradients -- what N separate
        // lsq calls give it (with init_mode the parameter gradients ignore the upstream one, lsq_kernel.h:116, so a zero-filled
        // stand-in would invent d_scale / d_shift for it; and it would cost a fill plus the tensor's share of the kernel).
        torch::autograd::variable_list out(3 * n + 10);
        std::vector<size_t> live;
        live.reserve(n);
        for (size_t i = 0; i < n; ++i)
            if (grads[i].defined()) live.push_back(i);
        if (live.empty()) return out;
        const size_t m = live.size();
        std::vector<Tensor> gl(m), xl(m), sl(m), bl(m);
        std::vector<int64_t> al(m);
        for (size_t k = 0; k < m; ++k) {
            const size_t i = live[k];
            gl[k] = grads[i]; xl[k] = saved[i]; sl[k] = saved[n + i]; bl[k] = saved[2 * n + i]; al[k] = axes[i];
        }
        const std::vector<Tensor> r = backward_per_channel_multi(gl, xl, sl, bl, al, s.qmin, s.qmax, s.tmin, s.tmax, s.use_gs, s.gs, s.sym,
                                                                 s.eval_mode, s.init_mode);
        for (size_t k = 0; k < m; ++k) {
            out[live[k]] = r[k];
            out[n + live[k]] = r[m + k];
            out[2 * n + live[k]] = r[2 * m + k];
        }
        return out;
    }
};

Tensor lsq_impl(const Tensor& x, const Tensor& scale, const Tensor& shift, int64_t qmin, int64_t qmax, int64_t tmin,
                int64_t tmax, int64_t axis, bool use_gs, double gs, bool is_affine, bool is_perchannel, bool eval_mode,
                bool init_mode, bool mask_backward);

// `lsq(x_i, scale_i, shift_i, ..., is_perchannel = true)` for every i, horizontally fused: the tensors that qualify
// (lsq_hip_per_channel_multi_ok: one GPU, one storage type, contiguous, 16-byte aligned, channel rows the single-tensor policy
// gives one workgroup each) share ONE autograd node and one launch per 32 of them each way; the others go through `lsq`.
std::vector<Tensor> lsq_foreach(at::TensorList xs, at::TensorList scales, at::TensorList shifts, at::IntArrayRef axes, int64_t qmin,
                                int64_t qmax, int64_t tmin, int64_t tmax, bool use_gs, double gs, bool is_affine, bool eval_mode,
                                bool init_mode) {
    const size_t n = xs.size();
    TORCH_CHECK(scales.size() == n && shifts.size() == n && axes.size() == n, "lsq_foreach: xs, scales, shifts and axes must have the same length");
    std::vector<Tensor> out(n);
    std::vector<size_t> fused;
    fused.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        const Tensor& x = xs[i];
        bool ok = x.is_cuda() && scales[i].is_cuda() && shifts[i].is_cuda() && x.numel() > 0 && x.is_contiguous() &&
                  axes[i] >= 0 && axes[i] < x.dim() && scales[i].dim() == 1 && shifts[i].dim() == 1 &&
                  (reinterpret_cast<uintptr_t>(x.data_ptr()) & 15u) == 0;
        if (ok && !fused.empty()) ok = x.device() == xs[fused[0]].device() && x.scalar_type() == xs[fused[0]].scalar_type();
        if (ok) {
            int code = -1;
            switch (x.scalar_type()) {
                case at::kFloat: code = LSQ_F32; break;
                case at::kDouble: code = LSQ_F64; break;
                case at::kBFloat16: code = LSQ_BF16; break;
                case at::kHalf: code = LSQ_F16; break;
                default: ok = false;
            }
            if (ok) {
                const Geometry g = geometry(x, axes[i]);
                c10::DeviceGuard guard(x.device());        // the answer depends on the device's CU count
                ok = lsq_hip_per_channel_multi_ok(code, g.outer, g.channels, g.inner, 1) != 0;
            }
        }
        if (ok) fused.push_back(i);
        else out[i] = lsq_impl(x, scales[i], shifts[i], qmin, qmax, tmin, tmax, axes[i], use_gs, gs, is_affine, true, eval_mode, init_mode, true);
    }
    if (fused.size() == 1) {
        const size_t i = fused[0];
        out[i] = lsq_impl(xs[i], scales[i], shifts[i], qmin, qmax, tmin, tmax, axes[i], use_gs, gs, is_affine, true, eval_mode, init_mode, true);
        fused.clear();
    }
    if (fused.empty()) return out;
    const size_t m = fused.size();
    torch::autograd::variable_list tensors;
    std::vector<int64_t> ax;
    tensors.reserve(3 * m);
    ax.reserve(m);
    for (size_t i : fused) { tensors.push_back(xs[i]); ax.push_back(axes[i]); }
    for (size_t i : fused) {      // front-op rule (lsq.cpp:124-126): a size-1 parameter is repeated up to the other's size
        const int64_t size = std::max(scales[i].size(0), shifts[i].size(0));
        tensors.push_back(scales[i].size(0) == size ? scales[i] : scales[i].repeat({size}));
    }
    for (size_t i : fused) {
        const int64_t size = std::max(scales[i].size(0), shifts[i].size(0));
        tensors.push_back(shifts[i].size(0) == size ? shifts[i] : shifts[i].repeat({size}));
    }
    const auto ys = LsqForeachNode::apply(at::TensorList(tensors), ax, qmin, qmax, tmin, tmax, use_gs, gs, !is_affine, eval_mode, init_mode);
    for (size_t k = 0; k < m; ++k) out[fused[k]] = ys[k];
    return out;
}

Tensor lsq(const Tensor& x, const Tensor& scale, const Tensor& shift, int64_t qmin, int64_t qmax, int64_t tmin, int64_t tmax,
           int64_t axis, bool use_gs, double gs, bool is_affine, bool is_perchannel, bool eval_mode, bool init_mode) {
    return lsq_impl(x, scale, shift, qmin, qmax, tmin, tmax, axis, use_gs, gs, is_affine, is_perchannel, eval_mode, init_mode, true);
}

// the reference's eval-mode backward to the letter: x is saved, the mask is recomputed from the parameters as they are at
// backward time (lsq_autograd.cpp:46-73) -- what LSQFakeQuantizer asks for while its observer rewrites them on every call
Tensor lsq_keep_input(const Tensor& x, const Tensor& scale, const Tensor& shift, int64_t qmin, int64_t qmax, int64_t tmin,
                      int64_t tmax, int64_t axis, bool use_gs, double gs, bool is_affine, bool is_perchannel, bool eval_mode,
                      bool init_mode) {
    return lsq_impl(x, scale, shift, qmin, qmax, tmin, tmax, axis, use_gs, gs, is_affine, is_perchannel, eval_mode, init_mode, false);
}

}  // namespace

#define LSQ_TAIL \
    "int quant_min, int quant_max, int type_min, int type_max, bool use_grad_scaling, float grad_scaler, " \
    "bool sym, bool eval_mode, bool init_mode"

TORCH_LIBRARY(torchlsq_native, m) {
    m.def("lsq_forward_per_tensor(Tensor x, Tensor scale, Tensor shift, " LSQ_TAIL ") -> Tensor");
    m.def("lsq_backward_per_tensor(Tensor grad, Tensor x, Tensor scale, Tensor shift, " LSQ_TAIL ") -> (Tensor, Tensor, Tensor)");
    m.def("lsq_forward_per_channel(Tensor x, Tensor scale, Tensor shift, int axis, " LSQ_TAIL ") -> Tensor");
    m.def("lsq_backward_per_channel(Tensor grad, Tensor x, Tensor scale, Tensor shift, int axis, " LSQ_TAIL
          ") -> (Tensor, Tensor, Tensor)");
    m.def("lsq_backward_per_tensor_wide(Tensor grad, Tensor x, Tensor scale, Tensor shift, " LSQ_TAIL
          ", int numel_for_scaler) -> (Tensor, Tensor)");
    m.def("lsq_backward_per_channel_wide(Tensor grad, Tensor x, Tensor scale, Tensor shift, int axis, " LSQ_TAIL
          ", int numel_for_scaler) -> (Tensor, Tensor)");
    m.def("lsq_backward_per_tensor_sharded(Tensor grad, Tensor x, Tensor scale, Tensor shift, " LSQ_TAIL
          ", int numel_for_scaler, int comm) -> (Tensor, Tensor, Tensor, int)");
    m.def("lsq_backward_per_channel_sharded(Tensor grad, Tensor x, Tensor scale, Tensor shift, int axis, " LSQ_TAIL
          ", int numel_for_scaler, int comm) -> (Tensor, Tensor, Tensor, int)");
    m.def("lsq_backward_packed(Tensor grad, Tensor x, Tensor scale, Tensor shift, bool per_channel, int axis, int quant_min, "
          "int quant_max, int type_min, int type_max, bool sym, bool init_mode) -> (Tensor, Tensor)");
    m.def("lsq_sharded_finish(Tensor packed, int channels, bool per_channel, int x_dtype_code, int quant_max, bool use_grad_scaling, "
          "float grad_scaler) -> (Tensor, Tensor)");
    m.def("lsq_backward_from_mask(Tensor grad, Tensor mask) -> Tensor");
    // composite (autograd handled by the node inside), like the reference's front op
    m.def("lsq(Tensor x, Tensor scale, Tensor shift, int quant_min, int quant_max, int type_min, int type_max, int axis, "
          "bool use_grad_scaling, float grad_scale, bool is_affine, bool is_perchannel, bool eval_mode, bool init_mode) -> Tensor",
          &lsq);
    m.def("lsq_keep_input(Tensor x, Tensor scale, Tensor shift, int quant_min, int quant_max, int type_min, int type_max, int axis, "
          "bool use_grad_scaling, float grad_scale, bool is_affine, bool is_perchannel, bool eval_mode, bool init_mode) -> Tensor",
          &lsq_keep_input);
    m.def("lsq_foreach(Tensor[] xs, Tensor[] scales, Tensor[] shifts, int[] axes, int quant_min, int quant_max, int type_min, "
          "int type_max, bool use_grad_scaling, float grad_scale, bool is_affine, bool eval_mode, bool init_mode) -> Tensor[]",
          &lsq_foreach);
    m.def("lsq_forward_per_channel_multi(Tensor[] xs, Tensor[] scales, Tensor[] shifts, int[] axes, " LSQ_TAIL ") -> Tensor[]");
    m.def("lsq_backward_per_channel_multi(Tensor[] grads, Tensor[] xs, Tensor[] scales, Tensor[] shifts, int[] axes, " LSQ_TAIL
          ") -> Tensor[]");
    m.def("_abi_version() -> int", []() -> int64_t { return lsq_hip_abi_version(); });
    // what this layer would decide (tests/test_host_policy.py holds it against the Python layer's decisions)
    m.def("_policy_probe(int per_channel, int tensor_bytes, bool eval_mode, bool init_mode, bool requires_grad, bool mask_backward) -> int[]",
          [](int64_t per_channel, int64_t bytes, bool eval_mode, bool init_mode, bool rg, bool mb) -> std::vector<int64_t> {
              return {lsq_hip_policy_ticket(g_ticket_mode.load(), per_channel ? 1 : 0, bytes),
                      lsq_hip_policy_saves_mask(eval_mode, init_mode, rg, mb), g_ticket_mode.load()};
          });
    m.def("_set_single_launch_backward(int mode) -> ()", [](int64_t mode) { g_ticket_mode.store(mode < 0 || mode > 2 ? 2 : static_cast<int>(mode)); });
}

TORCH_LIBRARY_IMPL(torchlsq_native, CUDA, m) {  // PyTorch-ROCm dispatches HIP tensors under the CUDA key
    m.impl("lsq_forward_per_tensor", &forward_per_tensor);
    m.impl("lsq_backward_per_tensor", &backward_per_tensor);
    m.impl("lsq_forward_per_channel", &forward_per_channel);
    m.impl("lsq_backward_per_channel", &backward_per_channel);
    m.impl("lsq_backward_per_tensor_wide", &backward_per_tensor_wide);
    m.impl("lsq_backward_per_channel_wide", &backward_per_channel_wide);
    m.impl("lsq_backward_per_tensor_sharded", &backward_per_tensor_sharded);
    m.impl("lsq_backward_per_channel_sharded", &backward_per_channel_sharded);
    m.impl("lsq_backward_packed", &backward_packed);
    m.impl("lsq_sharded_finish", &sharded_finish);
    m.impl("lsq_backward_from_mask", &backward_from_mask);
    m.impl("lsq_forward_per_channel_multi", &forward_per_channel_multi);
    m.impl("lsq_backward_per_channel_multi", &backward_per_channel_multi);
}
