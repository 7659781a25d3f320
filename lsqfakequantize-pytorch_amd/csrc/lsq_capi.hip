// lsq_capi.hip -- the extern "C" boundary declared in include/lsq_hip.h.
//
// Thin by design: argument validation, dtype dispatch, error bookkeeping.  No tensor memory is
// allocated or freed here and no state outlives a call (the only statics are immutable-once-filled lookup tables -- device
// properties, kernel register counts, workspace sizes per shape -- and a thread-local error string).
// The production library exports exactly the symbols of include/lsq_hip.h.  The `_ex` twins (a trailing launch-variant
// code) and the lsq_hip_debug_* knobs of lsq_internal.h exist only in the tools build (-DLSQ_TOOLS, `make tools`).
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <unordered_map>

#include <hip/hip_version.h>

#ifdef LSQ_TOOLS
#include "lsq_internal.h"
#endif
#include "lsq_kernels.hpp"

namespace {

thread_local char g_last_error[512] = "";
}  // namespace

namespace lsq {
char* error_buffer() { return g_last_error; }      // lsq_comm.hip reports through the same thread-local message
}  // namespace lsq

namespace {

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
    return code;
}

int hip_status(hipError_t e, const char* what) {
    if (e == hipSuccess) return LSQ_OK;
    return fail(static_cast<int>(e), "%s: %s (%s)", what, hipGetErrorName(e), hipGetErrorString(e));
}

bool dtype_ok(int dtype) { return dtype >= LSQ_F32 && dtype <= LSQ_F16; }
int io_vec(int dtype) { return dtype == LSQ_F32 ? 4 : dtype == LSQ_F64 ? 2 : 8; }

int check_common(int dtype, const lsq_params* p) {
    if (!dtype_ok(dtype)) return fail(LSQ_EINVAL, "unknown dtype code %d", dtype);
    if (!p) return fail(LSQ_EINVAL, "lsq_params pointer is NULL");
    if (p->quant_min > p->quant_max) return fail(LSQ_EINVAL, "quant_min %d > quant_max %d", p->quant_min, p->quant_max);
    if (p->type_min > p->type_max) return fail(LSQ_EINVAL, "type_min %d > type_max %d", p->type_min, p->type_max);
    return LSQ_OK;
}

int check_levels(const lsq_params* p, const lsq_fwd_extras* ex) {
    if (ex && ex->levels && ex->aux_kind == 0) {
        const int lo = p->quant_min - ex->level_bias, hi = p->quant_max - ex->level_bias;
        if (!((lo >= -128 && hi <= 127) || (lo >= 0 && hi <= 255)))      // the byte is (q - level_bias) mod 256
            return fail(LSQ_EINVAL, "levels: [quant_min, quant_max] - level_bias = [%d, %d] fits neither int8 nor uint8", lo, hi);
    }
    return LSQ_OK;
}

}  // namespace

#define LSQ_DISPATCH_IO(dtype, CALL)                          \
    switch (dtype) {                                          \
        case LSQ_F32: { using IO = lsq::io_f32; CALL; } break;   \
        case LSQ_F64: { using IO = lsq::io_f64; CALL; } break;   \
        case LSQ_BF16: { using IO = lsq::io_bf16; CALL; } break; \
        default: { using IO = lsq::io_f16; CALL; } break;        \
    }

// the four ops with a launch-variant code: exported (extern "C") by the tools build only, file-local otherwise
#ifdef LSQ_TOOLS
#define LSQ_EX_LINKAGE
#else
#define LSQ_EX_LINKAGE static
#endif

extern "C" {

int lsq_hip_abi_version(void) { return LSQ_HIP_ABI_VERSION; }

int64_t lsq_hip_runtime_version(void) { return static_cast<int64_t>(HIP_VERSION); }

const char* lsq_hip_last_error(void) { return g_last_error; }

double lsq_hip_grad_scaler(int dtype, int per_channel, int64_t numel, int32_t quant_max, int64_t channels,
                           int32_t use_grad_scaling, double grad_scaler) {
    const bool use = use_grad_scaling != 0;
    if (dtype == LSQ_F64) {
        return per_channel ? lsq::grad_scaler_per_channel<double>(numel, quant_max, channels, use, grad_scaler)
                           : lsq::grad_scaler_per_tensor<double>(numel, quant_max, use, grad_scaler);
    }
    return per_channel
               ? static_cast<double>(lsq::grad_scaler_per_channel<float>(numel, quant_max, channels, use, grad_scaler))
               : static_cast<double>(lsq::grad_scaler_per_tensor<float>(numel, quant_max, use, grad_scaler));
}

int lsq_hip_policy_ticket(int32_t mode, int32_t per_channel, int64_t tensor_bytes) {
    constexpr int64_t kTicketAutoBytes = int64_t{8} << 20;      // profiles/r03_ticket_sizes.txt
    if (mode == 1) return 1;
    if (mode == 2) return (!per_channel && tensor_bytes <= kTicketAutoBytes) ? 1 : 0;
    return 0;
}

int lsq_hip_policy_saves_mask(int32_t eval_mode, int32_t init_mode, int32_t input_requires_grad, int32_t mask_backward) {
    return (eval_mode && !init_mode && input_requires_grad && mask_backward) ? 1 : 0;
}

size_t lsq_hip_backward_per_tensor_workspace(int dtype, int64_t n) {
    (void)dtype;
    (void)n;
    return lsq::bwd_pt_workspace_bytes();
}

size_t lsq_hip_backward_per_channel_workspace(int dtype, int64_t outer, int64_t channels, int64_t inner) {
    if (!dtype_ok(dtype) || outer <= 0 || channels <= 0 || inner <= 0) return 256;
    // The size is what the launch policy will ask for (lsq::bwd_pc_workspace_bytes: the policy run as a plan); callers ask
    // once per backward with the same few shapes, so the answers of this thread are kept -- every distinct
    // (device, dtype, shape) of a model, not just the last few.  (The answer depends on the CU count of the current device.)
    struct Key {
        int dtype, device, knob;
        int64_t outer, channels, inner;
        bool operator==(const Key& o) const {
            return dtype == o.dtype && device == o.device && knob == o.knob && outer == o.outer && channels == o.channels && inner == o.inner;
        }
    };
    struct Hash {
        size_t operator()(const Key& k) const {
            uint64_t h = 1469598103934665603ull;
            for (uint64_t v : {static_cast<uint64_t>(k.dtype), static_cast<uint64_t>(k.device), static_cast<uint64_t>(k.knob),
                               static_cast<uint64_t>(k.outer), static_cast<uint64_t>(k.channels), static_cast<uint64_t>(k.inner)})
                h = (h ^ v) * 1099511628211ull;
            return static_cast<size_t>(h);
        }
    };
    thread_local std::unordered_map<Key, size_t, Hash> memo;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) device = 0;
    const Key key{dtype, device, lsq::knob::geometry_key(), outer, channels, inner};
    const auto hit = memo.find(key);
    if (hit != memo.end()) return hit->second;
    size_t bytes = 0;
    LSQ_DISPATCH_IO(dtype, bytes = lsq::bwd_pc_workspace_bytes<IO>(outer, channels, inner));
    if (memo.size() >= 65536) memo.clear();
    memo.emplace(key, bytes);
    return bytes;
}

LSQ_EX_LINKAGE int lsq_hip_forward_per_tensor_ex(int dtype, const void* x, void* y, int64_t n, const void* scale, const void* shift,
                                  const lsq_params* p, const lsq_fwd_extras* extras, void* stream, int variant) {
    if (int rc = check_common(dtype, p)) return rc;
    if (n < 0) return fail(LSQ_EINVAL, "negative element count %lld", static_cast<long long>(n));
    if (n == 0) return LSQ_OK;
    if (!x || !scale || !shift || (!y && !(extras && extras->levels))) return fail(LSQ_EINVAL, "forward_per_tensor: NULL buffer");
    if (int rc = check_levels(p, extras)) return rc;
    hipError_t e = hipSuccess;
    LSQ_DISPATCH_IO(dtype, e = lsq::forward_per_tensor<IO>(x, y, n, scale, shift, *p, extras, variant,
                                                            static_cast<hipStream_t>(stream)));
    return hip_status(e, "lsq_hip_forward_per_tensor");
}

int lsq_hip_forward_per_tensor(int dtype, const void* x, void* y, int64_t n, const void* scale, const void* shift,
                               const lsq_params* p, const lsq_fwd_extras* extras, void* stream) {
    return lsq_hip_forward_per_tensor_ex(dtype, x, y, n, scale, shift, p, extras, stream, 0);
}

LSQ_EX_LINKAGE int lsq_hip_backward_per_tensor_ex(int dtype, const void* grad, const void* x, void* dx, void* ds, void* db,
                                   double* dsdb_wide, int64_t n, const void* scale, const void* shift,
                                   const lsq_params* p, const lsq_bwd_extras* extras, void* workspace,
                                   size_t workspace_bytes, void* stream, int variant) {
    if (int rc = check_common(dtype, p)) return rc;
    if (n <= 0) return fail(LSQ_EINVAL, "backward_per_tensor: element count must be positive (the caller handles the "
                                        "empty case, reference lsq_cpu.cpp:76-78)");
    if (!grad || !x || !dx || !ds || !db || !scale || !shift) return fail(LSQ_EINVAL, "backward_per_tensor: NULL buffer");
    if (!workspace || workspace_bytes < lsq::bwd_pt_workspace_bytes())
        return fail(LSQ_EWORKSPACE, "backward_per_tensor: workspace of %zu bytes, need %zu", workspace_bytes,
                    lsq::bwd_pt_workspace_bytes());
    if (reinterpret_cast<uintptr_t>(workspace) & 15u) return fail(LSQ_EWORKSPACE, "workspace must be 16-byte aligned");
    uint32_t* ticket = extras ? static_cast<uint32_t*>(extras->ticket) : nullptr;
    if (reinterpret_cast<uintptr_t>(ticket) & 3u) return fail(LSQ_EINVAL, "ticket must be 4-byte aligned");
    hipError_t e = hipSuccess;
    LSQ_DISPATCH_IO(dtype, e = lsq::backward_per_tensor<IO>(grad, x, dx, ds, db, dsdb_wide, n, scale, shift, *p,
                                                             workspace, ticket, variant, static_cast<hipStream_t>(stream)));
    return hip_status(e, "lsq_hip_backward_per_tensor");
}

int lsq_hip_backward_per_tensor(int dtype, const void* grad, const void* x, void* dx, void* ds, void* db,
                                double* dsdb_wide, int64_t n, const void* scale, const void* shift,
                                const lsq_params* p, const lsq_bwd_extras* extras, void* workspace,
                                size_t workspace_bytes, void* stream) {
    return lsq_hip_backward_per_tensor_ex(dtype, grad, x, dx, ds, db, dsdb_wide, n, scale, shift, p, extras, workspace,
                                          workspace_bytes, stream, 0);
}

static int check_ocl(int64_t outer, int64_t channels, int64_t inner) {
    if (outer < 0 || channels <= 0 || inner < 0)
        return fail(LSQ_EINVAL, "bad [outer, C, inner] = [%lld, %lld, %lld]", static_cast<long long>(outer),
                    static_cast<long long>(channels), static_cast<long long>(inner));
    return LSQ_OK;
}

LSQ_EX_LINKAGE int lsq_hip_forward_per_channel_ex(int dtype, const void* x, void* y, int64_t outer, int64_t channels, int64_t inner,
                                   const void* scale, const void* shift, const lsq_params* p,
                                   const lsq_fwd_extras* extras, void* stream, int variant) {
    if (int rc = check_common(dtype, p)) return rc;
    if (int rc = check_ocl(outer, channels, inner)) return rc;
    if (outer == 0 || inner == 0) return LSQ_OK;
    if (!x || !scale || !shift || (!y && !(extras && extras->levels))) return fail(LSQ_EINVAL, "forward_per_channel: NULL buffer");
    if (int rc = check_levels(p, extras)) return rc;
    hipError_t e = hipSuccess;
    LSQ_DISPATCH_IO(dtype, e = lsq::forward_per_channel<IO>(x, y, outer, channels, inner, scale, shift, *p, extras,
                                                             variant, static_cast<hipStream_t>(stream)));
    return hip_status(e, "lsq_hip_forward_per_channel");
}

int lsq_hip_forward_per_channel(int dtype, const void* x, void* y, int64_t outer, int64_t channels, int64_t inner,
                                const void* scale, const void* shift, const lsq_params* p,
                                const lsq_fwd_extras* extras, void* stream) {
    return lsq_hip_forward_per_channel_ex(dtype, x, y, outer, channels, inner, scale, shift, p, extras, stream, 0);
}

LSQ_EX_LINKAGE int lsq_hip_backward_per_channel_ex(int dtype, const void* grad, const void* x, void* dx, void* ds, void* db,
                                    double* dsdb_wide, int64_t outer, int64_t channels, int64_t inner,
                                    const void* scale, const void* shift, const lsq_params* p,
                                    const lsq_bwd_extras* extras, void* workspace, size_t workspace_bytes, void* stream,
                                    int variant) {
    if (int rc = check_common(dtype, p)) return rc;
    if (int rc = check_ocl(outer, channels, inner)) return rc;
    if (outer == 0 || inner == 0)
        return fail(LSQ_EINVAL, "backward_per_channel: empty tensor (the caller handles it, reference lsq_cpu.cpp:221-223)");
    if (!grad || !x || !dx || !ds || !db || !scale || !shift) return fail(LSQ_EINVAL, "backward_per_channel: NULL buffer");
    if (!workspace) return fail(LSQ_EWORKSPACE, "backward_per_channel: NULL workspace");
    if (reinterpret_cast<uintptr_t>(workspace) & 15u) return fail(LSQ_EWORKSPACE, "workspace must be 16-byte aligned");
    (void)extras;      // lsq_bwd_extras.ticket: accepted and ignored by the per-channel backward (include/lsq_hip.h)
    hipError_t e = hipSuccess;
    LSQ_DISPATCH_IO(dtype, e = lsq::backward_per_channel<IO>(grad, x, dx, ds, db, dsdb_wide, outer, channels, inner,
                                                              scale, shift, *p, workspace, workspace_bytes, nullptr, variant,
                                                              static_cast<hipStream_t>(stream)));
    if (e == hipErrorInvalidValue)
        return fail(LSQ_EWORKSPACE, "backward_per_channel: workspace of %zu bytes is too small (ask "
                                    "lsq_hip_backward_per_channel_workspace)", workspace_bytes);
    return hip_status(e, "lsq_hip_backward_per_channel");
}

int lsq_hip_backward_per_channel(int dtype, const void* grad, const void* x, void* dx, void* ds, void* db,
                                 double* dsdb_wide, int64_t outer, int64_t channels, int64_t inner,
                                 const void* scale, const void* shift, const lsq_params* p,
                                 const lsq_bwd_extras* extras, void* workspace, size_t workspace_bytes, void* stream) {
    return lsq_hip_backward_per_channel_ex(dtype, grad, x, dx, ds, db, dsdb_wide, outer, channels, inner, scale, shift,
                                           p, extras, workspace, workspace_bytes, stream, 0);
}

int lsq_hip_plan_backward_per_channel(int dtype, int64_t outer, int64_t channels, int64_t inner, int aligned16,
                                      const lsq_params* p, int32_t* out8) {
    if (int rc = check_common(dtype, p)) return rc;
    if (outer <= 0 || channels <= 0 || inner <= 0 || !out8)
        return fail(LSQ_EINVAL, "plan_backward_per_channel: positive [outer, C, inner] and an output array");
    lsq::LaunchNote note{0, 0, 0, 0, 0, 0, 0, 0};
    size_t need = 0;
    hipError_t e = hipSuccess;
    // only the alignment of the (never dereferenced) buffer addresses matters to a plan
    LSQ_DISPATCH_IO(dtype, {
        void* const fake = reinterpret_cast<void*>(static_cast<uintptr_t>(aligned16 ? 4096 : 4096 + sizeof(typename IO::elem)));
        e = lsq::backward_per_channel<IO>(fake, fake, fake, fake, fake, nullptr, outer, channels, inner, fake, fake, *p, fake, 0,
                                          nullptr, 0, nullptr, &need, &note);
    });
    if (int rc = hip_status(e, "lsq_hip_plan_backward_per_channel")) return rc;
    out8[0] = note.grid_x; out8[1] = note.grid_y; out8[2] = note.resident_per_cu; out8[3] = note.vgprs_hint;
    out8[4] = note.kind; out8[5] = note.dma_depth; out8[6] = note.block; out8[7] = note.ring_nt;
    return LSQ_OK;
}

int lsq_hip_per_channel_multi_ok(int dtype, int64_t outer, int64_t channels, int64_t inner, int aligned16) {
    if (!dtype_ok(dtype)) return 0;
    bool ok = false;
    LSQ_DISPATCH_IO(dtype, ok = lsq::multi_eligible<IO>(outer, channels, inner, aligned16 != 0));
    return ok ? 1 : 0;
}

static int multi_call(bool backward, int dtype, const lsq_pc_item* items, int32_t count, const lsq_params* p, void* stream) {
    const char* what = backward ? "lsq_hip_backward_per_channel_multi" : "lsq_hip_forward_per_channel_multi";
    if (int rc = check_common(dtype, p)) return rc;
    if (count < 0) return fail(LSQ_EINVAL, "%s: negative item count", what);
    if (count == 0) return LSQ_OK;
    if (!items) return fail(LSQ_EINVAL, "%s: NULL item table", what);
    if (p->numel_for_scaler > 0) return fail(LSQ_EINVAL, "%s: numel_for_scaler must be <= 0 (every tensor uses its own element count)", what);
    for (int32_t i = 0; i < count; ++i) {
        const lsq_pc_item& it = items[i];
        if (!it.x || !it.scale || !it.shift || (backward ? (!it.grad || !it.dx || !it.ds || !it.db) : !it.y))
            return fail(LSQ_EINVAL, "%s: item %d has a NULL buffer", what, i);
        const bool aligned = lsq::is_aligned16(it.x) && (backward ? lsq::is_aligned16(it.grad) && lsq::is_aligned16(it.dx) : lsq::is_aligned16(it.y));
        if (!lsq_hip_per_channel_multi_ok(dtype, it.outer, it.channels, it.inner, aligned ? 1 : 0))
            return fail(LSQ_EINVAL, "%s: item %d ([%lld, %lld, %lld]) cannot take part in a multi-tensor launch "
                                    "(lsq_hip_per_channel_multi_ok); use the single-tensor entry point", what, i,
                        static_cast<long long>(it.outer), static_cast<long long>(it.channels), static_cast<long long>(it.inner));
    }
    hipError_t e = hipSuccess;
    if (backward) { LSQ_DISPATCH_IO(dtype, e = lsq::backward_per_channel_multi<IO>(items, count, *p, static_cast<hipStream_t>(stream))); }
    else { LSQ_DISPATCH_IO(dtype, e = lsq::forward_per_channel_multi<IO>(items, count, *p, static_cast<hipStream_t>(stream))); }
    return hip_status(e, what);
}

int lsq_hip_forward_per_channel_multi(int dtype, const lsq_pc_item* items, int32_t count, const lsq_params* p, void* stream) {
    return multi_call(false, dtype, items, count, p, stream);
}

int lsq_hip_backward_per_channel_multi(int dtype, const lsq_pc_item* items, int32_t count, const lsq_params* p, void* stream) {
    return multi_call(true, dtype, items, count, p, stream);
}

int lsq_hip_backward_from_mask(int dtype, const void* grad, const void* mask, void* dx, int64_t n, void* stream) {
    if (!dtype_ok(dtype)) return fail(LSQ_EINVAL, "unknown dtype code %d", dtype);
    if (n < 0) return fail(LSQ_EINVAL, "negative element count");
    if (n == 0) return LSQ_OK;
    if (!grad || !mask || !dx) return fail(LSQ_EINVAL, "backward_from_mask: NULL buffer");
    hipError_t e = hipSuccess;
    LSQ_DISPATCH_IO(dtype, e = lsq::backward_from_mask<IO>(grad, mask, dx, n, static_cast<hipStream_t>(stream)));
    return hip_status(e, "lsq_hip_backward_from_mask");
}

int lsq_hip_sharded_finish(int dtype, const double* packed, int64_t channels, int32_t per_channel, const lsq_params* p,
                           void* ds, void* db, void* stream) {
    if (int rc = check_common(dtype, p)) return rc;
    if (channels <= 0) return fail(LSQ_EINVAL, "sharded_finish: channel count must be positive");
    if (!per_channel && channels != 1) return fail(LSQ_EINVAL, "sharded_finish: a per-tensor quantizer has one channel");
    if (!packed || !ds || !db) return fail(LSQ_EINVAL, "sharded_finish: NULL buffer");
    if (reinterpret_cast<uintptr_t>(packed) & 7u) return fail(LSQ_EINVAL, "sharded_finish: packed must be 8-byte aligned");
    hipError_t e = dtype == LSQ_F64
                       ? lsq::sharded_finish<double>(packed, channels, per_channel != 0, *p, ds, db, static_cast<hipStream_t>(stream))
                       : lsq::sharded_finish<float>(packed, channels, per_channel != 0, *p, ds, db, static_cast<hipStream_t>(stream));
    return hip_status(e, "lsq_hip_sharded_finish");
}

int lsq_hip_relayout(int dtype, const void* src, void* dst, int64_t a, int64_t b, int64_t c, void* stream) {
    if (dtype < LSQ_F32 || dtype > LSQ_F16) return fail(LSQ_EINVAL, "relayout: unknown dtype %d", dtype);
    if (a < 0 || b < 0 || c < 0) return fail(LSQ_EINVAL, "relayout: negative extent");
    if (a == 0 || b == 0 || c == 0) return LSQ_OK;
    if (!src || !dst) return fail(LSQ_EINVAL, "relayout: NULL buffer");
    const int esz = dtype == LSQ_F64 ? 8 : (dtype == LSQ_F32 ? 4 : 2);
    if ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & static_cast<uintptr_t>(esz - 1))
        return fail(LSQ_EINVAL, "relayout: buffers must be element-aligned");
    return hip_status(lsq::relayout(esz, src, dst, a, b, c, static_cast<hipStream_t>(stream)), "lsq_hip_relayout");
}

#ifdef LSQ_TOOLS
void lsq_hip_debug_set_observe_wg_per_cu(int v) { lsq::knob::set(lsq::knob::kObserveWgPerCu, v); }
void lsq_hip_debug_force_ring(int v) { lsq::knob::set(lsq::knob::kForceRing, v < 0 || v > 2 ? 0 : v); }
void lsq_hip_debug_set_ww_min_rows(int v) { lsq::knob::set(lsq::knob::kWwMinRows, v); }
void lsq_hip_debug_set_ww_split64(int v) { lsq::knob::set(lsq::knob::kWwSplit64, v); }
void lsq_hip_debug_set_ww_big(int v) { lsq::knob::set(lsq::knob::kWwBig, v); }
void lsq_hip_debug_set_ring_nt(int v) { lsq::knob::set(lsq::knob::kRingNt, v); }
void lsq_hip_debug_set_ww_max_log2(int v) { lsq::knob::set(lsq::knob::kWwMaxLog2, v < 0 || v > 40 ? 0 : v); }
void lsq_hip_debug_set_seg_min_div(int v) { lsq::knob::set(lsq::knob::kSegMinDiv, v < 0 || v > 16 ? 0 : v); }
void lsq_hip_debug_set_fwd_direct(int v) { lsq::knob::set(lsq::knob::kFwdDirect, v < 0 || v > 4 ? 0 : v); }
void lsq_hip_debug_set_seg_no_up_front(int v) { lsq::knob::set(lsq::knob::kSegNoUpFront, v ? 1 : 0); }
void lsq_hip_debug_set_fin_ch(int v) { lsq::knob::set(lsq::knob::kFinCh, v); }
void lsq_hip_debug_set_own_min_run(int v) { lsq::knob::set(lsq::knob::kOwnMinRun, v < 0 ? 0 : v); }
void lsq_hip_debug_set_own_fat(int v) { lsq::knob::set(lsq::knob::kOwnFat, v < 0 || v > 2 ? 0 : v); }
void lsq_hip_debug_set_own(int v) { lsq::knob::set(lsq::knob::kOwn, v < 0 || v > 3 ? 0 : v); }

#ifdef LSQ_TIMELINE
void lsq_hip_debug_set_timeline(void* device_buffer) { lsq::knob::timeline_buffer().store(static_cast<unsigned long long*>(device_buffer)); }
#endif

void lsq_hip_debug_last_launch(int* out8) {
    const lsq::LaunchNote& n = lsq::last_launch_note();
    out8[0] = n.grid_x; out8[1] = n.grid_y; out8[2] = n.resident_per_cu; out8[3] = n.vgprs_hint;
    out8[4] = n.kind; out8[5] = n.dma_depth; out8[6] = n.block; out8[7] = n.ring_nt;
}
#endif

size_t lsq_hip_minmax_workspace(int dtype, int64_t outer, int64_t channels, int64_t inner) {
    if (!dtype_ok(dtype) || outer <= 0 || channels <= 0 || inner <= 0) return 256;
    return lsq::minmax_workspace_bytes(io_vec(dtype), dtype == LSQ_F64 ? 8 : 4, outer, channels, inner);
}

int lsq_hip_minmax_per_tensor(int dtype, const void* x, int64_t n, void* min_out, void* max_out, void* workspace,
                              size_t workspace_bytes, void* stream) {
    if (!dtype_ok(dtype)) return fail(LSQ_EINVAL, "unknown dtype code %d", dtype);
    if (n <= 0) return fail(LSQ_EINVAL, "minmax_per_tensor: element count must be positive");
    if (!x || !min_out || !max_out) return fail(LSQ_EINVAL, "minmax_per_tensor: NULL buffer");
    if (!workspace || workspace_bytes < lsq::minmax_workspace_bytes(io_vec(dtype), dtype == LSQ_F64 ? 8 : 4, 1, 1, 1) - 256 ||
        (reinterpret_cast<uintptr_t>(workspace) & 15u))
        return fail(LSQ_EWORKSPACE, "minmax_per_tensor: workspace too small or misaligned (%zu bytes)", workspace_bytes);
    hipError_t e = hipSuccess;
    LSQ_DISPATCH_IO(dtype, e = lsq::minmax_per_tensor<IO>(x, n, min_out, max_out, workspace, static_cast<hipStream_t>(stream)));
    return hip_status(e, "lsq_hip_minmax_per_tensor");
}

int lsq_hip_minmax_per_channel(int dtype, const void* x, int64_t outer, int64_t channels, int64_t inner, void* min_out,
                               void* max_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!dtype_ok(dtype)) return fail(LSQ_EINVAL, "unknown dtype code %d", dtype);
    if (int rc = check_ocl(outer, channels, inner)) return rc;
    if (outer == 0 || inner == 0) return fail(LSQ_EINVAL, "minmax_per_channel: empty tensor");
    if (!x || !min_out || !max_out) return fail(LSQ_EINVAL, "minmax_per_channel: NULL buffer");
    if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 15u))
        return fail(LSQ_EWORKSPACE, "minmax_per_channel: NULL or misaligned workspace");
    hipError_t e = hipSuccess;
    LSQ_DISPATCH_IO(dtype, e = lsq::minmax_per_channel<IO>(x, outer, channels, inner, min_out, max_out, workspace,
                                                            workspace_bytes, static_cast<hipStream_t>(stream)));
    if (e == hipErrorInvalidValue)
        return fail(LSQ_EWORKSPACE, "minmax_per_channel: workspace of %zu bytes is too small", workspace_bytes);
    return hip_status(e, "lsq_hip_minmax_per_channel");
}

size_t lsq_hip_meanstd_workspace(int dtype, int64_t outer, int64_t channels, int64_t inner) {
    if (!dtype_ok(dtype) || outer <= 0 || channels <= 0 || inner <= 0) return 256;
    return lsq::meanstd_workspace_bytes(io_vec(dtype), outer, channels, inner);
}

int lsq_hip_meanstd_per_tensor(int dtype, const void* x, int64_t n, void* mean_out, void* std_out, void* workspace,
                               size_t workspace_bytes, void* stream) {
    if (!dtype_ok(dtype)) return fail(LSQ_EINVAL, "unknown dtype code %d", dtype);
    if (n <= 0) return fail(LSQ_EINVAL, "meanstd_per_tensor: element count must be positive");
    if (!x || !mean_out || !std_out) return fail(LSQ_EINVAL, "meanstd_per_tensor: NULL buffer");
    if (!workspace || workspace_bytes < lsq::meanstd_workspace_bytes(io_vec(dtype), 1, 1, 1) - 256 ||
        (reinterpret_cast<uintptr_t>(workspace) & 15u))
        return fail(LSQ_EWORKSPACE, "meanstd_per_tensor: workspace too small or misaligned (%zu bytes)", workspace_bytes);
    hipError_t e = hipSuccess;
    LSQ_DISPATCH_IO(dtype, e = lsq::meanstd_per_tensor<IO>(x, n, mean_out, std_out, workspace, static_cast<hipStream_t>(stream)));
    return hip_status(e, "lsq_hip_meanstd_per_tensor");
}

int lsq_hip_meanstd_per_channel(int dtype, const void* x, int64_t outer, int64_t channels, int64_t inner, void* mean_out,
                                void* std_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!dtype_ok(dtype)) return fail(LSQ_EINVAL, "unknown dtype code %d", dtype);
    if (int rc = check_ocl(outer, channels, inner)) return rc;
    if (outer == 0 || inner == 0) return fail(LSQ_EINVAL, "meanstd_per_channel: empty tensor");
    if (!x || !mean_out || !std_out) return fail(LSQ_EINVAL, "meanstd_per_channel: NULL buffer");
    if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 15u))
        return fail(LSQ_EWORKSPACE, "meanstd_per_channel: NULL or misaligned workspace");
    hipError_t e = hipSuccess;
    LSQ_DISPATCH_IO(dtype, e = lsq::meanstd_per_channel<IO>(x, outer, channels, inner, mean_out, std_out, workspace,
                                                             workspace_bytes, static_cast<hipStream_t>(stream)));
    if (e == hipErrorInvalidValue)
        return fail(LSQ_EWORKSPACE, "meanstd_per_channel: workspace of %zu bytes is too small", workspace_bytes);
    return hip_status(e, "lsq_hip_meanstd_per_channel");
}

int lsq_hip_observer_update(int64_t channels, const float* cur_min, const float* cur_max, float* min_state,
                            float* max_state, const lsq_observer_update* u, float* scale_out, float* shift_out, void* stream) {
    if (channels <= 0) return fail(LSQ_EINVAL, "observer_update: channel count must be positive");
    if (!cur_min || !cur_max || !min_state || !max_state || !u || !scale_out || !shift_out)
        return fail(LSQ_EINVAL, "observer_update: NULL argument");
    if (u->mode != 1 && u->mode != 2) return fail(LSQ_EINVAL, "observer_update: mode must be 1 (min/max) or 2 (moving average)");
    if (u->quant_min >= u->quant_max) return fail(LSQ_EINVAL, "observer_update: quant_min %d >= quant_max %d", u->quant_min, u->quant_max);
    return hip_status(lsq::observer_update(channels, cur_min, cur_max, min_state, max_state, *u, scale_out, shift_out,
                                           static_cast<hipStream_t>(stream)),
                      "lsq_hip_observer_update");
}

}  // extern "C"
