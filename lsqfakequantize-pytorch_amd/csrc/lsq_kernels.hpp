// lsq_kernels.hpp -- declarations shared by the kernel translation units and the C ABI layer.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/lsq_hip.h"
#include "lsq_math.hpp"

namespace lsq {

constexpr int kBlock = 256;          // 4 wave64 per workgroup
constexpr int kMaxCUs = 1024;        // upper bound used to size workspaces (MI355X: 256)
constexpr int kMaxBlocksPerCU = 16;

// ---- launch variants (tuning knobs; the defaults are what bench/profiles were measured with) ----
struct Variant {
    int unroll;         // 16-byte packets in flight per lane per stream
    bool nt_load;       // non-temporal loads  (global_load ... nt)
    bool nt_store;      // non-temporal stores (global_store ... nt)
    int blocks_per_cu;  // persistent-grid size = CUs * blocks_per_cu
    bool chunked;       // tile -> workgroup map: false = strided (tile t -> wg t % grid), true = contiguous chunks
    int dma;            // window-mode backward, 16-byte packets: 0 = the storage type's default, 1 = registers, 2 = LDS-DMA ring
};
// encoded as unroll | nt_load << 8 | nt_store << 9 | chunked << 10 | blocks_per_cu << 16 ; 0 = "use the default"
constexpr int encode_variant(int unroll, bool ntl, bool nts, int bpc) {
    return unroll | (ntl ? 1 << 8 : 0) | (nts ? 1 << 9 : 0) | (bpc << 16);
}
// Tuned on MI355X at BASELINE config 2 (tools/tune_stream.py, profiles/r01_tune_*.log): the forward
// (1 read : 1 write) is flat within ~2 % from 2 to 16 workgroups per CU, best at 16; the fused backward
// (2 reads : 1 write) is best with a SMALL persistent grid, 2 workgroups per CU -- the same optimum a
// no-arithmetic 2R:1W probe kernel shows, i.e. an HBM access-pattern effect, not a compute one.
constexpr int kDefaultFwdVariant = encode_variant(4, true, true, 16);
constexpr int kDefaultBwdVariant = encode_variant(4, true, true, 2);
// Per-channel (profiles/r01_pc_variant_sweep2.txt, BASELINE config 5, measured AFTER the finalize kernels
// were parallelised -- before that their serial chain of `splits` dependent loads made few workgroups look
// best): 16 workgroups/CU in both directions for 4/8-byte elements; the 16-bit backward (twice the arithmetic
// per byte, half the packets per lane) prefers the software-pipelined loop at unroll 1, 4/CU
// (profiles/r01_pc_pipeline_sweep.txt).
constexpr int kDefaultPcFwdVariant = encode_variant(4, true, true, 16);
constexpr int kDefaultPcBwdWideVariant = encode_variant(4, true, true, 16);   // fp32 / fp64 storage
constexpr int kDefaultPcBwdNarrowVariant = encode_variant(1, true, true, 4) | (1 << 10);  // bf16 / fp16 storage (bit 10: pipelined)
constexpr int kDefaultPcSegVariant = encode_variant(4, true, true, 16);
constexpr int kDefaultPcSegNarrowVariant = encode_variant(1, true, true, 16);   // bf16 / fp16 storage

// ---- launch-policy overrides: tools / tests ONLY -------------------------------------------------------------------
// The tools build of this library (tools/_tune/liblsq_hip_tools.so: `make tools`, -DLSQ_TOOLS) keeps a handful of
// process-wide integers that the A/B scripts under tools/ and the branch-pinning tests set through the lsq_hip_debug_*
// entry points of lsq_internal.h.  The production library (liblsq_hip.so) has none of this: every knob::get() below is the
// constant 0 ("no override"), the policy code that consults it folds away, no lsq_hip_debug_* symbol is exported, and the
// library keeps no mutable global state (include/lsq_hip.h).
namespace knob {
enum Id { kForceRing, kWwMinRows, kWwSplit64, kWwBig, kRingNt, kFinCh, kObserveWgPerCu, kWwMaxLog2, kSegMinDiv, kFwdDirect, kSegNoUpFront, kOwn, kOwnMinRun, kOwnFat, kCount };
#ifdef LSQ_TOOLS
inline std::atomic<int>& slot(Id id) {
    static std::atomic<int> v[kCount];
    return v[id];
}
inline int get(Id id) { return slot(id).load(std::memory_order_relaxed); }
inline void set(Id id, int v) { slot(id).store(v); }
#else
constexpr int get(Id) { return 0; }
#endif
#if defined(LSQ_TOOLS) && defined(LSQ_TIMELINE)
inline std::atomic<unsigned long long*>& timeline_buffer() {     // lsq_hip_debug_set_timeline (experiment build)
    static std::atomic<unsigned long long*> p{nullptr};
    return p;
}
#endif
// all geometry knobs as one key (the workspace memo of lsq_capi.hip)
inline int geometry_key() {
    unsigned k = 0;
    for (Id id : {kWwMinRows, kWwSplit64, kWwBig, kRingNt, kWwMaxLog2, kSegMinDiv, kOwn, kOwnMinRun, kOwnFat}) k = k * 41u + static_cast<unsigned>(get(id));
    return static_cast<int>(k & 0x7fffffffu);
}
}  // namespace knob

inline Variant decode_variant(int code, int dflt) {
    if (code == 0) code = dflt;
    Variant v;
    v.unroll = code & 0xff;
    v.nt_load = ((code >> 8) & 1) != 0;
    v.nt_store = ((code >> 9) & 1) != 0;
    v.blocks_per_cu = (code >> 16) & 0xff;
    v.chunked = ((code >> 10) & 1) != 0;
    v.dma = (code >> 12) & 3;
    if (v.dma == 0) v.dma = knob::get(knob::kForceRing);
    if (v.unroll != 1 && v.unroll != 2 && v.unroll != 4 && v.unroll != 8) v.unroll = 4;
    if (v.blocks_per_cu < 1) v.blocks_per_cu = 1;
    if (v.blocks_per_cu > kMaxBlocksPerCU) v.blocks_per_cu = kMaxBlocksPerCU;
    return v;
}

// LAUNCH(U, NTL, NTS) is a macro taking the compile-time unroll and non-temporal flags.
// Production builds compile ONE code path per kernel (the tuned default); -DLSQ_TUNING compiles the
// whole table for the kernels that pass FULL = true (fp32 per-tensor), for tools/tune_stream.py.
// DEFU = the compile-time unroll of the kernel's tuned default.
#ifdef LSQ_TUNING
#define LSQ_VARIANT_ROW(v, LAUNCH, NTL, NTS)                 \
    switch ((v).unroll) {                                    \
        case 1: LAUNCH(1, NTL, NTS); break;                  \
        case 2: LAUNCH(2, NTL, NTS); break;                  \
        case 8: LAUNCH(8, NTL, NTS); break;                  \
        default: LAUNCH(4, NTL, NTS); break;                 \
    }
#define LSQ_DISPATCH_VARIANT(FULL, DEFU, v, LAUNCH)                                   \
    do {                                                                              \
        if constexpr (FULL) {                                                         \
            if ((v).nt_load && (v).nt_store) { LSQ_VARIANT_ROW(v, LAUNCH, true, true) }        \
            else if ((v).nt_load) { LSQ_VARIANT_ROW(v, LAUNCH, true, false) }          \
            else if ((v).nt_store) { LSQ_VARIANT_ROW(v, LAUNCH, false, true) }         \
            else { LSQ_VARIANT_ROW(v, LAUNCH, false, false) }                          \
        } else {                                                                      \
            LAUNCH(DEFU, true, true);                                                 \
        }                                                                             \
    } while (0)
#else
#define LSQ_DISPATCH_VARIANT(FULL, DEFU, v, LAUNCH) \
    do {                                            \
        (void)(v);                                  \
        LAUNCH(DEFU, true, true);                   \
    } while (0)
#endif

// ---- device info (immutable after first use; benign race on initialisation) --------------------
struct DeviceInfo {
    int cu_count;
};
inline const DeviceInfo& device_info() {
    static DeviceInfo table[64];
    static std::atomic<int> ready[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!ready[dev].load(std::memory_order_acquire)) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        if (cus > kMaxCUs) cus = kMaxCUs;
        table[dev].cu_count = cus;
        ready[dev].store(1, std::memory_order_release);
    }
    return table[dev];
}

// Workgroups (kBlock threads = one wave64 per SIMD) of `kernel` that a CU holds at once, from the kernel's register
// allocation: 512 VGPRs per SIMD lane, allocated in blocks of 8, at most 8 waves per SIMD.  0 = unknown.  (The HIP
// occupancy API is one workgroup per CU too high for kernels with 81-112 SGPRs on this part, MI355X_MICROARCH.md, so the
// count is derived from the register number instead.)  Immutable per kernel: cached after the first query.
inline int resident_blocks_per_cu(const void* kernel, size_t lds_bytes = 0);
// (a launch consults this once: a lock-free open-addressing table keyed by the kernel's address, filled at a kernel's
// first launch -- no mutex, no runtime query on the steady-state path)
inline int registers_of(const void* kernel) {
    constexpr unsigned kSlots = 1024;      // far more than the kernels this library instantiates
    static std::atomic<const void*> keys[kSlots];
    static std::atomic<int> regs[kSlots];
    unsigned h = static_cast<unsigned>((reinterpret_cast<uintptr_t>(kernel) >> 4) * 2654435761u) % kSlots;
    for (unsigned probe = 0; probe < kSlots; ++probe, h = (h + 1) % kSlots) {
        const void* k = keys[h].load(std::memory_order_acquire);
        if (k == kernel) {
            const int r = regs[h].load(std::memory_order_acquire);
            if (r != 0) return r > 0 ? r : 0;
            break;                          // another thread is filling the slot: ask the runtime ourselves
        }
        if (k == nullptr) {
            hipFuncAttributes attr;
            int r = -1;
            if (hipFuncGetAttributes(&attr, kernel) == hipSuccess && attr.numRegs > 0) r = attr.numRegs;
            else (void)hipGetLastError();
            const void* expected = nullptr;
            if (keys[h].compare_exchange_strong(expected, kernel, std::memory_order_acq_rel)) {
                regs[h].store(r, std::memory_order_release);
                return r > 0 ? r : 0;
            }
            if (expected == kernel) return r > 0 ? r : 0;
            continue;                       // the slot went to another kernel meanwhile: keep probing
        }
    }
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, kernel) == hipSuccess && attr.numRegs > 0) return attr.numRegs;
    (void)hipGetLastError();
    return 0;
}
inline int resident_blocks_by_registers(const void* kernel) {
    const int r = registers_of(kernel);
    if (r <= 0) return 0;
    const int regs = (r + 7) & ~7;
    return std::max(1, std::min(8, 512 / regs));
}

// ... and the CU's 160 KiB of LDS allow (allocated in 1 KiB units here; the LDS-DMA rings take 32 KiB per workgroup)
inline int resident_blocks_per_cu(const void* kernel, size_t lds_bytes) {
    int n = resident_blocks_by_registers(kernel);
    if (n > 0 && lds_bytes > 0) n = std::max(1, std::min<int>(n, static_cast<int>((160 * 1024) / ((lds_bytes + 1023) & ~size_t(1023)))));
    return n;
}

// what a per-channel backward launch looks like: filled by a PLAN of the launch policy (lsq_hip_plan_backward_per_channel,
// production and tools build alike) and -- tools build only -- kept per thread for the last real launch
// (lsq_hip_debug_last_launch)
struct LaunchNote {
    int grid_x, grid_y, resident_per_cu, vgprs_hint;
    int kind;        // 1 = 256-lane windows, 2 = row-group windows, 3 = segment mode, 4 = owner windows
    int dma_depth;   // 0 = register loops, else the LDS-DMA ring depth
    int block;       // workgroup size
    int ring_nt;
};
#ifdef LSQ_TOOLS
inline LaunchNote& last_launch_note() {
    thread_local LaunchNote note = {0, 0, 0, 0, 0, 0, 0, 0};
    return note;
}
#endif

inline bool is_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
// a pointer to storage elements that is a multiple of the element size (every tensor view is; a raw C caller may not be)
template <typename IO>
inline bool is_elem_aligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & (sizeof(typename IO::elem) - 1u)) == 0; }

template <typename T>
inline Range<T> make_range(const lsq_params& p) {
    Range<T> r;
    r.qmin = static_cast<T>(p.quant_min);
    r.qmax = static_cast<T>(p.quant_max);
    r.tmin = static_cast<T>(p.type_min);
    r.tmax = static_cast<T>(p.type_max);
    return r;
}

// ---- gradient scaler (host) ---------------------------------------------------------------------
// lsq_cpu.cpp:103-104: static_cast<scalar_t>(grad_scaler / sqrt(x.numel()*qmax)).  The product is
// int64 * scalar_t evaluated in scalar_t, sqrt is the scalar_t overload, the quotient is formed in
// double and rounded once to scalar_t (chain pinned by tests/golden/small_cases.json "scaler_chain").
template <typename T>
inline T grad_scaler_per_tensor(int64_t numel, int32_t quant_max, bool use_grad_scaling, double grad_scaler) {
    if (!use_grad_scaling) return static_cast<T>(grad_scaler);
    const T prod = static_cast<T>(numel) * static_cast<T>(quant_max);
    const T root = std::sqrt(prod);
    return static_cast<T>(grad_scaler / static_cast<double>(root));
}
// lsq_cpu.cpp:250-251: grad_scaler / sqrt(x.numel()*qmax / x.size(axis))
template <typename T>
inline T grad_scaler_per_channel(int64_t numel, int32_t quant_max, int64_t channels, bool use_grad_scaling,
                                 double grad_scaler) {
    if (!use_grad_scaling) return static_cast<T>(grad_scaler);
    const T prod = static_cast<T>(numel) * static_cast<T>(quant_max);
    const T per_ch = prod / static_cast<T>(channels);
    const T root = std::sqrt(per_ch);
    return static_cast<T>(grad_scaler / static_cast<double>(root));
}

// ---- packed int8 levels of one packet -----------------------------------------------------------
template <int VEC>
struct LevelPack {
    int8_t b[VEC];
    __device__ __forceinline__ void store(int8_t* dst) const {
        if constexpr (VEC == 8) {
            uint64_t w;
            __builtin_memcpy(&w, b, 8);
            *reinterpret_cast<uint64_t*>(dst) = w;
        } else if constexpr (VEC == 4) {
            uint32_t w;
            __builtin_memcpy(&w, b, 4);
            *reinterpret_cast<uint32_t*>(dst) = w;
        } else if constexpr (VEC == 2) {
            uint16_t w;
            __builtin_memcpy(&w, b, 2);
            *reinterpret_cast<uint16_t*>(dst) = w;
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) dst[j] = b[j];
        }
    }
    __device__ __forceinline__ void load(const int8_t* src) {
        if constexpr (VEC == 8) {
            const uint64_t w = *reinterpret_cast<const uint64_t*>(src);
            __builtin_memcpy(b, &w, 8);
        } else if constexpr (VEC == 4) {
            const uint32_t w = *reinterpret_cast<const uint32_t*>(src);
            __builtin_memcpy(b, &w, 4);
        } else if constexpr (VEC == 2) {
            const uint16_t w = *reinterpret_cast<const uint16_t*>(src);
            __builtin_memcpy(b, &w, 2);
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) b[j] = src[j];
        }
    }
};

// ---- last-workgroup ticket (lsq_bwd_extras.ticket) ------------------------------------------------
// "The workgroup that finishes last folds the partial sums."  Hand-off between workgroups of one launch, following
// cdna_hip_programming.md section 6 guideline 16 (per-CU L1 and per-XCD L2 are not coherent with each other):
//   producer: the partial is stored WRITE-THROUGH at agent scope (global_store ... sc1) by ONE lane, which then drains
//             its stores (s_waitcnt vmcnt(0)) and bumps an agent-scope counter that WRAPS to zero at the `expected`-th
//             arrival (global_atomic_inc) -- so the ticket is all zero again when the launch ends;
//   consumer: the lane that saw the last arrival tells its workgroup through LDS + __syncthreads(), and the workgroup
//             reads the partials with agent-scope loads (global_load ... sc1), which bypass the non-coherent caches.
// No release / acquire FENCE: an agent-scope release writes the XCD's whole L2 back (buffer_wbl2 sc1) -- measured: +14 us
// on a 53 us backward full of dirty dx lines (gpurun_out/r02b) -- and nothing but the 16-byte partial needs publishing.
__device__ __forceinline__ void store_partial_agent(double2* p, double s, double b) {
    __hip_atomic_store(&p->x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&p->y, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool ticket_arrive_is_last(uint32_t* counter, uint32_t expected) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this lane's write-through stores have completed
    const uint32_t prev = atomicInc(counter, expected - 1u);   // old >= expected - 1 ? 0 : old + 1
    return prev == expected - 1u;
}
// another workgroup's partial, read past the non-coherent caches
__device__ __forceinline__ double2 load_partial_agent(const double2* p) {
    double2 v;
    v.x = __hip_atomic_load(&p->x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v.y = __hip_atomic_load(&p->y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
}

// ---- entry points implemented in lsq_per_tensor.hip / lsq_per_channel.hip ------------------------
size_t bwd_pt_workspace_bytes();
template <typename IO>
size_t bwd_pc_workspace_bytes(int64_t outer, int64_t channels, int64_t inner);

template <typename IO>
hipError_t forward_per_tensor(const void* x, void* y, int64_t n, const void* scale, const void* shift,
                              const lsq_params& p, const lsq_fwd_extras* ex, int variant, hipStream_t stream);
template <typename IO>
hipError_t backward_per_tensor(const void* grad, const void* x, void* dx, void* ds, void* db, double* wide,
                               int64_t n, const void* scale, const void* shift, const lsq_params& p,
                               void* workspace, uint32_t* ticket, int variant, hipStream_t stream);
template <typename IO>
hipError_t forward_per_channel(const void* x, void* y, int64_t outer, int64_t channels, int64_t inner,
                               const void* scale, const void* shift, const lsq_params& p,
                               const lsq_fwd_extras* ex, int variant, hipStream_t stream);
template <typename IO>
hipError_t backward_per_channel(const void* grad, const void* x, void* dx, void* ds, void* db, double* wide,
                                int64_t outer, int64_t channels, int64_t inner, const void* scale,
                                const void* shift, const lsq_params& p, void* workspace, size_t workspace_bytes,
                                uint32_t* ticket, int variant, hipStream_t stream, size_t* plan_need = nullptr,
                                LaunchNote* plan_note = nullptr);

// many per-channel quantizers in one launch (lsq_multi.hip)
template <typename IO>
bool multi_eligible(int64_t outer, int64_t channels, int64_t inner, bool aligned16);
template <typename IO>
hipError_t forward_per_channel_multi(const lsq_pc_item* items, int32_t count, const lsq_params& p, hipStream_t stream);
template <typename IO>
hipError_t backward_per_channel_multi(const lsq_pc_item* items, int32_t count, const lsq_params& p, hipStream_t stream);

// batch-sharded backward: scaler from the global element count + rounding, after the all-reduce (lsq_per_tensor.hip)
hipError_t relayout(int elem_bytes, const void* src, void* dst, int64_t A, int64_t B, int64_t C, hipStream_t stream);   // lsq_relayout.hip
template <typename T>
hipError_t sharded_finish(const double* packed, int64_t channels, bool per_channel, const lsq_params& p, void* ds, void* db,
                          hipStream_t stream);

template <typename IO>
hipError_t backward_from_mask(const void* grad, const void* mask, void* dx, int64_t n, hipStream_t stream);

// observer statistics (lsq_observe.hip)
size_t minmax_workspace_bytes(int io_vec, int elem_arith_bytes, int64_t outer, int64_t channels, int64_t inner);
template <typename IO>
hipError_t minmax_per_tensor(const void* x, int64_t n, void* out_min, void* out_max, void* workspace,
                             hipStream_t stream);
template <typename IO>
hipError_t minmax_per_channel(const void* x, int64_t outer, int64_t channels, int64_t inner, void* out_min,
                              void* out_max, void* workspace, size_t workspace_bytes, hipStream_t stream);

size_t meanstd_workspace_bytes(int io_vec, int64_t outer, int64_t channels, int64_t inner);
template <typename IO>
hipError_t meanstd_per_tensor(const void* x, int64_t n, void* out_mean, void* out_std, void* workspace, hipStream_t stream);
template <typename IO>
hipError_t meanstd_per_channel(const void* x, int64_t outer, int64_t channels, int64_t inner, void* out_mean,
                               void* out_std, void* workspace, size_t workspace_bytes, hipStream_t stream);

hipError_t observer_update(int64_t channels, const float* cur_min, const float* cur_max, float* min_state, float* max_state,
                           const lsq_observer_update& u, float* scale_out, float* shift_out, hipStream_t stream);

}  // namespace lsq
