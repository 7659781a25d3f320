// lsq_cpu_twin.cpp -- liblsq_cpu.so: the LSQ ops for tensors in HOST memory (include/lsq_cpu.h).
//
// The counterpart of the reference's CPU dispatch, /root/reference/torchlsq/csrc/ops/cpu/lsq_cpu.cpp:298-311, for the
// MI355X build: a user who prepares, calibrates or evaluates a model on the CPU finds the same operators there.
// Per-element arithmetic = lsq_kernel.h:6-14 (forward), :94-145 (backward), :151-256 (per-channel wrappers), each
// operation individually rounded (this file is compiled with -ffp-contract=off) and with the reference's own
// std::fmin / std::fmax / std::nearbyint, so y and dx are the reference's bits.  Structure is this build's own:
//   * the backward is ONE pass (the reference writes three N-sized buffers and sums two of them afterwards);
//   * d_scale / d_shift are accumulated in fp64 over fixed blocks of the index space and the block sums are added in
//     block order, so the result does not depend on the OpenMP thread count;
//   * per-channel constants {s, 1/s, zp} are computed once per channel, not per element.
// Not a fallback of the GPU path and not the test oracle (oracle/ is never linked or loaded by the product).
#include <algorithm>
#include <cmath>
#include <omp.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <vector>

#include "lsq_cpu.h"

namespace {

thread_local char g_error[256] = "";

// Threads of the parallel loops: what the caller's framework allows this thread (lsq_cpu_set_num_threads: the Python
// layer passes torch.get_num_threads() with every call), not whatever libgomp's own default is -- a DataLoader worker or a
// process pinned to one torch thread must not fan out over every core.  0 = never set: OpenMP's default.
thread_local int g_threads = 0;
inline int loop_threads() { return g_threads > 0 ? g_threads : omp_get_max_threads(); }

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    return code;
}

// ---- storage <-> arithmetic ----------------------------------------------------------------------------
struct F32 { using elem = float; using arith = float;
    static float load(const float* p, int64_t i) { return p[i]; }
    static void store(float* p, int64_t i, float v) { p[i] = v; } };
struct F64 { using elem = double; using arith = double;
    static double load(const double* p, int64_t i) { return p[i]; }
    static void store(double* p, int64_t i, double v) { p[i] = v; } };
struct BF16 { using elem = uint16_t; using arith = float;
    static float load(const uint16_t* p, int64_t i) { uint32_t b = static_cast<uint32_t>(p[i]) << 16; float f; std::memcpy(&f, &b, 4); return f; }
    static void store(uint16_t* p, int64_t i, float v) {     // round to nearest even; NaN stays NaN
        uint32_t b; std::memcpy(&b, &v, 4);
        if ((b & 0x7fffffffu) > 0x7f800000u) { p[i] = static_cast<uint16_t>((b >> 16) | 0x40u); return; }
        b += 0x7fffu + ((b >> 16) & 1u);
        p[i] = static_cast<uint16_t>(b >> 16);
    } };

template <typename T> struct Quant {      // one quantizer's constants (lsq_cpu.cpp:44-47 / lsq_kernel.h:157-158,12)
    T s, inv_s, zp, qmin, qmax;
};

template <typename T>
Quant<T> make_quant(T s_sanitized, T shift, const lsq_params& p) {
    Quant<T> q;
    q.s = s_sanitized;
    q.inv_s = static_cast<T>(1) / s_sanitized;
    q.qmin = static_cast<T>(p.quant_min);
    q.qmax = static_cast<T>(p.quant_max);
    q.zp = std::nearbyint(std::fmin(static_cast<T>(p.type_max), std::fmax(static_cast<T>(p.type_min), -shift * q.inv_s)));
    return q;
}
template <typename T> T eps_of() { return std::numeric_limits<T>::epsilon(); }
template <typename T> T sanitize_pt(T s0) { return std::max(std::abs(s0), eps_of<T>()); }          // lsq_cpu.cpp:45-46
template <typename T> T sanitize_pc(T sc) { return std::fmax(eps_of<T>(), std::abs(sc)); }          // lsq_kernel.h:157

template <typename T>
inline T fwd_elem(T x, const Quant<T>& q, bool init) {                                             // lsq_kernel.h:6-14
    if (init) return x;
    const T lvl = std::nearbyint(std::fmin(q.qmax, std::fmax(q.qmin, x * q.inv_s + q.zp)));
    return (lvl - q.zp) * q.s;
}

template <typename T>
inline T bwd_elem(T g, T x, const Quant<T>& q, const lsq_params& p, T gs, double& acc_s, double& acc_b) {   // :94-145
    const T xq = std::fmax(std::fmin(x * q.inv_s + q.zp, q.qmax), q.qmin);
    const bool inside = (q.qmin < xq) && (xq < q.qmax);
    const T mask = inside ? static_cast<T>(1) : static_cast<T>(0);
    const T dX = p.init_mode ? g : g * mask;
    if (p.eval_mode) return dX;
    const T xfq = (std::nearbyint(xq) - q.zp) * q.s;
    const T G = p.init_mode ? static_cast<T>(2) * (xfq - x) : g;
    const T dB = p.sym ? static_cast<T>(0) : (static_cast<T>(1) - mask) * G;
    const T dS = inside ? G * (xfq - x) * q.inv_s : (xq <= q.qmin ? G * (q.qmin - q.zp) : G * (q.qmax - q.zp));
    acc_s += static_cast<double>(dS * gs);
    acc_b += static_cast<double>(dB * gs);
    return dX;
}

// gradient scaler: the reference's precision chain (lsq_cpu.cpp:103-104, :250-251)
template <typename T>
T grad_scaler(const lsq_params& p, int64_t numel, int64_t channels) {
    if (!p.use_grad_scaling) return static_cast<T>(p.grad_scaler);
    const int64_t n = p.numel_for_scaler > 0 ? p.numel_for_scaler : numel;
    T prod = static_cast<T>(n) * static_cast<T>(p.quant_max);
    if (channels > 0) prod = prod / static_cast<T>(channels);
    return static_cast<T>(p.grad_scaler / static_cast<double>(std::sqrt(prod)));
}

constexpr int64_t kBlock = 1 << 14;   // elements per reduction block (fixed: the summation order never depends on threads)

template <typename IO>
int forward_pt(const void* xv, void* yv, int64_t n, const void* scale, const void* shift, const lsq_params& p) {
    using T = typename IO::arith;
    const auto* x = static_cast<const typename IO::elem*>(xv);
    auto* y = static_cast<typename IO::elem*>(yv);
    const Quant<T> q = make_quant<T>(sanitize_pt<T>(static_cast<const T*>(scale)[0]), static_cast<const T*>(shift)[0], p);
    const bool init = p.init_mode != 0;
#pragma omp parallel for schedule(static) num_threads(loop_threads())
    for (int64_t b0 = 0; b0 < n; b0 += kBlock) {
        const int64_t b1 = std::min(n, b0 + kBlock);
        for (int64_t i = b0; i < b1; ++i) IO::store(y, i, fwd_elem<T>(IO::load(x, i), q, init));
    }
    return LSQ_OK;
}

template <typename IO>
int backward_pt(const void* gv, const void* xv, void* dxv, void* dsv, void* dbv, double* wide, int64_t n,
                const void* scale, const void* shift, const lsq_params& p) {
    using T = typename IO::arith;
    const auto* g = static_cast<const typename IO::elem*>(gv);
    const auto* x = static_cast<const typename IO::elem*>(xv);
    auto* dx = static_cast<typename IO::elem*>(dxv);
    const Quant<T> q = make_quant<T>(sanitize_pt<T>(static_cast<const T*>(scale)[0]), static_cast<const T*>(shift)[0], p);
    const T gs = grad_scaler<T>(p, n, 0);
    const int64_t n_blocks = (n + kBlock - 1) / kBlock;
    std::vector<double> part(static_cast<size_t>(2 * n_blocks), 0.0);
#pragma omp parallel for schedule(static) num_threads(loop_threads())
    for (int64_t b = 0; b < n_blocks; ++b) {
        const int64_t b0 = b * kBlock, b1 = std::min(n, b0 + kBlock);
        double as = 0.0, ab = 0.0;
        for (int64_t i = b0; i < b1; ++i) IO::store(dx, i, bwd_elem<T>(IO::load(g, i), IO::load(x, i), q, p, gs, as, ab));
        part[2 * b] = as;
        part[2 * b + 1] = ab;
    }
    double ts = 0.0, tb = 0.0;
    for (int64_t b = 0; b < n_blocks; ++b) { ts += part[2 * b]; tb += part[2 * b + 1]; }
    if (p.eval_mode) { ts = 0.0; tb = 0.0; }
    static_cast<T*>(dsv)[0] = static_cast<T>(ts);
    static_cast<T*>(dbv)[0] = static_cast<T>(tb);
    if (wide) { wide[0] = ts; wide[1] = tb; }
    return LSQ_OK;
}

template <typename T>
std::vector<Quant<T>> channel_table(const void* scale, const void* shift, int64_t C, const lsq_params& p) {
    std::vector<Quant<T>> t(static_cast<size_t>(C));
    for (int64_t c = 0; c < C; ++c)
        t[c] = make_quant<T>(sanitize_pc<T>(static_cast<const T*>(scale)[c]), static_cast<const T*>(shift)[c], p);
    return t;
}

template <typename IO>
int forward_pc(const void* xv, void* yv, int64_t outer, int64_t C, int64_t inner, const void* scale, const void* shift,
               const lsq_params& p) {
    using T = typename IO::arith;
    const auto* x = static_cast<const typename IO::elem*>(xv);
    auto* y = static_cast<typename IO::elem*>(yv);
    const std::vector<Quant<T>> tab = channel_table<T>(scale, shift, C, p);
    const bool init = p.init_mode != 0;
    const int64_t rows = outer * C;
#pragma omp parallel for schedule(static) num_threads(loop_threads())
    for (int64_t r = 0; r < rows; ++r) {
        const Quant<T>& q = tab[r % C];
        const int64_t base = r * inner;
        for (int64_t i = 0; i < inner; ++i) IO::store(y, base + i, fwd_elem<T>(IO::load(x, base + i), q, init));
    }
    return LSQ_OK;
}

// The reduction runs over `chunks` fixed slabs of the outer index (or, for few outer indices -- weights -- over the
// channels themselves); every (slab, channel) sum is complete before the slabs are added in slab order.
template <typename IO>
int backward_pc(const void* gv, const void* xv, void* dxv, void* dsv, void* dbv, double* wide, int64_t outer, int64_t C,
                int64_t inner, const void* scale, const void* shift, const lsq_params& p) {
    using T = typename IO::arith;
    const auto* g = static_cast<const typename IO::elem*>(gv);
    const auto* x = static_cast<const typename IO::elem*>(xv);
    auto* dx = static_cast<typename IO::elem*>(dxv);
    const std::vector<Quant<T>> tab = channel_table<T>(scale, shift, C, p);
    const T gs = grad_scaler<T>(p, outer * C * inner, C);
    const int64_t chunks = std::max<int64_t>(1, std::min<int64_t>(outer, 256));
    std::vector<double> part(static_cast<size_t>(2 * chunks * C), 0.0);
    const int64_t units = chunks * C;      // (slab, channel) pairs: independent, any thread may take any of them
#pragma omp parallel for schedule(static) num_threads(loop_threads())
    for (int64_t u = 0; u < units; ++u) {
        const int64_t k = u / C, c = u % C;
        const int64_t o0 = k * outer / chunks, o1 = (k + 1) * outer / chunks;
        const Quant<T>& q = tab[c];
        double as = 0.0, ab = 0.0;
        for (int64_t o = o0; o < o1; ++o) {
            const int64_t base = (o * C + c) * inner;
            for (int64_t i = 0; i < inner; ++i)
                IO::store(dx, base + i, bwd_elem<T>(IO::load(g, base + i), IO::load(x, base + i), q, p, gs, as, ab));
        }
        part[2 * u] = as;
        part[2 * u + 1] = ab;
    }
    T* ds = static_cast<T*>(dsv);
    T* db = static_cast<T*>(dbv);
    for (int64_t c = 0; c < C; ++c) {
        double ts = 0.0, tb = 0.0;
        for (int64_t k = 0; k < chunks; ++k) { ts += part[2 * (k * C + c)]; tb += part[2 * (k * C + c) + 1]; }
        if (p.eval_mode) { ts = 0.0; tb = 0.0; }
        ds[c] = static_cast<T>(ts);
        db[c] = static_cast<T>(tb);
        if (wide) { wide[c] = ts; wide[C + c] = tb; }
    }
    return LSQ_OK;
}

int check(int dtype, const lsq_params* p) {
    if (dtype < LSQ_F32 || dtype > LSQ_BF16) return fail(LSQ_EINVAL, "dtype code %d is not served on the CPU (float32, float64, bfloat16)", dtype);
    if (!p) return fail(LSQ_EINVAL, "lsq_params pointer is NULL");
    if (p->quant_min > p->quant_max) return fail(LSQ_EINVAL, "quant_min %d > quant_max %d", p->quant_min, p->quant_max);
    if (p->type_min > p->type_max) return fail(LSQ_EINVAL, "type_min %d > type_max %d", p->type_min, p->type_max);
    return LSQ_OK;
}

}  // namespace

// lsq_hip_sharded_finish for host memory: gradient scaler (lsq_cpu.cpp:103-104 / :250-251, same chain as this file's
// backward) from the all-reduced element count, applied once to the fp64 sums, one rounding to the parameter type.
template <typename T>
static int sharded_finish(const double* packed, int64_t C, bool per_channel, const lsq_params& p, void* ds_, void* db_) {
    T* ds = static_cast<T*>(ds_);
    T* db = static_cast<T*>(db_);
    const double count = packed[2 * C];
    T gs = static_cast<T>(p.grad_scaler);
    if (p.use_grad_scaling) {
        T prod = static_cast<T>(count) * static_cast<T>(p.quant_max);
        if (per_channel) prod = prod / static_cast<T>(C);
        gs = static_cast<T>(p.grad_scaler / static_cast<double>(std::sqrt(prod)));
    }
    if (!(count > 0.0)) gs = static_cast<T>(0);
    for (int64_t c = 0; c < C; ++c) {
        ds[c] = static_cast<T>(packed[c] * static_cast<double>(gs));
        db[c] = static_cast<T>(packed[C + c] * static_cast<double>(gs));
    }
    return LSQ_OK;
}

#define LSQ_CPU_DISPATCH(dtype, CALL)                 \
    switch (dtype) {                                  \
        case LSQ_F32: { using IO = F32; return CALL; }   \
        case LSQ_F64: { using IO = F64; return CALL; }   \
        default: { using IO = BF16; return CALL; }       \
    }

extern "C" {

int lsq_cpu_abi_version(void) { return LSQ_HIP_ABI_VERSION; }
void lsq_cpu_set_num_threads(int n) { g_threads = n > 0 ? n : 0; }
const char* lsq_cpu_last_error(void) { return g_error; }

int lsq_cpu_forward_per_tensor(int dtype, const void* x, void* y, int64_t n, const void* scale, const void* shift,
                               const lsq_params* p) {
    if (int rc = check(dtype, p)) return rc;
    if (n < 0) return fail(LSQ_EINVAL, "negative element count");
    if (n == 0) return LSQ_OK;
    if (!x || !y || !scale || !shift) return fail(LSQ_EINVAL, "forward_per_tensor: NULL buffer");
    LSQ_CPU_DISPATCH(dtype, forward_pt<IO>(x, y, n, scale, shift, *p));
}

int lsq_cpu_backward_per_tensor(int dtype, const void* grad, const void* x, void* dx, void* ds, void* db,
                                double* dsdb_wide, int64_t n, const void* scale, const void* shift, const lsq_params* p) {
    if (int rc = check(dtype, p)) return rc;
    if (n <= 0) return fail(LSQ_EINVAL, "backward_per_tensor: element count must be positive (the caller handles the "
                                        "empty case, reference lsq_cpu.cpp:76-78)");
    if (!grad || !x || !dx || !ds || !db || !scale || !shift) return fail(LSQ_EINVAL, "backward_per_tensor: NULL buffer");
    LSQ_CPU_DISPATCH(dtype, backward_pt<IO>(grad, x, dx, ds, db, dsdb_wide, n, scale, shift, *p));
}

int lsq_cpu_forward_per_channel(int dtype, const void* x, void* y, int64_t outer, int64_t channels, int64_t inner,
                                const void* scale, const void* shift, const lsq_params* p) {
    if (int rc = check(dtype, p)) return rc;
    if (outer < 0 || channels <= 0 || inner < 0) return fail(LSQ_EINVAL, "bad [outer, C, inner]");
    if (outer == 0 || inner == 0) return LSQ_OK;
    if (!x || !y || !scale || !shift) return fail(LSQ_EINVAL, "forward_per_channel: NULL buffer");
    LSQ_CPU_DISPATCH(dtype, forward_pc<IO>(x, y, outer, channels, inner, scale, shift, *p));
}

int lsq_cpu_backward_per_channel(int dtype, const void* grad, const void* x, void* dx, void* ds, void* db,
                                 double* dsdb_wide, int64_t outer, int64_t channels, int64_t inner, const void* scale,
                                 const void* shift, const lsq_params* p) {
    if (int rc = check(dtype, p)) return rc;
    if (outer <= 0 || channels <= 0 || inner <= 0) return fail(LSQ_EINVAL, "backward_per_channel: empty or bad [outer, C, inner]");
    if (!grad || !x || !dx || !ds || !db || !scale || !shift) return fail(LSQ_EINVAL, "backward_per_channel: NULL buffer");
    LSQ_CPU_DISPATCH(dtype, backward_pc<IO>(grad, x, dx, ds, db, dsdb_wide, outer, channels, inner, scale, shift, *p));
}

int lsq_cpu_sharded_finish(int dtype, const double* packed, int64_t channels, int32_t per_channel, const lsq_params* p,
                           void* ds, void* db) {
    if (int rc = check(dtype, p)) return rc;
    if (channels <= 0 || (!per_channel && channels != 1)) return fail(LSQ_EINVAL, "sharded_finish: bad channel count");
    if (!packed || !ds || !db) return fail(LSQ_EINVAL, "sharded_finish: NULL buffer");
    return dtype == LSQ_F64 ? sharded_finish<double>(packed, channels, per_channel != 0, *p, ds, db)
                            : sharded_finish<float>(packed, channels, per_channel != 0, *p, ds, db);
}

}  // extern "C"
