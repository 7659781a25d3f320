// lsq_math.hpp -- per-element LSQ arithmetic for gfx950, shared by the per-tensor and the
// per-channel kernels.
//
// The arithmetic contract is the reference's scalar header
//   /root/reference/torchlsq/csrc/ops/kernels/lsq_kernel.h   (fwd :6-14, fused bwd :94-123, eval :126-145)
// as built for its CPU backend (global_scope.h:8-20: FASTROUND = std::nearbyint, i.e. round half
// to even; FMIN/FMAX = std::fmin/std::fmax, NaN-suppressing), because the CPU path is the parity
// oracle.  To be bit-identical with that build (x86-64 without FMA):
//   * every multiply and add is an individually rounded IEEE operation -- this directory is
//     compiled with -ffp-contract=off, so `x * inv_s + zp` is v_mul_f32 + v_add_f32, never v_fma;
//   * 1/s is a correctly rounded division (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt);
//   * rounding is v_rndne_f32 / v_rndne_f64 (__builtin_rint*: current mode = nearest-even);
//   * fp32 denormals are preserved (hipcc default for gfx9).
// Written for the CDNA4 execution model: everything here is branch-free straight-line VALU code
// (selects, no divergent control flow), so a wave64 runs it at full lane occupancy.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lsq {

// ---- arithmetic-type helpers ------------------------------------------------------------------
__device__ __forceinline__ float rne(float v) { return __builtin_rintf(v); }
__device__ __forceinline__ double rne(double v) { return __builtin_rint(v); }
__device__ __forceinline__ float fmin_(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ double fmin_(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double fmax_(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ float fabs_(float a) { return __builtin_fabsf(a); }
__device__ __forceinline__ double fabs_(double a) { return __builtin_fabs(a); }

template <typename T> struct eps_of;
template <> struct eps_of<float> { static constexpr float value = 1.1920928955078125e-07f; };   // FLT_EPSILON
template <> struct eps_of<double> { static constexpr double value = 2.220446049250313e-16; };    // DBL_EPSILON

// Quantisation range in the arithmetic type (lsq_cpu.cpp:40-43: static_cast<scalar_t>(quant_min) ...).
template <typename T>
struct Range {
    T qmin, qmax, tmin, tmax;
};

// Per-quantiser constants derived once from (scale, shift): what every element of the reference
// recomputes (lsq_kernel.h:12 and :157-158).
template <typename T>
struct QParams {
    T s;      // sanitised scale
    T inv_s;  // 1 / s
    T zp;     // rne(fmin(tmax, fmax(tmin, -b * inv_s)))
};

// per-tensor sanitising: s = std::max(|scale[0]|, eps)  (lsq_cpu.cpp:45-46; std::max keeps a NaN scale)
template <typename T>
__device__ __forceinline__ T sanitize_scale_per_tensor(T scale0) {
    const T a = fabs_(scale0);
    return (a < eps_of<T>::value) ? eps_of<T>::value : a;
}

// per-channel sanitising: _s = FMAX(eps, ABS(s))  (lsq_kernel.h:157; fmax drops a NaN scale)
template <typename T>
__device__ __forceinline__ T sanitize_scale_per_channel(T scale_c) {
    return fmax_(eps_of<T>::value, fabs_(scale_c));
}

template <typename T>
__device__ __forceinline__ QParams<T> make_qparams(T s_sanitized, T shift, const Range<T>& r) {
    QParams<T> q;
    q.s = s_sanitized;
    q.inv_s = static_cast<T>(1) / s_sanitized;
    q.zp = rne(fmin_(r.tmax, fmax_(r.tmin, -shift * q.inv_s)));  // lsq_kernel.h:12
    return q;
}

// ---- forward (lsq_kernel.h:6-14) ---------------------------------------------------------------
// the integer level, still in floating point: FASTROUND(FMIN(qmax, FMAX(qmin, x*inv_s + zp)))
// The two clamps of the reference differ in their order, hence in where a NaN ends up:
//   forward  (lsq_kernel.h:13)   fmin(qmax, fmax(qmin, t))   NaN -> qmin
//   backward (lsq_kernel.h:108)  fmax(fmin(t, qmax), qmin)   NaN -> qmax
// fp32: ONE v_med3_f32 each instead of v_max + v_min.  v_med3_f32 returns min3 of its operands when one of them is a NaN
// (t is an arithmetic result, hence quiet), i.e. qmin -- the forward's answer; the backward's clamp is the same
// instruction on the negated values, -med3(-t, -qmax, -qmin): a NaN goes to min3 = -qmax, and the negations fold into
// source modifiers of the instructions around it.  Signed zeros order as in v_min / v_max (-0 < +0), so a clamp to a zero
// bound gives the same zero as the two-instruction form.  Held to the reference bit for bit, NaN / +-inf / +-0 included,
// by the golden cases of tests/test_parity_gpu.py.  fp64 has no med3 and keeps the two-instruction form.
__device__ __forceinline__ float clamp_fwd(float t, float qmin, float qmax) { return __builtin_amdgcn_fmed3f(t, qmin, qmax); }
__device__ __forceinline__ double clamp_fwd(double t, double qmin, double qmax) { return fmin_(qmax, fmax_(qmin, t)); }
// -xq = med3(-t, -qmax, -qmin).  The (empty) asm keeps the result opaque: the compiler otherwise pushes a later negation
// through the intrinsic -- "-med3(-a, -b, -c) == med3(a, b, c)", true for numbers, not for a NaN, which would end up on
// qmin like in the forward (found by the NaN golden case).
__device__ __forceinline__ float neg_clamp_bwd(float t, float nqmin, float nqmax) {
    float m = __builtin_amdgcn_fmed3f(-t, nqmax, nqmin);
    asm("" : "+v"(m));
    return m;
}
__device__ __forceinline__ float clamp_bwd(float t, float qmin, float qmax) { return -neg_clamp_bwd(t, -qmin, -qmax); }
__device__ __forceinline__ double clamp_bwd(double t, double qmin, double qmax) { return fmax_(fmin_(t, qmax), qmin); }

template <typename T>
__device__ __forceinline__ T clamped(T x, const QParams<T>& q, const Range<T>& r) {
    return clamp_fwd(x * q.inv_s + q.zp, r.qmin, r.qmax);
}
template <typename T>
__device__ __forceinline__ T level(T x, const QParams<T>& q, const Range<T>& r) {
    return rne(clamped<T>(x, q, r));
}
// One auxiliary byte per element next to y (lsq_fwd_extras): either the integer level (minus a bias) or the
// "strictly inside the range" flag the eval-mode backward needs (lsq_kernel.h:109 / :139; the two clamp
// orders of forward and backward give the same flag: both put a NaN on a border).
template <typename T>
__device__ __forceinline__ int8_t aux_byte(T c, const Range<T>& r, T bias, int aux_kind) {
    return aux_kind ? static_cast<int8_t>((r.qmin < c) && (c < r.qmax))
                    : static_cast<int8_t>(static_cast<int>(rne(c) - bias));
}

template <typename T>
__device__ __forceinline__ T dequant(T lvl, const QParams<T>& q) {
    return (lvl - q.zp) * q.s;
}

// ---- backward (lsq_kernel.h:94-123) ------------------------------------------------------------
// Returns dX; ds_term / db_term are the per-element contributions (already multiplied by the
// gradient scaler, :122) that the reference writes to ds_buffer / db_buffer (lsq_cpu.cpp:131-133).
// RAW: leave the gradient scaler out (ds_term = dS, db_term = dB): the caller multiplies its SUMS by it once.  Only the
// 16-bit-storage kernels do that (their parity is defined by this build); a sum of individually scaled and
// rounded terms and the scaled sum differ by at most 2^-24 per term, far inside the 1e-6 bar.
template <typename T, bool SYM, bool INIT, bool RAW = false>
__device__ __forceinline__ T backward_elem(T grad, T x, const QParams<T>& q, const Range<T>& r,
                                           T grad_scaler, T& ds_term, T& db_term) {
    const T xq = clamp_bwd(x * q.inv_s + q.zp, r.qmin, r.qmax);    // :108, clamp = min then max, unrounded
    const bool inside = (r.qmin < xq) & (xq < r.qmax);             // :109
    const T mask = inside ? static_cast<T>(1) : static_cast<T>(0);
    const T dX = INIT ? grad : grad * mask;                        // :112 (a real multiply: inf*0 = NaN)
    const T d = rne(xq) - q.zp;
    const T xfq = d * q.s;                                         // :115
    const T err = xfq - x;
    const T g_ = INIT ? static_cast<T>(2) * err : grad;            // :116
    // :120 border = xq <= qmin ? g_ * (qmin - zp) : g_ * (qmax - zp).  An element that is not inside sits exactly ON a
    // border (the clamp put it there, a NaN on qmax), the borders are integers, so rne(xq) == xq == that border and `d`
    // already IS (qmin - zp) resp. (qmax - zp), the same subtraction of the same two values: no comparison, no second
    // constant.
    const T border = g_ * d;
    const T inner = g_ * err * q.inv_s;                            // (both arms are values: the ternary below is a select)
    const T dS = inside ? inner : border;                          // :121
    ds_term = RAW ? dS : dS * grad_scaler;                         // :122
    if (SYM) {
        db_term = static_cast<T>(0);                               // :118 (0 * scaler)
    } else {
        const T dB = (static_cast<T>(1) - mask) * g_;              // :118 static_cast<scalar_t>(!mask) * _grad
        db_term = RAW ? dB : dB * grad_scaler;
    }
    return dX;
}

// eval mode (lsq_kernel.h:126-145): dx only
template <typename T, bool INIT>
__device__ __forceinline__ T backward_elem_eval(T grad, T x, const QParams<T>& q, const Range<T>& r) {
    if (INIT) return grad;
    const T xq = clamp_bwd(x * q.inv_s + q.zp, r.qmin, r.qmax);
    const bool inside = (r.qmin < xq) && (xq < r.qmax);
    return grad * (inside ? static_cast<T>(1) : static_cast<T>(0));
}

// ---- the same arithmetic on PAIRS of fp32 elements ------------------------------------------------------------------
// gfx950 has packed fp32 multiply / add (v_pk_mul_f32, v_pk_add_f32: two lanes' worth of one IEEE operation per
// instruction, each half individually rounded -- no contraction, same bits as the scalar forms) but no packed min / max /
// round / compare / select.  The streaming kernels whose time is instruction issue rather than HBM (16-bit storage: half
// the bytes per element) therefore compute on explicit 2-vectors: every multiply and add of lsq_kernel.h below is ONE
// packed instruction per two elements, the rest one instruction per element.  Element for element this is
// backward_elem / level / dequant above, operation by operation (the negated clamp only moves signs into source
// modifiers: rne(-v) == -rne(v), (-a) - b == -(a + b) bit for bit under round-to-nearest-even).
using f2 = __attribute__((ext_vector_type(2))) float;

struct QPair {   // the constants of two adjacent elements (the same channel or two channels)
    f2 s, inv_s, zp;
};
__device__ __forceinline__ QPair make_qpair(const QParams<float>& a, const QParams<float>& b) {
    QPair q;
    q.s = f2{a.s, b.s}; q.inv_s = f2{a.inv_s, b.inv_s}; q.zp = f2{a.zp, b.zp};
    return q;
}
__device__ __forceinline__ f2 select2(bool c0, bool c1, f2 a, f2 b) { return f2{c0 ? a.x : b.x, c1 ? a.y : b.y}; }

// forward (lsq_kernel.h:6-14): the clamped, unrounded level position c (what aux_byte wants) and y
__device__ __forceinline__ f2 forward_pair(f2 x, const QPair& q, const Range<float>& r, f2& c) {
    const f2 t = x * q.inv_s + q.zp;
    c = f2{clamp_fwd(t.x, r.qmin, r.qmax), clamp_fwd(t.y, r.qmin, r.qmax)};
    const f2 lvl = f2{rne(c.x), rne(c.y)};
    return (lvl - q.zp) * q.s;
}

// backward (lsq_kernel.h:94-123): returns dX; ds / db are the per-element terms WITHOUT the gradient scaler (the caller
// multiplies terms or sums, see backward_elem's RAW)
template <bool SYM, bool INIT>
__device__ __forceinline__ f2 backward_pair(f2 grad, f2 x, const QPair& q, const Range<float>& r, f2& ds, f2& db) {
    const float nqmin = -r.qmin, nqmax = -r.qmax;
    const f2 t = x * q.inv_s + q.zp;
    const f2 nxq = f2{neg_clamp_bwd(t.x, nqmin, nqmax), neg_clamp_bwd(t.y, nqmin, nqmax)};                        // -xq, :108
    const bool in0 = (nxq.x < nqmin) & (nqmax < nxq.x), in1 = (nxq.y < nqmin) & (nqmax < nxq.y);                  // :109
    const f2 mask = f2{in0 ? 1.0f : 0.0f, in1 ? 1.0f : 0.0f};
    const f2 dX = INIT ? grad : grad * mask;                      // :112
    const f2 d = -f2{rne(nxq.x), rne(nxq.y)} - q.zp;              // rne(xq) - zp
    const f2 err = d * q.s - x;                                   // :115 xfq - x
    const f2 g_ = INIT ? 2.0f * err : grad;                       // :116
    const f2 border = g_ * d;                                     // :120 (see backward_elem: d IS the border constant)
    const f2 inner = g_ * err * q.inv_s;
    ds = select2(in0, in1, inner, border);                        // :121
    db = SYM ? f2{0.0f, 0.0f} : (1.0f - mask) * g_;               // :118
    return dX;
}

template <bool INIT>
__device__ __forceinline__ f2 backward_pair_eval(f2 grad, f2 x, const QPair& q, const Range<float>& r) {
    if (INIT) return grad;
    const float nqmin = -r.qmin, nqmax = -r.qmax;
    const f2 t = x * q.inv_s + q.zp;
    const f2 nxq = f2{neg_clamp_bwd(t.x, nqmin, nqmax), neg_clamp_bwd(t.y, nqmin, nqmax)};
    const bool in0 = (nxq.x < nqmin) & (nqmax < nxq.x), in1 = (nxq.y < nqmin) & (nqmax < nxq.y);
    return grad * f2{in0 ? 1.0f : 0.0f, in1 ? 1.0f : 0.0f};
}

// ---- storage types: 16-byte vectors in HBM, arithmetic type in registers ------------------------
// IO traits: `vec` is the 16-byte register image, VEC elements per vector, arith = math type.
struct io_f32 {
    using elem = float;
    using arith = float;
    static constexpr int VEC = 4;
    __device__ static __forceinline__ float load1(const void* p, int64_t i) { return static_cast<const float*>(p)[i]; }
    __device__ static __forceinline__ void store1(void* p, int64_t i, float v) { static_cast<float*>(p)[i] = v; }
    __device__ static __forceinline__ float to_elem(float v) { return v; }
    __device__ static __forceinline__ float passthrough(float v) { return v; }
};
struct io_f64 {
    using elem = double;
    using arith = double;
    static constexpr int VEC = 2;
    __device__ static __forceinline__ double load1(const void* p, int64_t i) { return static_cast<const double*>(p)[i]; }
    __device__ static __forceinline__ void store1(void* p, int64_t i, double v) { static_cast<double*>(p)[i] = v; }
    __device__ static __forceinline__ double to_elem(double v) { return v; }
    __device__ static __forceinline__ double passthrough(double v) { return v; }
};
struct io_bf16 {
    using elem = __bf16;
    using arith = float;
    static constexpr int VEC = 8;
    __device__ static __forceinline__ float load1(const void* p, int64_t i) {
        return static_cast<float>(static_cast<const __bf16*>(p)[i]);
    }
    __device__ static __forceinline__ __bf16 to_elem(float v) { return static_cast<__bf16>(v); }  // RNE (v_cvt_pk_bf16_f32)
    // v == float(some bf16): its storage bits are the high half of v's -- NaN payload and sign included, where the rounding
    // conversion writes the canonical NaN
    __device__ static __forceinline__ __bf16 passthrough(float v) {
        const unsigned short h = static_cast<unsigned short>(__float_as_uint(v) >> 16);
        __bf16 out;
        __builtin_memcpy(&out, &h, 2);
        return out;
    }
    __device__ static __forceinline__ void store1(void* p, int64_t i, float v) { static_cast<__bf16*>(p)[i] = to_elem(v); }
};
struct io_f16 {
    using elem = _Float16;
    using arith = float;
    static constexpr int VEC = 8;
    __device__ static __forceinline__ float load1(const void* p, int64_t i) {
        return static_cast<float>(static_cast<const _Float16*>(p)[i]);
    }
    // The fp32 result and its rounding to fp16 are kept apart by an (empty) asm: hipcc otherwise selects
    // "fp16 -> fp32, multiply by the 0/1 mask, -> fp16" as v_fma_mixlo_f16 a, b, +0, and the +0 addend turns the
    // -0 of (negative grad) * 0 into +0 (found by tests/test_fuzz_gpu.py; fp32 and bf16 storage keep the sign).
    __device__ static __forceinline__ _Float16 to_elem(float v) {
        asm("" : "+v"(v));
        return static_cast<_Float16>(v);
    }
    __device__ static __forceinline__ void store1(void* p, int64_t i, float v) { static_cast<_Float16*>(p)[i] = to_elem(v); }
    __device__ static __forceinline__ _Float16 passthrough(float v) { return to_elem(v); }   // exact for v == float(some fp16); a quiet NaN keeps sign and payload
};

// init_mode hands values through (lsq_kernel.h:9 y = x; :112 / :140 dX = grad): EXACT = the value in hand IS a storage value
// converted to the arithmetic type, and goes back as its bits -- not through the rounding conversion.
template <typename IO, bool EXACT>
__device__ __forceinline__ typename IO::elem out_elem(typename IO::arith v) {
    if constexpr (EXACT) return IO::passthrough(v);
    else return IO::to_elem(v);
}
template <typename IO, bool EXACT>
__device__ __forceinline__ void store_out(void* p, int64_t i, typename IO::arith v) {
    static_cast<typename IO::elem*>(p)[i] = out_elem<IO, EXACT>(v);
}

// A 16-byte packet of IO::VEC storage elements, moved with one global_load/store_dwordx4.
// The packet pointer is only promised ELEMENT alignment (PacketWord: a 16-byte vector type declared with the alignment of
// one storage element): gfx9 under HSA runs with unaligned access enabled, the compiler knows it (it emits the same single
// global_load_dwordx4 / global_store_dwordx4 either way: tests/test_device_code.py) and the hardware splits an access that
// straddles a cache line itself.  That is what lets a sliced view -- x[1:], a storage offset that is not a multiple of 16
// bytes, buffers whose offsets differ from each other -- run the packet kernels instead of one element per lane
// (reference: TensorIterator walks any view in place, lsq_cpu.cpp:31-36,80-90).
template <int ELEM_BYTES> struct PacketWordOf;
template <> struct PacketWordOf<2> { typedef __attribute__((ext_vector_type(4))) unsigned int type __attribute__((aligned(2))); };
template <> struct PacketWordOf<4> { typedef __attribute__((ext_vector_type(4))) unsigned int type __attribute__((aligned(4))); };
template <> struct PacketWordOf<8> { typedef __attribute__((ext_vector_type(4))) unsigned int type __attribute__((aligned(8))); };
template <typename IO>
struct PacketWord {
    typedef typename PacketWordOf<static_cast<int>(sizeof(typename IO::elem))>::type type;
};

template <typename IO>
struct alignas(16) Packet {
    typename IO::elem v[IO::VEC];
};

template <typename IO>
__device__ __forceinline__ Packet<IO> load_packet(const void* base, int64_t elem_index) {
    using V4 = __attribute__((ext_vector_type(4))) unsigned int;
    const V4 raw = *reinterpret_cast<const typename PacketWord<IO>::type*>(static_cast<const typename IO::elem*>(base) + elem_index);
    Packet<IO> p;
    __builtin_memcpy(&p, &raw, 16);
    return p;
}

template <typename IO>
__device__ __forceinline__ Packet<IO> load_packet_nt(const void* base, int64_t elem_index) {
    using V4 = __attribute__((ext_vector_type(4))) unsigned int;
    const V4 raw = __builtin_nontemporal_load(
        reinterpret_cast<const typename PacketWord<IO>::type*>(static_cast<const typename IO::elem*>(base) + elem_index));
    Packet<IO> p;
    __builtin_memcpy(&p, &raw, 16);
    return p;
}

template <typename IO>
__device__ __forceinline__ void store_packet(void* base, int64_t elem_index, const Packet<IO>& p) {
    using V4 = __attribute__((ext_vector_type(4))) unsigned int;
    V4 raw;
    __builtin_memcpy(&raw, &p, 16);
    *reinterpret_cast<typename PacketWord<IO>::type*>(static_cast<typename IO::elem*>(base) + elem_index) = raw;
}

template <typename IO>
__device__ __forceinline__ void store_packet_nt(void* base, int64_t elem_index, const Packet<IO>& p) {
    using V4 = __attribute__((ext_vector_type(4))) unsigned int;
    V4 raw;
    __builtin_memcpy(&raw, &p, 16);
    __builtin_nontemporal_store(raw, reinterpret_cast<typename PacketWord<IO>::type*>(static_cast<typename IO::elem*>(base) + elem_index));
}

// ---- wave64 / block reductions -----------------------------------------------------------------
__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, mask, 64);
    hi = __shfl_xor(hi, mask, 64);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_down_f64(double v, int delta) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_down(lo, delta, 64);
    hi = __shfl_down(hi, delta, 64);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double shfl_up_f64(double v, int delta) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_up(lo, delta, 64);
    hi = __shfl_up(hi, delta, 64);
    return __hiloint2double(hi, lo);
}

// butterfly sum over the 64 lanes of a wavefront; every lane ends with the total
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_f64(v, m);
    return v;
}

}  // namespace lsq
