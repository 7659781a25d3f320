// oracle/ref_scalar_driver.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// A torch-free loop driver around the REFERENCE's own scalar math.  The per-element
// functions are NOT restated here: they are #included, unmodified, from
//   /root/reference/torchlsq/csrc/ops/kernels/lsq_kernel.h   (+ ../global_scope.h)
// by oracle/build_ref.py (-I <reference>/torchlsq/csrc/ops/kernels -DQUANTOPS_CPU), and this
// file only walks plain arrays the way reference lsq_cpu.cpp:49-51,105-135,186-189,252-284
// walks its TensorIterator.  Built into oracle/_ref/liblsq_ref_scalar.so.
//
// Purpose: pin oracle/lsq_oracle.c (the plain-C restatement) bit-for-bit against the
// reference's arithmetic without torch in the loop (tests/test_oracle_vs_reference.py).
#include <cstdint>
#include <tuple>
#include "lsq_kernel.h"

namespace {

template <typename T>
void fwd_pt(const T* x, T* y, int64_t n, T s, T inv_s, T b, T qmin, T qmax, T tmin, T tmax,
            int init_mode) {
    for (int64_t i = 0; i < n; ++i)
        y[i] = lsq_forward_kernel_per_tensor<T>(x[i], s, inv_s, b, qmin, qmax, tmin, tmax,
                                                init_mode != 0);
}

template <typename T>
void bwd_pt(const T* g, const T* x, T* dx, T* ds_buf, T* db_buf, int64_t n, T s, T inv_s, T b,
            T qmin, T qmax, T tmin, T tmax, T grad_scaler, int sym, int eval_mode, int init_mode) {
    for (int64_t i = 0; i < n; ++i) {
        std::tuple<T, T, T> r =
            eval_mode ? lsq_backward_kernel_per_tensor_eval<T>(g[i], x[i], s, inv_s, b, qmin, qmax,
                                                               tmin, tmax, init_mode != 0)
                      : lsq_backward_kernel_per_tensor<T>(g[i], x[i], s, inv_s, b, qmin, qmax, tmin,
                                                          tmax, grad_scaler, sym != 0,
                                                          init_mode != 0);
        dx[i] = std::get<0>(r);
        ds_buf[i] = std::get<1>(r);
        db_buf[i] = std::get<2>(r);
    }
}

template <typename T>
void fwd_pc(const T* x, T* y, int64_t outer, int64_t C, int64_t inner, const T* s, const T* b,
            T qmin, T qmax, T tmin, T tmax, int init_mode, T eps) {
    for (int64_t o = 0; o < outer; ++o)
        for (int64_t c = 0; c < C; ++c)
            for (int64_t k = 0; k < inner; ++k) {
                const int64_t i = (o * C + c) * inner + k;
                y[i] = lsq_forward_kernel_per_channel<T>(x[i], s[c], b[c], qmin, qmax, tmin, tmax,
                                                         init_mode != 0, eps);
            }
}

template <typename T>
void bwd_pc(const T* g, const T* x, T* dx, T* ds_buf, T* db_buf, int64_t outer, int64_t C,
            int64_t inner, const T* s, const T* b, T qmin, T qmax, T tmin, T tmax, T grad_scaler,
            int sym, int eval_mode, int init_mode, T eps) {
    for (int64_t o = 0; o < outer; ++o)
        for (int64_t c = 0; c < C; ++c)
            for (int64_t k = 0; k < inner; ++k) {
                const int64_t i = (o * C + c) * inner + k;
                std::tuple<T, T, T> r =
                    eval_mode ? lsq_backward_kernel_per_channel_eval<T>(g[i], x[i], s[c], b[c], qmin,
                                                                        qmax, tmin, tmax,
                                                                        init_mode != 0, eps)
                              : lsq_backward_kernel_per_channel<T>(g[i], x[i], s[c], b[c], qmin, qmax,
                                                                   tmin, tmax, grad_scaler, sym != 0,
                                                                   init_mode != 0, eps);
                dx[i] = std::get<0>(r);
                ds_buf[i] = std::get<1>(r);
                db_buf[i] = std::get<2>(r);
            }
}

}  // namespace

extern "C" {

#define DEFINE_FOR(T, SUF)                                                                        \
    void ref_fwd_pt_##SUF(const T* x, T* y, int64_t n, T s, T inv_s, T b, T qmin, T qmax, T tmin, \
                          T tmax, int init_mode) {                                                \
        fwd_pt<T>(x, y, n, s, inv_s, b, qmin, qmax, tmin, tmax, init_mode);                       \
    }                                                                                             \
    void ref_bwd_pt_##SUF(const T* g, const T* x, T* dx, T* ds_buf, T* db_buf, int64_t n, T s,    \
                          T inv_s, T b, T qmin, T qmax, T tmin, T tmax, T grad_scaler, int sym,   \
                          int eval_mode, int init_mode) {                                         \
        bwd_pt<T>(g, x, dx, ds_buf, db_buf, n, s, inv_s, b, qmin, qmax, tmin, tmax, grad_scaler,  \
                  sym, eval_mode, init_mode);                                                     \
    }                                                                                             \
    void ref_fwd_pc_##SUF(const T* x, T* y, int64_t outer, int64_t C, int64_t inner, const T* s,  \
                          const T* b, T qmin, T qmax, T tmin, T tmax, int init_mode, T eps) {     \
        fwd_pc<T>(x, y, outer, C, inner, s, b, qmin, qmax, tmin, tmax, init_mode, eps);           \
    }                                                                                             \
    void ref_bwd_pc_##SUF(const T* g, const T* x, T* dx, T* ds_buf, T* db_buf, int64_t outer,     \
                          int64_t C, int64_t inner, const T* s, const T* b, T qmin, T qmax,       \
                          T tmin, T tmax, T grad_scaler, int sym, int eval_mode, int init_mode,   \
                          T eps) {                                                                \
        bwd_pc<T>(g, x, dx, ds_buf, db_buf, outer, C, inner, s, b, qmin, qmax, tmin, tmax,        \
                  grad_scaler, sym, eval_mode, init_mode, eps);                                   \
    }

DEFINE_FOR(float, f32)
DEFINE_FOR(double, f64)

}  // extern "C"
