/* oracle/lsq_oracle_impl.h -- TEST INFRASTRUCTURE ONLY.  Included twice by lsq_oracle.c with
 *   T      = float / double      SUF    = f32 / f64
 *   T_EPS  = FLT_EPSILON / DBL_EPSILON   (std::numeric_limits<scalar_t>::epsilon(), lsq_cpu.cpp:45-46)
 *   T_FMIN / T_FMAX / T_RNE / T_FABS / T_SQRT = fminf.. / fmin..   (global_scope.h:12,19-20: FASTROUND =
 *                                               std::nearbyint, FMIN/FMAX = std::fmin/std::fmax, ABS = std::abs)
 * Plain-C restatement of the reference CPU path.  Every function cites the reference lines it
 * follows (paths relative to /root/reference/torchlsq/csrc/ops/).
 *
 * Arithmetic rules that make the restatement bit-exact with the reference build (g++ -O3 for
 * baseline x86-64, i.e. no FMA contraction): every product and sum below is a separate,
 * individually rounded operation in T; lsq_oracle.c is compiled with -ffp-contract=off.
 */

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

/* kernels/lsq_kernel.h:12 (and :30,:51,:76,:108,:138): the zero point every element recomputes */
static inline T FN(zero_point)(T b, T inv_s, T tmin, T tmax) {
    return T_RNE(T_FMIN(tmax, T_FMAX(tmin, -b * inv_s)));
}

/* kernels/lsq_kernel.h:6-14  lsq_forward_kernel_per_tensor */
static inline T FN(fwd_elem)(T x, T s, T inv_s, T b, T qmin, T qmax, T tmin, T tmax, int init_mode) {
    const T zp = FN(zero_point)(b, inv_s, tmin, tmax);
    if (init_mode) return x;
    return (T_RNE(T_FMIN(qmax, T_FMAX(qmin, x * inv_s + zp))) - zp) * s;
}

/* the integer level the forward rounds to: FASTROUND(FMIN(qmax, FMAX(qmin, x*inv_s + zp))),
 * kernels/lsq_kernel.h:13 -- exposed so the "quantized integer values bit-exact" bar has a carrier */
static inline T FN(level_elem)(T x, T inv_s, T b, T qmin, T qmax, T tmin, T tmax) {
    const T zp = FN(zero_point)(b, inv_s, tmin, tmax);
    return T_RNE(T_FMIN(qmax, T_FMAX(qmin, x * inv_s + zp)));
}

/* kernels/lsq_kernel.h:94-123 lsq_backward_kernel_per_tensor (fused dX,dS,dB) and
 * kernels/lsq_kernel.h:126-145 lsq_backward_kernel_per_tensor_eval */
static inline void FN(bwd_elem)(T grad, T x, T s, T inv_s, T b, T qmin, T qmax, T tmin, T tmax,
                                T grad_scaler, int sym, int eval_mode, int init_mode, T* dX, T* dS,
                                T* dB) {
    const T zp = FN(zero_point)(b, inv_s, tmin, tmax);
    const T xq = T_FMAX(T_FMIN(x * inv_s + zp, qmax), qmin); /* :108 -- clamp order min, then max; NOT rounded */
    const int mask = (qmin < xq) && (xq < qmax);             /* :109 strict, on the unrounded xq */
    *dX = init_mode ? grad : (grad * (T)mask);               /* :112 / :140 */
    if (eval_mode) {                                         /* :142-144 */
        *dS = (T)0;
        *dB = (T)0;
        return;
    }
    const T xfq = (T_RNE(xq) - zp) * s;                               /* :115 */
    const T g_ = init_mode ? (T)((T)2 * (xfq - x)) : grad;            /* :116 */
    const T db_ = sym ? (T)0 : ((T)(!mask) * g_);                     /* :118 */
    const T border = (xq <= qmin) ? (g_ * (qmin - zp)) : (g_ * (qmax - zp)); /* :120 */
    const T ds_ = mask ? (T)(g_ * (xfq - x) * inv_s) : border;        /* :121 */
    *dS = ds_ * grad_scaler;                                          /* :122 */
    *dB = db_ * grad_scaler;
}

/* cpu/lsq_cpu.cpp:45-47 (fwd) and :100-102 (bwd): s = max(|scale[0]|, eps), inv_s = 1/s */
static inline T FN(sanitize_scale)(T scale0) {
    const T a = T_FABS(scale0);
    return (a < T_EPS) ? T_EPS : a; /* std::max(a, eps) == (a < eps) ? eps : a  (a NaN scale stays NaN) */
}

/* cpu/lsq_cpu.cpp:103-104: use_grad_scaling ? (scalar_t)(grad_scaler / sqrt(x.numel()*qmax)) : (scalar_t)grad_scaler
 * `x.numel()*qmax` is int64 * scalar_t -> scalar_t; sqrt resolves to the scalar_t overload; the
 * division is double / (double)scalar_t; the cast rounds once to scalar_t. */
T FN(lsq_oracle_grad_scaler_pt)(int64_t numel, int quant_max, int use_grad_scaling, double grad_scaler) {
    if (!use_grad_scaling) return (T)grad_scaler;
    const T prod = (T)numel * (T)quant_max;
    const T r = T_SQRT(prod);
    return (T)(grad_scaler / (double)r);
}

/* cpu/lsq_cpu.cpp:250-251: grad_scaler / sqrt(x.numel()*qmax / x.size(axis)) */
T FN(lsq_oracle_grad_scaler_pc)(int64_t numel, int quant_max, int64_t channels, int use_grad_scaling,
                                double grad_scaler) {
    if (!use_grad_scaling) return (T)grad_scaler;
    const T prod = (T)numel * (T)quant_max;
    const T q = prod / (T)channels;
    const T r = T_SQRT(q);
    return (T)(grad_scaler / (double)r);
}

/* cpu/lsq_cpu.cpp:15-53 lsq_forward_per_tensor_impl (dense memory order; strides are the
 * caller's business, as the TensorIterator visits every element exactly once) */
void FN(lsq_oracle_fwd_pt)(const T* x, T* y, int64_t n, T scale0, T shift0, int quant_min,
                           int quant_max, int type_min, int type_max, int init_mode) {
    const T qmin = (T)quant_min, qmax = (T)quant_max, tmin = (T)type_min, tmax = (T)type_max;
    const T b = shift0;
    const T s = FN(sanitize_scale)(scale0);
    const T inv_s = (T)1 / s;
    int64_t i;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (i = 0; i < n; ++i) y[i] = FN(fwd_elem)(x[i], s, inv_s, b, qmin, qmax, tmin, tmax, init_mode);
}

/* integer levels of the forward (kernels/lsq_kernel.h:13), as int32 */
void FN(lsq_oracle_levels_pt)(const T* x, int32_t* q, int64_t n, T scale0, T shift0, int quant_min,
                              int quant_max, int type_min, int type_max) {
    const T qmin = (T)quant_min, qmax = (T)quant_max, tmin = (T)type_min, tmax = (T)type_max;
    const T s = FN(sanitize_scale)(scale0);
    const T inv_s = (T)1 / s;
    int64_t i;
    for (i = 0; i < n; ++i) q[i] = (int32_t)FN(level_elem)(x[i], inv_s, shift0, qmin, qmax, tmin, tmax);
}

/* cpu/lsq_cpu.cpp:56-141 lsq_backward_per_tensor_impl: ONE fused pass writing dx, ds_buffer,
 * db_buffer (:105-135), then ds = ds_buffer.sum(), db = db_buffer.sum() (:138-139).
 * ATen's Tensor::sum is PyTorch code outside the reference repo and its summation order is an
 * implementation detail; the restatement returns the (near-)exact fp64 sum of the T-typed
 * per-element terms in ds_wide / db_wide and its rounding to T in ds / db.  ds_buf/db_buf may be
 * NULL (terms are then not materialised).  numel_for_scaler is x.numel() for the reference's
 * own behaviour (:103); the sharded multi-GPU path passes the GLOBAL numel instead.
 * abs_ds/abs_db (may be NULL) receive sum|term| -- the scale of the tolerance for mixed-sign sums. */
void FN(lsq_oracle_bwd_pt)(const T* g, const T* x, T* dx, T* ds_buf, T* db_buf, int64_t n, T scale0,
                           T shift0, int quant_min, int quant_max, int type_min, int type_max,
                           int use_grad_scaling, double grad_scaler, int64_t numel_for_scaler, int sym,
                           int eval_mode, int init_mode, T* ds, T* db, double* ds_wide,
                           double* db_wide, double* abs_ds, double* abs_db) {
    const T qmin = (T)quant_min, qmax = (T)quant_max, tmin = (T)type_min, tmax = (T)type_max;
    const T b = shift0;
    const T s = FN(sanitize_scale)(scale0);
    const T inv_s = (T)1 / s;
    const T gs = FN(lsq_oracle_grad_scaler_pt)(numel_for_scaler, quant_max, use_grad_scaling, grad_scaler);
    double acc_s = 0.0, acc_b = 0.0, aabs_s = 0.0, aabs_b = 0.0;
    int64_t i;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) reduction(+ : acc_s, acc_b, aabs_s, aabs_b)
#endif
    for (i = 0; i < n; ++i) {
        T dX, dS, dB;
        FN(bwd_elem)(g[i], x[i], s, inv_s, b, qmin, qmax, tmin, tmax, gs, sym, eval_mode, init_mode,
                     &dX, &dS, &dB);
        dx[i] = dX;
        if (ds_buf) ds_buf[i] = dS;
        if (db_buf) db_buf[i] = dB;
        acc_s += (double)dS;
        acc_b += (double)dB;
        aabs_s += fabs((double)dS);
        aabs_b += fabs((double)dB);
    }
    if (ds) *ds = (T)acc_s;
    if (db) *db = (T)acc_b;
    if (ds_wide) *ds_wide = acc_s;
    if (db_wide) *db_wide = acc_b;
    if (abs_ds) *abs_ds = aabs_s;
    if (abs_db) *abs_db = aabs_b;
}

/* cpu/lsq_cpu.cpp:145-193 lsq_forward_per_channel_impl + kernels/lsq_kernel.h:151-160: the tensor
 * is viewed as [outer, C, inner] in memory order (scale/shift viewed [1..C..1] and broadcast,
 * :168-176); per element _s = fmax(eps, |s|), inv_s = 1/_s (lsq_kernel.h:157-158). */
void FN(lsq_oracle_fwd_pc)(const T* x, T* y, int64_t outer, int64_t C, int64_t inner, const T* scale,
                           const T* shift, int quant_min, int quant_max, int type_min, int type_max,
                           int init_mode) {
    const T qmin = (T)quant_min, qmax = (T)quant_max, tmin = (T)type_min, tmax = (T)type_max;
    int64_t r;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (r = 0; r < outer * C; ++r) {
        const int64_t c = r % C;
        const T s = T_FMAX(T_EPS, T_FABS(scale[c]));
        const T inv_s = (T)1 / s;
        const T* xr = x + r * inner;
        T* yr = y + r * inner;
        int64_t k;
        for (k = 0; k < inner; ++k)
            yr[k] = FN(fwd_elem)(xr[k], s, inv_s, shift[c], qmin, qmax, tmin, tmax, init_mode);
    }
}

void FN(lsq_oracle_levels_pc)(const T* x, int32_t* q, int64_t outer, int64_t C, int64_t inner,
                              const T* scale, const T* shift, int quant_min, int quant_max,
                              int type_min, int type_max) {
    const T qmin = (T)quant_min, qmax = (T)quant_max, tmin = (T)type_min, tmax = (T)type_max;
    int64_t r, k;
    for (r = 0; r < outer * C; ++r) {
        const int64_t c = r % C;
        const T s = T_FMAX(T_EPS, T_FABS(scale[c]));
        const T inv_s = (T)1 / s;
        for (k = 0; k < inner; ++k)
            q[r * inner + k] = (int32_t)FN(level_elem)(x[r * inner + k], inv_s, shift[c], qmin, qmax, tmin, tmax);
    }
}

/* cpu/lsq_cpu.cpp:197-294 lsq_backward_per_channel_impl + kernels/lsq_kernel.h:219-256.
 * ds[c] = sum over (outer, inner) of the per-element terms (:287-292).  Outputs as in bwd_pt,
 * arrays of length C.  numel_for_scaler: x.numel() (:250) or the global numel when sharded. */
void FN(lsq_oracle_bwd_pc)(const T* g, const T* x, T* dx, T* ds_buf, T* db_buf, int64_t outer,
                           int64_t C, int64_t inner, const T* scale, const T* shift, int quant_min,
                           int quant_max, int type_min, int type_max, int use_grad_scaling,
                           double grad_scaler, int64_t numel_for_scaler, int sym, int eval_mode,
                           int init_mode, T* ds, T* db, double* ds_wide, double* db_wide,
                           double* abs_ds, double* abs_db) {
    const T qmin = (T)quant_min, qmax = (T)quant_max, tmin = (T)type_min, tmax = (T)type_max;
    const T gs = FN(lsq_oracle_grad_scaler_pc)(numel_for_scaler, quant_max, C, use_grad_scaling, grad_scaler);
    int64_t c;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (c = 0; c < C; ++c) {
        const T s = T_FMAX(T_EPS, T_FABS(scale[c]));
        const T inv_s = (T)1 / s;
        double acc_s = 0.0, acc_b = 0.0, aabs_s = 0.0, aabs_b = 0.0;
        int64_t o, k;
        for (o = 0; o < outer; ++o) {
            const int64_t base = (o * C + c) * inner;
            for (k = 0; k < inner; ++k) {
                T dX, dS, dB;
                FN(bwd_elem)(g[base + k], x[base + k], s, inv_s, shift[c], qmin, qmax, tmin, tmax, gs,
                             sym, eval_mode, init_mode, &dX, &dS, &dB);
                dx[base + k] = dX;
                if (ds_buf) ds_buf[base + k] = dS;
                if (db_buf) db_buf[base + k] = dB;
                acc_s += (double)dS;
                acc_b += (double)dB;
                aabs_s += fabs((double)dS);
                aabs_b += fabs((double)dB);
            }
        }
        if (ds) ds[c] = (T)acc_s;
        if (db) db[c] = (T)acc_b;
        if (ds_wide) ds_wide[c] = acc_s;
        if (db_wide) db_wide[c] = acc_b;
        if (abs_ds) abs_ds[c] = aabs_s;
        if (abs_db) abs_db[c] = aabs_b;
    }
}

#undef FN
#undef CAT
#undef CAT_
