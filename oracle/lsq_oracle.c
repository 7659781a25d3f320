/* oracle/lsq_oracle.c -- TEST INFRASTRUCTURE ONLY: the CPU oracle for the LSQ fake-quantize hot path.
 *
 * A plain-C restatement of the reference's CPU algorithm
 *   /root/reference/torchlsq/csrc/ops/kernels/lsq_kernel.h   (per-element math)
 *   /root/reference/torchlsq/csrc/ops/global_scope.h         (rounding / fmin / fmax choice)
 *   /root/reference/torchlsq/csrc/ops/cpu/lsq_cpu.cpp        (scalar prep, grad scaler, passes, sums)
 * for float and double.  See lsq_oracle_impl.h for the per-function citations.
 *
 * PARITY STATUS: PINNED.  tests/test_oracle_pinned.py checks this restatement
 *   (a) bit-for-bit against the reference's own lsq_kernel.h compiled from /root/reference
 *       (oracle/_ref/liblsq_ref_scalar.so, built by oracle/build_ref.py), and
 *   (b) against golden vectors produced by the reference's real op library
 *       (oracle/_ref/libtorchlsq_ref_ops.so == its four CPU translation units) with
 *       tests/golden/make_golden.py; the vectors are committed under tests/golden/.
 * The reference itself ships no tests or golden vectors (README.md:176 "TO DO: Add unit tests").
 *
 * Who may use this: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- as the
 * checker / reported baseline only.  The product (lsqfakequantize-pytorch_amd/) never imports,
 * links or executes it and fails loudly when its HIP library is missing.
 *
 * Build: gcc -O2 -fPIC -shared -ffp-contract=off [-fopenmp] lsq_oracle.c -lm   (oracle/build_oracle.py)
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define T float
#define SUF f32
#define T_EPS FLT_EPSILON
#define T_FMIN fminf
#define T_FMAX fmaxf
#define T_RNE nearbyintf
#define T_FABS fabsf
#define T_SQRT sqrtf
#include "lsq_oracle_impl.h"
#undef T
#undef SUF
#undef T_EPS
#undef T_FMIN
#undef T_FMAX
#undef T_RNE
#undef T_FABS
#undef T_SQRT

#define T double
#define SUF f64
#define T_EPS DBL_EPSILON
#define T_FMIN fmin
#define T_FMAX fmax
#define T_RNE nearbyint
#define T_FABS fabs
#define T_SQRT sqrt
#include "lsq_oracle_impl.h"
#undef T
#undef SUF
#undef T_EPS
#undef T_FMIN
#undef T_FMAX
#undef T_RNE
#undef T_FABS
#undef T_SQRT

int lsq_oracle_abi_version(void) { return 1; }

#ifdef _OPENMP
#include <omp.h>
int lsq_oracle_max_threads(void) { return omp_get_max_threads(); }
void lsq_oracle_set_threads(int n) { omp_set_num_threads(n); }
#else
int lsq_oracle_max_threads(void) { return 1; }
void lsq_oracle_set_threads(int n) { (void)n; }
#endif
