// valu_probe.hip -- issue rate of the VALU instructions the bf16 backward is made of (gfx950), to price its
// per-element instruction mix: wave64 instructions per cycle per SIMD for plain fp32, packed fp32, fp64 add,
// fp32->fp64 convert, select, compare, round, min.  Diagnostic tool (tools/), not part of the product.
//   hipcc -O2 --offload-arch=gfx950 tools/probes/valu_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X X X X X X X X

template <int KIND>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float k = 1.0000001f;
    const f2 kk = {k, k};
    const double kd = 1e-30;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {   // v_mul_f32
            asm volatile(
                "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
        } else if (KIND == 1) {   // v_pk_mul_f32
            asm volatile(
                "v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(kk));
        } else if (KIND == 2) {   // v_add_f64
            asm volatile(
                "v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(kd));
        } else if (KIND == 3) {   // v_cvt_f64_f32 (source fixed, results independent)
            asm volatile(
                "v_cvt_f64_f32 %0, %8\n v_cvt_f64_f32 %1, %9\n v_cvt_f64_f32 %2, %10\n v_cvt_f64_f32 %3, %11\n"
                "v_cvt_f64_f32 %4, %8\n v_cvt_f64_f32 %5, %9\n v_cvt_f64_f32 %6, %10\n v_cvt_f64_f32 %7, %11\n"
                : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
                : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
        } else if (KIND == 4) {   // v_cndmask_b32 (vcc)
            asm volatile(
                "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k) : "vcc");
        } else if (KIND == 5) {   // v_cmp_lt_f32 -> sgpr pair
            asm volatile(
                "v_cmp_lt_f32 s[20:21], %0, %8\n v_cmp_lt_f32 s[22:23], %1, %8\n v_cmp_lt_f32 s[24:25], %2, %8\n v_cmp_lt_f32 s[26:27], %3, %8\n"
                "v_cmp_lt_f32 s[20:21], %4, %8\n v_cmp_lt_f32 s[22:23], %5, %8\n v_cmp_lt_f32 s[24:25], %6, %8\n v_cmp_lt_f32 s[26:27], %7, %8\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k)
                : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        } else if (KIND == 6) {   // v_rndne_f32
            asm volatile(
                "v_rndne_f32 %0, %0\n v_rndne_f32 %1, %1\n v_rndne_f32 %2, %2\n v_rndne_f32 %3, %3\n"
                "v_rndne_f32 %4, %4\n v_rndne_f32 %5, %5\n v_rndne_f32 %6, %6\n v_rndne_f32 %7, %7\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == 7) {   // v_min_f32
            asm volatile(
                "v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n"
                "v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_min_f32 %7, %7, %8\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
        } else if (KIND == 8) {   // v_pk_add_f32
            asm volatile(
                "v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(kk));
        } else if (KIND == 9) {   // v_add_f32
            asm volatile(
                "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + static_cast<float>(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + p0.x + p0.y + p1.x + p1.y +
              p2.x + p2.y + p3.x + p3.y + p4.x + p5.y + p6.x + p7.y;
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int KIND>
static void run(const char* name, float* out, int cus, double clk_ghz) {
    const int iters = 20000, grid = cus * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<KIND>, dim3(grid), dim3(256), 0, 0, out, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<KIND>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = static_cast<double>(grid) * 4 * iters * 8;   // 4 waves per workgroup, 8 instructions per iteration
    const double per_s = wave_instr / (ms * 1e-3);
    const double per_simd_cyc = per_s / (cus * 4.0) / (clk_ghz * 1e9);
    printf("%-16s %8.3f ms  %7.2f T wave64-instr/s = %6.2f T lane-ops/s   %.3f instr/cycle/SIMD at %.2f GHz (%.2f cycles per wave64 instruction)\n",
           name, ms, per_s / 1e12, per_s * 64 / 1e12, per_simd_cyc, clk_ghz, 1.0 / per_simd_cyc);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate / 1e6;
    printf("# %s, %d CUs, clockRate %.2f GHz; 8 workgroups x 4 wave64 per CU, 8 independent instructions per loop iteration\n",
           prop.name, cus, clk);
    float* out;
    hipMalloc(&out, 4096);
    run<0>("v_mul_f32", out, cus, clk);
    run<9>("v_add_f32", out, cus, clk);
    run<1>("v_pk_mul_f32", out, cus, clk);
    run<8>("v_pk_add_f32", out, cus, clk);
    run<2>("v_add_f64", out, cus, clk);
    run<3>("v_cvt_f64_f32", out, cus, clk);
    run<4>("v_cndmask_b32", out, cus, clk);
    run<5>("v_cmp_lt_f32", out, cus, clk);
    run<6>("v_rndne_f32", out, cus, clk);
    run<7>("v_min_f32", out, cus, clk);
    hipFree(out);
    return 0;
}
