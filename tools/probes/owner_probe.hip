// tools/probes/owner_probe.hip -- can a workgroup OWN its channels for all rows (so that the per-channel backward needs no
// partials and no finalize launch) and still stream at the HBM rate?  (round-4 experiment (c), DESIGN_HISTORY.md section 7.)
// No-arithmetic 2R:1W (y = g + x on 16-byte packets) kernel with that access pattern on the [rows][L] view of an NCHW
// activation quantized on axis 1 (L = C * inner): workgroup j owns `run` consecutive 16-byte packets of every row -- k whole
// channels when run * 16 = k * inner * elem_size -- and its `waves` waves deal the (row, packet) pairs of those runs among
// themselves, U wave-iterations of 64 packets in flight each.  xcd = 1 remaps blockIdx so that neighbouring runs (which
// share a 128-byte line at each end when run * 16 is not a multiple of 128) land on the same XCD's L2.
// Tuning tool, not product.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
using V4 = __attribute__((ext_vector_type(4))) float;

template <int U>
__global__ void owner_add(const V4* __restrict__ x, const V4* __restrict__ g, V4* __restrict__ y, int64_t rows,
                          int64_t row_packets, int run, int xcd) {
    int64_t j = blockIdx.x;
    if (xcd) {                                   // consecutive owners on one XCD (blocks are dealt round-robin over 8 XCDs)
        const int64_t per = (gridDim.x + 7) / 8;
        j = (blockIdx.x % 8) * per + blockIdx.x / 8;
        if (j >= gridDim.x) return;              // (only exact for grids that are a multiple of 8: the probe's are)
    }
    const int64_t p0 = j * run;
    const int64_t total = rows * static_cast<int64_t>(run);     // (row, packet) pairs of this owner
    const int64_t step = static_cast<int64_t>(blockDim.x) * U;
    for (int64_t i0 = threadIdx.x; i0 < total; i0 += step) {
        V4 a[U], b[U];
        int64_t e[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int64_t i = i0 + static_cast<int64_t>(u) * blockDim.x;
            i = i < total ? i : total - 1;
            const int64_t r = i / run, q = i - r * run;
            e[u] = r * row_packets + p0 + q;
            a[u] = __builtin_nontemporal_load(x + e[u]);
            b[u] = __builtin_nontemporal_load(g + e[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i0 + static_cast<int64_t>(u) * blockDim.x < total) __builtin_nontemporal_store(a[u] + b[u], y + e[u]);
    }
}
}  // namespace

extern "C" int owner_probe_run(int u, int waves, int run, int xcd, const void* x, const void* g, void* y, int64_t rows,
                               int64_t row_packets, void* stream) {
    if (run <= 0 || row_packets % run != 0 || waves < 1 || waves > 16) return -1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned grid = static_cast<unsigned>(row_packets / run);
#define RUN(UU) hipLaunchKernelGGL((owner_add<UU>), dim3(grid), dim3(64 * waves), 0, s, static_cast<const V4*>(x), \
                                   static_cast<const V4*>(g), static_cast<V4*>(y), rows, row_packets, run, xcd)
    if (u == 1) RUN(1);
    else if (u == 2) RUN(2);
    else if (u == 4) RUN(4);
    else return -1;
    return static_cast<int>(hipGetLastError());
}
