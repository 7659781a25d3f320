// tools/probes/colblock_probe.hip -- access-pattern probe for the last-axis (token layout) per-channel backward, round 5:
// does a COLUMN-BLOCK x ROW-SLAB decomposition of a [rows][row_packets] tensor (workgroup = w packet columns of a slab of rows,
// its lanes R = block / w row groups) stream like whole-row workgroups do?  Narrow column blocks shrink the partial-sum rows a
// workgroup has to publish (w x V channels instead of the whole row) -- if the pattern itself holds the HBM rate.
// No-arithmetic 2R:1W (y = g + x on 16-byte packets), U rows in flight per lane.  Tuning tool, not product.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
using V4 = __attribute__((ext_vector_type(4))) float;

template <int U>
__global__ void colblock_add(const V4* __restrict__ x, const V4* __restrict__ g, V4* __restrict__ y, int64_t rows,
                             int row_packets, int w, int n_cb, int slabs) {
    const int cb = blockIdx.x % n_cb, slab = blockIdx.x / n_cb;
    const int R = blockDim.x / w;
    const int rg = threadIdx.x / w, l = threadIdx.x - rg * w;
    if (rg >= R) return;
    const int64_t r0 = rows * slab / slabs, r1 = rows * (slab + 1) / slabs;
    const int64_t col = static_cast<int64_t>(cb) * w + l;
    for (int64_t r = r0 + rg; r < r1; r += static_cast<int64_t>(R) * U) {
        V4 a[U], b[U];
        int64_t e[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int64_t rr = r + static_cast<int64_t>(u) * R;
            rr = rr < r1 ? rr : r1 - 1;
            e[u] = rr * row_packets + col;
            a[u] = __builtin_nontemporal_load(x + e[u]);
            b[u] = __builtin_nontemporal_load(g + e[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (r + static_cast<int64_t>(u) * R < r1) __builtin_nontemporal_store(a[u] + b[u], y + e[u]);
    }
}
}  // namespace

extern "C" int colblock_probe_run(int u, int block, int w, int slabs, const void* x, const void* g, void* y, int64_t rows,
                                  int row_packets, void* stream) {
    if (w <= 0 || row_packets % w != 0 || block % 64 != 0 || block < w || slabs < 1) return -1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int n_cb = row_packets / w;
    const unsigned grid = static_cast<unsigned>(n_cb) * static_cast<unsigned>(slabs);
#define RUN(UU) hipLaunchKernelGGL((colblock_add<UU>), dim3(grid), dim3(block), 0, s, static_cast<const V4*>(x), \
                                   static_cast<const V4*>(g), static_cast<V4*>(y), rows, row_packets, w, n_cb, slabs)
    if (u == 1) RUN(1);
    else if (u == 2) RUN(2);
    else if (u == 4) RUN(4);
    else return -1;
    return static_cast<int>(hipGetLastError());
}
