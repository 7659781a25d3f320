// tools/stream_probe.hip -- HBM ceilings for the hot path's traffic shapes (tuning tool, not product).
// Same lane/tile structure as lsq_per_tensor.hip but with (almost) no arithmetic:
//   kind 0: copy   y = x            (1 read : 1 write  -- the forward's shape)
//   kind 1: add    y = g + x        (2 reads : 1 write -- the backward's shape)
//   kind 2: read2  sum(g + x)       (2 reads : 0 writes)
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
using V4 = __attribute__((ext_vector_type(4))) float;

template <int KIND, int UNROLL, int BLOCK>
__global__ __launch_bounds__(BLOCK) void probe_kernel(const V4* __restrict__ x, const V4* __restrict__ g,
                                                      V4* __restrict__ y, int64_t n_packets, float* sink) {
    constexpr int64_t kTile = static_cast<int64_t>(BLOCK) * UNROLL;
    const int64_t n_full = n_packets / kTile;
    V4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int64_t tile = blockIdx.x; tile < n_full; tile += gridDim.x) {
        const int64_t p0 = tile * kTile + threadIdx.x;
        V4 a[UNROLL], b[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            a[u] = __builtin_nontemporal_load(x + p0 + u * BLOCK);
            if (KIND != 0) b[u] = __builtin_nontemporal_load(g + p0 + u * BLOCK);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (KIND == 0) __builtin_nontemporal_store(a[u], y + p0 + u * BLOCK);
            if (KIND == 1) __builtin_nontemporal_store(a[u] + b[u], y + p0 + u * BLOCK);
            if (KIND == 2) acc += a[u] + b[u];
        }
    }
    if (KIND == 2 && acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
}
}  // namespace

extern "C" int lsq_probe_run(int kind, int unroll, int block, const void* x, const void* g, void* y, int64_t n_elems,
                             int grid, float* sink, void* stream) {
    const int64_t np = n_elems / 4;
    hipStream_t s = static_cast<hipStream_t>(stream);
#define RUN(K, U, B)                                                                                          \
    hipLaunchKernelGGL((probe_kernel<K, U, B>), dim3(grid), dim3(B), 0, s, static_cast<const V4*>(x),         \
                       static_cast<const V4*>(g), static_cast<V4*>(y), np, sink)
#define BY_UB(K)                                                  \
    if (block == 256) { if (unroll == 4) RUN(K, 4, 256); else if (unroll == 8) RUN(K, 8, 256); else RUN(K, 2, 256); } \
    else if (block == 512) { if (unroll == 4) RUN(K, 4, 512); else if (unroll == 8) RUN(K, 8, 512); else RUN(K, 2, 512); } \
    else { if (unroll == 4) RUN(K, 4, 1024); else if (unroll == 8) RUN(K, 8, 1024); else RUN(K, 2, 1024); }
    if (kind == 0) { BY_UB(0) } else if (kind == 1) { BY_UB(1) } else { BY_UB(2) }
    return static_cast<int>(hipGetLastError());
}
