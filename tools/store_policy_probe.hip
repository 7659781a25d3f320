// tools/store_policy_probe.hip -- does the cache policy of the streaming stores change the 2R:1W / 1R:1W rate or the
// cost of a kernel boundary?  (tuning tool, not product).  Same lane/tile structure as lsq_per_tensor.hip.
//   policy 0: plain store   1: nt   2: sc0 sc1   3: sc1   4: sc0 sc1 nt   5: sc0
//   loads: ld 0 plain, 1 nt
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
using V4 = __attribute__((ext_vector_type(4))) float;

template <int POLICY>
__device__ __forceinline__ void store16(V4* p, V4 v) {
    if constexpr (POLICY == 0) *p = v;
    else if constexpr (POLICY == 1) __builtin_nontemporal_store(v, p);
    else if constexpr (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POLICY == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POLICY == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
}

template <int KIND, int POLICY, int LD>
__global__ __launch_bounds__(256) void policy_kernel(const V4* __restrict__ x, const V4* __restrict__ g, V4* __restrict__ y,
                                                     int64_t n_packets) {
    constexpr int UNROLL = 4;
    constexpr int64_t kTile = 256 * UNROLL;
    const int64_t n_full = n_packets / kTile;
    for (int64_t tile = blockIdx.x; tile < n_full; tile += gridDim.x) {
        const int64_t p0 = tile * kTile + threadIdx.x;
        V4 a[UNROLL], b[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            a[u] = LD ? __builtin_nontemporal_load(x + p0 + u * 256) : x[p0 + u * 256];
            if (KIND == 1) b[u] = LD ? __builtin_nontemporal_load(g + p0 + u * 256) : g[p0 + u * 256];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) store16<POLICY>(y + p0 + u * 256, KIND == 1 ? a[u] + b[u] : a[u]);
    }
}
}  // namespace

extern "C" int policy_probe_run(int kind, int policy, int ld, const void* x, const void* g, void* y, int64_t n_elems, int grid,
                                void* stream) {
    const int64_t np = n_elems / 4;
    hipStream_t s = static_cast<hipStream_t>(stream);
#define RUN(K, P, L) hipLaunchKernelGGL((policy_kernel<K, P, L>), dim3(grid), dim3(256), 0, s, static_cast<const V4*>(x), \
                                        static_cast<const V4*>(g), static_cast<V4*>(y), np)
#define BY_L(K, P) if (ld) RUN(K, P, 1); else RUN(K, P, 0)
#define BY_P(K) switch (policy) { case 0: BY_L(K, 0); break; case 1: BY_L(K, 1); break; case 2: BY_L(K, 2); break; \
                                  case 3: BY_L(K, 3); break; case 4: BY_L(K, 4); break; default: BY_L(K, 5); break; }
    if (kind == 0) { BY_P(0) } else { BY_P(1) }
    return static_cast<int>(hipGetLastError());
}
