// tools/probes/window_probe.hip -- does the WIDTH of the contiguous piece a workgroup reads per row matter on cold data?
// No-arithmetic 2R:1W (y = g + x) kernel with the access pattern of the window-mode per-channel kernels: the tensor is
// [rows][L]; workgroup (w, s) owns window w -- 256 lanes x P packets of 16 bytes, contiguous within the row -- of the rows
// of split s and walks them U rows at a time (P x U loads of g and of x in flight per lane).  P = 1 is what the product
// does (4 KiB per row and workgroup); P = 4 reads 16 KiB per row like a K1/K2 tile.  Tuning tool, not product.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
using V4 = __attribute__((ext_vector_type(4))) float;

template <int P, int U>
__global__ __launch_bounds__(256) void window_add(const V4* __restrict__ x, const V4* __restrict__ g, V4* __restrict__ y,
                                                  int64_t rows, int64_t row_packets, int64_t rows_per_split) {
    const int64_t p_base = static_cast<int64_t>(blockIdx.x) * (256 * P) + threadIdx.x;
    const int64_t r0 = static_cast<int64_t>(blockIdx.y) * rows_per_split;
    const int64_t r1 = r0 + rows_per_split < rows ? r0 + rows_per_split : rows;
    for (int64_t r = r0; r < r1; r += U) {
        V4 a[U][P], b[U][P];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t rr = r + u < r1 ? r + u : r1 - 1;
#pragma unroll
            for (int k = 0; k < P; ++k) {
                int64_t p = p_base + k * 256;
                p = p < row_packets ? p : row_packets - 1;
                a[u][k] = __builtin_nontemporal_load(x + rr * row_packets + p);
                b[u][k] = __builtin_nontemporal_load(g + rr * row_packets + p);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (r + u < r1) {
#pragma unroll
                for (int k = 0; k < P; ++k) {
                    const int64_t p = p_base + k * 256;
                    if (p < row_packets) __builtin_nontemporal_store(a[u][k] + b[u][k], y + (r + u) * row_packets + p);
                }
            }
        }
    }
}
}  // namespace

extern "C" int window_probe_run(int p, int u, const void* x, const void* g, void* y, int64_t rows, int64_t row_elems,
                                int splits, void* stream) {
    const int64_t rp = row_elems / 4;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t rps = (rows + splits - 1) / splits;
#define RUN(PP, UU)                                                                                                    \
    hipLaunchKernelGGL((window_add<PP, UU>), dim3(static_cast<unsigned>((rp + 256 * PP - 1) / (256 * PP)), splits), dim3(256), 0, \
                       s, static_cast<const V4*>(x), static_cast<const V4*>(g), static_cast<V4*>(y), rows, rp, rps)
    if (p == 1 && u == 4) RUN(1, 4);
    else if (p == 1 && u == 8) RUN(1, 8);
    else if (p == 2 && u == 2) RUN(2, 2);
    else if (p == 2 && u == 4) RUN(2, 4);
    else if (p == 4 && u == 1) RUN(4, 1);
    else if (p == 4 && u == 2) RUN(4, 2);
    else if (p == 8 && u == 1) RUN(8, 1);
    else return -1;
    return static_cast<int>(hipGetLastError());
}
