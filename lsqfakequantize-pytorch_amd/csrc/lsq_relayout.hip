// lsq_relayout.hip -- a gradient that arrives in ANOTHER dense memory order than the input it belongs to.
//
// The reference's backward iterates (dx, ds_buffer, db_buffer, grad, x) through ONE TensorIterator
// (/root/reference/torchlsq/csrc/ops/cpu/lsq_cpu.cpp:80-90, :229-249): whatever the strides of `grad` are, element i of grad
// meets element i of x, and dx takes x's layout (MemoryFormat::Preserve).  This library's kernels walk x's MEMORY order with
// 16-byte packets, so a grad in another dense order -- the one real case: a channels-last activation whose upstream gradient
// is contiguous NCHW, or the reverse -- has to be brought into x's order first.  The host layers used to do that with the
// framework's generic strided copy (1.9-2.0 TB/s on MI355X for [256,2048,7,7]: +104 us on a 61 us backward,
// profiles/r06_layout_workloads.txt); this is the same pass as an LDS-tiled transposition:
//
//     dst[a][c][b] = src[a][b][c]        a < A, b < B, c < C        (NCHW -> NHWC: A = N, B = C, C = H*W; the reverse: B = H*W, C = C)
//
// 64 x 64 tiles through LDS (row pitch 65 elements: the column-wise read-out is bank-conflict-free); the read side walks
// runs of min(C, 64) contiguous elements -- a tile that spans the whole of C is ONE contiguous chunk of the source --, the
// write side runs of min(B, 64).  HBM-bound: 2 elements of traffic per element, nothing else.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lsq_kernels.hpp"

namespace lsq {

constexpr int kRelTile = 64;

template <typename W>
__global__ __launch_bounds__(kBlock) void relayout_kernel(const W* __restrict__ src, W* __restrict__ dst, int64_t B, int64_t C,
                                                          int64_t tiles_b, int64_t tiles_c) {
    __shared__ W tile[kRelTile][kRelTile + 1];
    const int64_t t = blockIdx.x;
    const int64_t tc = t % tiles_c;
    const int64_t tb = (t / tiles_c) % tiles_b;
    const int64_t a = t / (tiles_c * tiles_b);
    const int64_t b0 = tb * kRelTile, c0 = tc * kRelTile;
    const int nb = static_cast<int>(B - b0 < kRelTile ? B - b0 : kRelTile);
    const int nc = static_cast<int>(C - c0 < kRelTile ? C - c0 : kRelTile);
    const W* s = src + (a * B + b0) * C + c0;
    W* d = dst + (a * C + c0) * B + b0;
    const int lane = threadIdx.x & 63, row0 = threadIdx.x >> 6;          // 4 waves: wave w takes rows w, w + 4, ...
    if (nc == C) {
        // the tile's nb rows of the source are ONE contiguous run of nb * C elements: read it flat (full wave instructions
        // whatever C is -- C = 49 would otherwise leave 15 lanes of every row instruction idle)
        const int n = nb * nc;
        const int q = kBlock / nc, r = kBlock - q * nc;                   // one division per lane, then (b, c) advance by (q, r)
        int b = static_cast<int>(threadIdx.x) / nc, c = static_cast<int>(threadIdx.x) - b * nc;
        for (int i = threadIdx.x; i < n; i += kBlock) {
            tile[b][c] = s[i];
            b += q; c += r;
            if (c >= nc) { c -= nc; ++b; }
        }
    } else {
#pragma unroll 4
        for (int b = row0; b < nb; b += kBlock / 64)
            if (lane < nc) tile[b][lane] = s[static_cast<int64_t>(b) * C + lane];
    }
    __syncthreads();
    if (nb == B) {
        const int n = nb * nc;
        const int q = kBlock / nb, r = kBlock - q * nb;
        int c = static_cast<int>(threadIdx.x) / nb, b = static_cast<int>(threadIdx.x) - c * nb;
        for (int i = threadIdx.x; i < n; i += kBlock) {
            d[i] = tile[b][c];
            c += q; b += r;
            if (b >= nb) { b -= nb; ++c; }
        }
    } else {
#pragma unroll 4
        for (int c = row0; c < nc; c += kBlock / 64)
            if (lane < nb) d[static_cast<int64_t>(c) * B + lane] = tile[lane][c];
    }
}

hipError_t relayout(int elem_bytes, const void* src, void* dst, int64_t A, int64_t B, int64_t C, hipStream_t stream) {
    const int64_t tiles_b = (B + kRelTile - 1) / kRelTile, tiles_c = (C + kRelTile - 1) / kRelTile;
    const int64_t tiles = A * tiles_b * tiles_c;
    if (tiles <= 0) return hipSuccess;
    if (tiles > INT32_MAX) return hipErrorInvalidConfiguration;
    const dim3 grid(static_cast<unsigned>(tiles));
    switch (elem_bytes) {
        case 2:
            hipLaunchKernelGGL((relayout_kernel<uint16_t>), grid, dim3(kBlock), 0, stream, static_cast<const uint16_t*>(src),
                               static_cast<uint16_t*>(dst), B, C, tiles_b, tiles_c);
            break;
        case 4:
            hipLaunchKernelGGL((relayout_kernel<uint32_t>), grid, dim3(kBlock), 0, stream, static_cast<const uint32_t*>(src),
                               static_cast<uint32_t*>(dst), B, C, tiles_b, tiles_c);
            break;
        case 8:
            hipLaunchKernelGGL((relayout_kernel<uint64_t>), grid, dim3(kBlock), 0, stream, static_cast<const uint64_t*>(src),
                               static_cast<uint64_t*>(dst), B, C, tiles_b, tiles_c);
            break;
        default:
            return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace lsq
