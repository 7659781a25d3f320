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
// write side runs of min(B, 64).  HBM-bound in fp32 (4.9 TB/s on [256,2048,7,7], the rate of a contiguous copy); 2-byte
// elements run at the same ELEMENT rate (2.8 TB/s: 128-byte runs scattered at stride B) -- a variant with 128 x 64 tiles and
// 4-byte pair stores was measured at half of that (140 VGPRs, three workgroups per CU) and dropped.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lsq_kernels.hpp"

namespace lsq {

constexpr int kRelTile = 64;

// Persistent workgroups, tiles round-robin; the NEXT tile's global loads are issued (into registers) before the current
// tile's stores, so a workgroup always has a tile of reads or a tile of writes in flight.
template <typename W>
__global__ __launch_bounds__(kBlock) void relayout_kernel(const W* __restrict__ src, W* __restrict__ dst, int64_t B, int64_t C,
                                                          int64_t tiles_b, int64_t tiles_c, int64_t tiles) {
    constexpr int TB = kRelTile;
    constexpr int kPerLane = TB * kRelTile / kBlock;       // 16 elements of a tile per lane
    __shared__ W tile[TB][kRelTile + 1];
    const int tid = static_cast<int>(threadIdx.x), lane = tid & 63, row0 = tid >> 6;       // 4 waves: wave w takes rows w, w + 4, ...
    // A tile that spans the whole of C is ONE contiguous run of nb * C source elements (read flat: full wave instructions
    // whatever C is -- C = 49 would otherwise leave 15 lanes of every row instruction idle); likewise the destination when a
    // tile spans the whole of B.  Element i = tid + 256 k of such a run sits at (i / n, i % n): the same for every tile.
    const bool flat_r = C <= kRelTile, flat_w = B <= TB;
    const int nc_all = static_cast<int>(C < kRelTile ? C : kRelTile), nb_all = static_cast<int>(B < TB ? B : TB);
    int rq[kPerLane], wq[kPerLane];          // (row << 8 | column) of this lane's k-th element, read side / write side
    {
        const int q = kBlock / nc_all, r = kBlock - q * nc_all;
        int b = tid / nc_all, c = tid - b * nc_all;
#pragma unroll
        for (int k = 0; k < kPerLane; ++k) {
            rq[k] = (b << 8) | c;
            b += q; c += r;
            if (c >= nc_all) { c -= nc_all; ++b; }
        }
        const int q2 = kBlock / nb_all, r2 = kBlock - q2 * nb_all;
        int cc = tid / nb_all, bb = tid - cc * nb_all;
#pragma unroll
        for (int k = 0; k < kPerLane; ++k) {
            wq[k] = (bb << 8) | cc;
            cc += q2; bb += r2;
            if (bb >= nb_all) { bb -= nb_all; ++cc; }
        }
    }
    struct Where { const W* s; W* d; int nb, nc; };
    auto where = [&](int64_t t) {
        const int64_t tc = t % tiles_c, tb = (t / tiles_c) % tiles_b, a = t / (tiles_c * tiles_b);
        const int64_t b0 = tb * TB, c0 = tc * kRelTile;
        Where w;
        w.nb = static_cast<int>(B - b0 < TB ? B - b0 : TB);
        w.nc = static_cast<int>(C - c0 < kRelTile ? C - c0 : kRelTile);
        w.s = src + (a * B + b0) * C + c0;
        w.d = dst + (a * C + c0) * B + b0;
        return w;
    };
    W regs[kPerLane];
    auto fetch = [&](const Where& w) {
        if (flat_r) {
            const int n = w.nb * w.nc;
#pragma unroll
            for (int k = 0; k < kPerLane; ++k) {
                const int i = tid + k * kBlock;
                regs[k] = i < n ? w.s[i] : W(0);
            }
        } else {
#pragma unroll
            for (int k = 0; k < kPerLane; ++k) {
                const int b = row0 + k * (kBlock / 64);
                regs[k] = (b < w.nb && lane < w.nc) ? w.s[static_cast<int64_t>(b) * C + lane] : W(0);
            }
        }
    };
    int64_t t = blockIdx.x;
    if (t >= tiles) return;
    Where cur = where(t);
    fetch(cur);
    for (;;) {
        // registers -> LDS
        if (flat_r) {
            const int n = cur.nb * cur.nc;
#pragma unroll
            for (int k = 0; k < kPerLane; ++k)
                if (tid + k * kBlock < n) tile[rq[k] >> 8][rq[k] & 255] = regs[k];      // (i < n <=> row < nb <= TB)
        } else {
#pragma unroll
            for (int k = 0; k < kPerLane; ++k) tile[row0 + k * (kBlock / 64)][lane] = regs[k];
        }
        __syncthreads();
        const int64_t tn = t + gridDim.x;
        const bool more = tn < tiles;
        Where nxt = cur;
        if (more) {
            nxt = where(tn);
            fetch(nxt);                                   // in flight while this tile is written out
        }
        if (flat_w) {
            const int n = cur.nb * cur.nc;
#pragma unroll
            for (int k = 0; k < kPerLane; ++k) {
                const int i = tid + k * kBlock;
                if (i < n) cur.d[i] = tile[wq[k] >> 8][wq[k] & 255];
            }
        } else {
#pragma unroll
            for (int k = 0; k < kRelTile / (kBlock / 64); ++k) {
                const int c = row0 + k * (kBlock / 64);
                if (c < cur.nc && lane < cur.nb) cur.d[static_cast<int64_t>(c) * B + lane] = tile[lane][c];
            }
        }
        if (!more) break;
        __syncthreads();                                  // the tile is read out: it may be overwritten
        t = tn;
        cur = nxt;
    }
}

template <typename W>
static hipError_t launch_relayout(const void* src, void* dst, int64_t A, int64_t B, int64_t C, hipStream_t stream) {
    constexpr int TB = kRelTile;
    const int64_t tiles_b = (B + TB - 1) / TB, tiles_c = (C + kRelTile - 1) / kRelTile;
    const int64_t tiles = A * tiles_b * tiles_c;
    if (tiles <= 0) return hipSuccess;
    const int64_t resident = static_cast<int64_t>(device_info().cu_count) * 8;      // 8 workgroups of 256 lanes per CU
    const dim3 grid(static_cast<unsigned>(tiles < resident ? tiles : resident));
    hipLaunchKernelGGL((relayout_kernel<W>), grid, dim3(kBlock), 0, stream, static_cast<const W*>(src), static_cast<W*>(dst), B, C,
                       tiles_b, tiles_c, tiles);
    return hipGetLastError();
}

hipError_t relayout(int elem_bytes, const void* src, void* dst, int64_t A, int64_t B, int64_t C, hipStream_t stream) {
    switch (elem_bytes) {
        case 2: return launch_relayout<uint16_t>(src, dst, A, B, C, stream);
        case 4: return launch_relayout<uint32_t>(src, dst, A, B, C, stream);
        case 8: return launch_relayout<uint64_t>(src, dst, A, B, C, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace lsq
