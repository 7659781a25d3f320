// tests/geom_driver.cpp -- the host-side launch geometry of the per-channel kernels (lsq_pc_geom.hpp: make_geom, make_geom_ww,
// plan_own / make_geom_own, make_seg_geom, pick_splits) over a few hundred thousand random shapes, compiled HOST-ONLY with
// UndefinedBehaviorSanitizer (tests/test_sanitizers_cpu.py): no division by zero, no overflow, and the invariants the kernels
// rely on.  No GPU is touched (device_info() falls back to 256 CUs when there is none).
#include <cstdio>
#include <cstdlib>

#include "lsq_pc_geom.hpp"

using namespace lsq;

static uint64_t st = 88172645463325252ull;
static uint64_t rnd() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; }
static int64_t pick(int64_t lo, int64_t hi) { return lo + static_cast<int64_t>(rnd() % static_cast<uint64_t>(hi - lo + 1)); }

#define REQUIRE(cond)                                                                                                    \
    do {                                                                                                                 \
        if (!(cond)) {                                                                                                   \
            std::printf("FAILED %s at outer=%lld C=%lld inner=%lld vec=%d\n", #cond, (long long)outer, (long long)C,     \
                        (long long)inner, vec);                                                                          \
            return 1;                                                                                                    \
        }                                                                                                                \
    } while (0)

int main() {
    const int cus = device_info().cu_count;
    long own_taken = 0;
    for (long it = 0; it < 300000; ++it) {
        const int vec = (int[]){1, 2, 4, 8}[rnd() % 4];
        int64_t outer = pick(1, (rnd() % 8 == 0) ? 100000 : 600), C = pick(1, (rnd() % 8 == 0) ? 70000 : 4100),
                inner = pick(1, (rnd() % 8 == 0) ? 200000 : 900);
        if (outer * C * inner > (int64_t{1} << 33)) continue;
        const int target = cus * static_cast<int>(pick(1, 16));
        const int resident = (rnd() % 2) ? cus * static_cast<int>(pick(1, 8)) : 0;
        if ((C * inner) % vec == 0) {
            const PcGeom g = make_geom(outer, C, inner, vec, target, 27, resident);
            REQUIRE(g.R >= 1 && g.splits >= 1 && g.n_windows >= 1 && g.k_slots >= 1);
            REQUIRE(g.n_tiles == (outer + g.R - 1) / g.R);
            REQUIRE(static_cast<int64_t>(g.splits) <= g.n_tiles);
            REQUIRE(g.n_windows * g.wpos >= g.L || g.R > 1);
            REQUIRE(g.k_slots <= C);
        }
        if (inner == 1 && vec > 1 && C % vec == 0) {
            for (int block : {256, 768, 1024}) {
                const PcGeom g = make_geom_ww(outer, C, vec, target, 16, resident, (rnd() % 2) != 0, block);
                REQUIRE(g.R >= 1 && g.ww_lanes >= 1 && g.block_threads >= 64 && g.block_threads <= 1024 && g.block_threads % 64 == 0);
                REQUIRE(static_cast<int64_t>(g.R) * g.ww_lanes <= g.block_threads);
                REQUIRE(g.splits >= 1 && static_cast<int64_t>(g.splits) <= g.n_tiles);
            }
        }
        for (int depth : {2, 4}) {
            const OwnPlan o = plan_own(outer, C, inner, vec, vec == 8 ? 2 : 4, depth, cus);
            if (o.k == 0) continue;
            ++own_taken;
            REQUIRE(vec > 1 && inner >= vec && C % o.k == 0 && (o.k * inner) % vec == 0);
            REQUIRE(o.lanes_per_row == o.k * inner / vec && o.lanes_per_row <= 256);
            REQUIRE(o.R >= 2 && static_cast<int64_t>(o.R) <= outer);
            REQUIRE(o.block_threads % 64 == 0 && o.block_threads <= 512 && o.R * o.lanes_per_row <= o.block_threads);
            REQUIRE(o.block_threads - o.R * o.lanes_per_row < 64);                    // stand-in lanes: less than one wave
            REQUIRE((outer + o.R - 1) / o.R >= 2 * depth);                            // the ring has its rows
            REQUIRE(C / o.k >= (3 * static_cast<int64_t>(cus)) / 4 && o.per_cu >= 1 && o.per_cu <= 4);
            const PcGeom g = make_geom_own(outer, C, inner, vec, o);
            REQUIRE(g.n_windows * o.k == C && g.wpos == o.k * inner && g.splits == 1 && g.own == o.lanes_per_row);
            // LDS: the table front + the ring of every wave, `per_cu` workgroups to a CU
            const size_t lds = bwd_lds_front_bytes(g, 16) + static_cast<size_t>(g.block_threads / 64) * depth * kDmaStageBytes;
            REQUIRE(lds * o.per_cu <= 160 * 1024);
        }
        if (vec > 1 && inner % vec == 0) {
            const SegGeom s = make_seg_geom(outer, C, inner, vec, target);
            REQUIRE(s.segs >= 1 && s.osplits >= 1 && s.sub_per_seg >= 1 && s.o_per_split >= 1);
            REQUIRE(static_cast<int64_t>(s.segs) * s.sub_per_seg >= s.n_sub && static_cast<int64_t>(s.osplits) * s.o_per_split >= outer);
        }
    }
    std::printf("ok (%ld owner plans checked, %d CUs assumed)\n", own_taken, cus);
    return 0;
}
