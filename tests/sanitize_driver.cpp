// tests/sanitize_driver.cpp -- drives every entry point of include/lsq_cpu.h over awkward sizes; built and run by
// tests/test_sanitizers_cpu.py with -fsanitize=address,undefined (heap / stack overruns, misaligned or out-of-range loads,
// signed overflow, invalid float -> int casts).  Exit code 0 = clean.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "lsq_cpu.h"

namespace {
uint64_t state = 0x9E3779B97F4A7C15ull;
double rnd() {
    state ^= state << 13; state ^= state >> 7; state ^= state << 17;
    return static_cast<double>(state >> 11) / static_cast<double>(1ull << 53) * 4.0 - 1.5;
}
uint16_t to_bf16(float f) { uint32_t b; std::memcpy(&b, &f, 4); b += 0x7fffu + ((b >> 16) & 1u); return static_cast<uint16_t>(b >> 16); }

template <typename E>
void fill(std::vector<E>& v, int dtype) {
    for (auto& e : v) {
        const double d = rnd();
        if (dtype == LSQ_BF16) e = static_cast<E>(to_bf16(static_cast<float>(d)));
        else e = static_cast<E>(d);
    }
}

template <typename E, typename P>
int run(int dtype) {
    int bad = 0;
    const int64_t shapes[][3] = {{1, 1, 1}, {1, 1, 7}, {3, 5, 1}, {2, 3, 49}, {1, 8, 255}, {17, 2, 33}, {4, 16, 64}, {1, 1, 100003}};
    for (const auto& s : shapes) {
        const int64_t outer = s[0], C = s[1], inner = s[2], n = outer * C * inner;
        std::vector<E> x(n), g(n), y(n), dx(n);
        fill(x, dtype); fill(g, dtype);
        std::vector<P> scale(C), shift(C), ds(C), db(C);
        for (int64_t c = 0; c < C; ++c) { scale[c] = static_cast<P>(0.01 + 0.05 * (c % 7)); shift[c] = static_cast<P>(0.1 * ((c % 5) - 2)); }
        if (C > 1) scale[1] = static_cast<P>(-0.02);         // a negative scale: |scale| is used
        std::vector<double> wide(2 * C + 1);
        for (int mode = 0; mode < 8; ++mode) {
            lsq_params p{};
            p.quant_min = (mode & 1) ? -8 : 0; p.quant_max = (mode & 1) ? 7 : 127; p.type_min = (mode & 1) ? -128 : 0; p.type_max = (mode & 1) ? 127 : 255;
            p.use_grad_scaling = 1; p.grad_scaler = 1.0; p.sym = (mode >> 1) & 1; p.init_mode = (mode >> 2) & 1; p.eval_mode = 0; p.numel_for_scaler = 0;
            bad |= lsq_cpu_forward_per_channel(dtype, x.data(), y.data(), outer, C, inner, scale.data(), shift.data(), &p);
            bad |= lsq_cpu_backward_per_channel(dtype, g.data(), x.data(), dx.data(), ds.data(), db.data(), wide.data(), outer, C, inner,
                                                scale.data(), shift.data(), &p);
            wide[2 * C] = static_cast<double>(n);
            bad |= lsq_cpu_sharded_finish(dtype, wide.data(), C, 1, &p, ds.data(), db.data());
            bad |= lsq_cpu_forward_per_tensor(dtype, x.data(), y.data(), n, scale.data(), shift.data(), &p);
            double w2[3] = {0, 0, static_cast<double>(n)};
            bad |= lsq_cpu_backward_per_tensor(dtype, g.data(), x.data(), dx.data(), ds.data(), db.data(), w2, n, scale.data(), shift.data(), &p);
            bad |= lsq_cpu_sharded_finish(dtype, w2, 1, 0, &p, ds.data(), db.data());
            p.eval_mode = 1;
            bad |= lsq_cpu_backward_per_tensor(dtype, g.data(), x.data(), dx.data(), ds.data(), db.data(), nullptr, n, scale.data(), shift.data(), &p);
        }
    }
    return bad;
}
}  // namespace

int main() {
    int bad = 0;
    for (int threads : {1, 3, 8}) {
        lsq_cpu_set_num_threads(threads);
        bad |= run<float, float>(LSQ_F32);
        bad |= run<double, double>(LSQ_F64);
        bad |= run<uint16_t, float>(LSQ_BF16);
    }
    // rejected arguments must be reported, not dereferenced
    lsq_params p{};
    p.quant_max = 127; p.type_max = 255;
    float one = 1.f;
    if (lsq_cpu_forward_per_tensor(LSQ_F16, &one, &one, 1, &one, &one, &p) == 0) bad = 1;
    if (lsq_cpu_forward_per_tensor(LSQ_F32, nullptr, &one, 1, &one, &one, &p) == 0) bad = 1;
    if (lsq_cpu_backward_per_channel(LSQ_F32, &one, &one, &one, &one, &one, nullptr, 0, 1, 1, &one, &one, &p) == 0) bad = 1;
    std::printf("%s (abi %d)\n", bad ? "FAILED" : "ok", lsq_cpu_abi_version());
    return bad ? 1 : 0;
}
