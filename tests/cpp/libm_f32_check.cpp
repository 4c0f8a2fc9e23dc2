// tests/cpp/libm_f32_check.cpp -- compares an implementation of atanf / atan2f with THIS machine's libm, bit for bit:
// atanf on all 2^32 arguments, atan2f on `pairs` pairs (random bit patterns, lidar-like coordinates, ratios on the reduction
// thresholds, special values).  Compiled twice by the tests: -DIMPL_HEADER="libm_f32.h" with -I oracle (IMPL_NS mmlo_libm,
// atanf_fdlibm / atan2f_fdlibm) and with -I multi-modal-loam_amd/csrc (IMPL_NS mml_libm, atanf_fd / atan2f_fd).
// usage: libm_f32_check <pairs> <threads>;  prints "atanf <mismatches>" and "atan2f <mismatches>"; exit code 1 on any.
#include IMPL_HEADER
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

static inline int32_t bits(float f) {
    int32_t i;
    std::memcpy(&i, &f, 4);
    return i;
}
static inline float from_bits(uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
static inline bool same(float a, float b) { return (a != a && b != b) || bits(a) == bits(b); }

int main(int argc, char** argv) {
    const long pairs = argc > 1 ? atol(argv[1]) : 100000000L;
    const int T = argc > 2 ? atoi(argv[2]) : 4;
    std::atomic<long> bad1{0}, bad2{0};
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
        th.emplace_back([&, t] {
            long b = 0;
            for (uint64_t u = t; u < (1ull << 32); u += T) {
                const float x = from_bits((uint32_t)u);
                const float want = ::atanf(x), got = IMPL_NS::IMPL_ATAN(x);
                if (!same(want, got) && b++ < 5) std::printf("atanf(%08x): libm %08x, restated %08x\n", (unsigned)u, bits(want), bits(got));
            }
            bad1 += b;
        });
    for (auto& x : th) x.join();
    th.clear();
    for (int t = 0; t < T; ++t)
        th.emplace_back([&, t] {
            std::mt19937_64 g(1234 + t);
            std::uniform_real_distribution<float> lidar(-200.f, 200.f);
            static const float sp[] = {0.f, -0.f, 1.f, -1.f, INFINITY, -INFINITY, 1e-45f, -1e-45f, 1e-38f, 3e38f, -3e38f, 0.4375f, 0.6875f, 1.1875f, 2.4375f, 33554432.f, 1.1754944e-38f};
            long b = 0;
            for (long i = 0; i < pairs / T; ++i) {
                const uint64_t r = g();
                float x, y;
                const int mode = (int)(i & 7);
                if (mode < 3) {  // random bit patterns
                    x = from_bits((uint32_t)r);
                    y = from_bits((uint32_t)(r >> 32));
                } else if (mode < 6) {  // lidar-like magnitudes
                    x = lidar(g);
                    y = lidar(g);
                } else if (mode == 6) {  // |y| within a few ulps of |x| * threshold
                    static const float thr[] = {1.f, 0.4375f, 0.6875f, 1.1875f, 2.4375f};
                    x = lidar(g);
                    const float yy = x * thr[(r >> 48) % 5];
                    y = from_bits((uint32_t)(bits(yy) + (int)((r >> 32) & 15) - 8) ^ (((r >> 40) & 1) ? 0x80000000u : 0u));
                } else {  // special values against each other and against random patterns
                    x = sp[r % 17];
                    y = ((r >> 20) & 1) ? from_bits((uint32_t)(r >> 32)) : sp[(r >> 8) % 17];
                    if ((r >> 21) & 1) std::swap(x, y);
                }
                const float want = ::atan2f(y, x), got = IMPL_NS::IMPL_ATAN2(y, x);
                if (!same(want, got) && b++ < 5) std::printf("atan2f(%08x, %08x): libm %08x, restated %08x\n", bits(y), bits(x), bits(want), bits(got));
            }
            bad2 += b;
        });
    for (auto& x : th) x.join();
    std::printf("atanf %ld\natan2f %ld\n", (long)bad1, (long)bad2);
    return (bad1 || bad2) ? 1 : 0;
}
