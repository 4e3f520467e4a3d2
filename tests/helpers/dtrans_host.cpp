// TEST HELPER: the deterministic float exp / log / pow (langevin-mcmc_amd/csrc/device/dtrans.h), sin / cos / acos / atan2 (dtrig.h) and the restated
// glibc logf of the normal distribution (drng.h GlibcLogf) of the product compiled by the host compiler with the product's arithmetic contract
// (-ffp-contract=off), for the accuracy tests and as the CPU side of the GPU bit-equality tests.
// modes: 0 exp, 1 log, 2 pow, 3 sin, 4 cos, 5 acos, 6 atan2(x, y), 7 GlibcLogf, 8 the HOST libm's logf (what libstdc++'s normal_distribution calls)
#include "../../langevin-mcmc_amd/csrc/device/drng.h"
#include <cstring>
#include <thread>
#include <vector>
static float One(int mode, float x, float y) {
    switch (mode) {
        case 0: return lmcd::lexpf(x);
        case 1: return lmcd::llogf(x);
        case 2: return lmcd::lpowf(x, y);
        case 3: return lmcd::dsinf(x);
        case 4: return lmcd::dcosf(x);
        case 5: return lmcd::dacosf(x);
        case 6: return lmcd::datan2f(x, y);
        case 7: return lmcd::GlibcLogf(x);
        default: return logf(x);
    }
}
extern "C" void lmc_test_trans_host(int n, int mode, const float *x, const float *y, float *o) {
    for (int i = 0; i < n; i++) o[i] = One(mode, x[i], y[i]);
}
// GlibcLogf against the host's logf on EVERY float with bit pattern in [lo_bits, hi_bits]: number of arguments whose results differ in any bit
extern "C" unsigned long long lmc_test_logf_exhaustive(unsigned lo_bits, unsigned hi_bits, int threads, float *first_bad) {
    std::vector<unsigned long long> bad(threads, 0);
    std::vector<float> at(threads, 0.f);
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++)
        th.emplace_back([&, t]() {
            const unsigned long long n = (unsigned long long)hi_bits - lo_bits + 1, a = n * t / threads, b = n * (t + 1) / threads;
            for (unsigned long long q = a; q < b; q++) {
                const uint32_t u = lo_bits + (uint32_t)q;
                float x;
                memcpy(&x, &u, 4);
                const float g = logf(x), m = lmcd::GlibcLogf(x);
                if (memcmp(&g, &m, 4)) {
                    if (!bad[t]) at[t] = x;
                    bad[t]++;
                }
            }
        });
    for (auto &x : th) x.join();
    unsigned long long total = 0;
    for (int t = threads - 1; t >= 0; t--) {
        total += bad[t];
        if (bad[t] && first_bad) *first_bad = at[t];
    }
    return total;
}
