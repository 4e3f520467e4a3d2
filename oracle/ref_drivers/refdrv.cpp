// Reference-side known-answer drivers.  TEST INFRASTRUCTURE ONLY.
// Compiles the reference's own headers from where they lie (-I/root/reference/src,
// -I/root/reference/nanoflann/include) into oracle/_ref/librefdrv.so; nothing is copied.
// Used to pin the oracle restatement of
//   RNG = pcg32_k64_fast                      (src/commondef.h:63, src/pcg_random.hpp:1692)
//   libstdc++ uniform/normal distributions    (src/mlt.cpp:63, src/gaussian.cpp:44)
//   fastlog                                   (src/fastmath.h:364-381)
//   nanoflann radiusSearch w/ knn early-out   (src/global_cache.h:96-124, nanoflann.hpp:225-262,1291)
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>
#include "pcg_random.hpp"
#include "fastmath.h"
#include <nanoflann.hpp>

typedef pcg32_k64_fast RNG;

// nanoflann (reference-modified: RadiusResultSet stops after knn matches)
struct Cloud {
    const float *pts; int n; int dim;
    inline size_t kdtree_get_point_count() const { return n; }
    inline float kdtree_get_pt(const size_t idx, const size_t d) const { return pts[idx * dim + d]; }
    template <class BBOX> bool kdtree_get_bbox(BBOX &) const { return false; }
};

template <int DIM>
static void kd_query_t(int npts, const float *pts, int nq, const float *q, float radius_sq, int knn,
                       int *out_n, int *out_idx, float *out_dist) {
    using namespace nanoflann;
    Cloud cloud{pts, npts, DIM};
    typedef KDTreeSingleIndexAdaptor<L2_Simple_Adaptor<float, Cloud>, Cloud, DIM> KDTree;
    KDTree tree(DIM, cloud, KDTreeSingleIndexAdaptorParams(10));
    tree.buildIndex();
    for (int i = 0; i < nq; i++) {
        std::vector<std::pair<size_t, float>> m;
        SearchParams params;
        size_t nm = tree.radiusSearch(q + (size_t)i * DIM, radius_sq, m, params, knn);
        out_n[i] = (int)nm;
        for (int k = 0; k < knn; k++) {
            out_idx[i * knn + k] = k < (int)nm ? (int)m[k].first : -1;
            out_dist[i * knn + k] = k < (int)nm ? m[k].second : 0.f;
        }
    }
}


extern "C" {

void ref_pcg_u32(uint64_t seed, int n, uint32_t *out) {
    RNG rng(seed);
    for (int i = 0; i < n; i++) out[i] = rng();
}

void ref_pcg_uniform(uint64_t seed, int n, float *out) {
    RNG rng(seed);
    std::uniform_real_distribution<float> uni(0.f, 1.f);
    for (int i = 0; i < n; i++) out[i] = uni(rng);
}

// one normal_distribution object for all n draws (saved second variate is kept)
void ref_pcg_normal(uint64_t seed, int n, float mean, float stddev, float *out) {
    RNG rng(seed);
    std::normal_distribution<float> nd(mean, stddev);
    for (int i = 0; i < n; i++) out[i] = nd(rng);
}

// mixed stream as in one small step: u, u, then a fresh normal object with k draws, repeated
void ref_pcg_mixed(uint64_t seed, int rounds, int k, float *out) {
    RNG rng(seed);
    std::uniform_real_distribution<float> uni(0.f, 1.f);
    int o = 0;
    for (int r = 0; r < rounds; r++) {
        out[o++] = uni(rng);
        out[o++] = uni(rng);
        std::normal_distribution<float> nd(0.f, 1.f);
        for (int i = 0; i < k; i++) out[o++] = nd(rng);
    }
}

// sizeof + raw state dump: [state(u64 as 2xu32 lo,hi), table 64 x u32]
int ref_pcg_sizeof() { return (int)sizeof(RNG); }
void ref_pcg_dump(uint64_t seed, int ndraws, uint32_t *out66) {
    RNG rng(seed);
    for (int i = 0; i < ndraws; i++) rng();
    // layout verified by ref_pcg_sizeof()==264: base engine state_ (u64) first, then data_[64]
    memcpy(out66, &rng, 264);
}

void ref_fastlog(int n, const float *in, float *out) {
    for (int i = 0; i < n; i++) out[i] = fastlog(in[i]);
}
void ref_fastpow(int n, const float *x, const float *p, float *out) {
    for (int i = 0; i < n; i++) out[i] = fastpow(x[i], p[i]);
}

int ref_kd_query(int dim, int npts, const float *pts, int nq, const float *q, float radius_sq, int knn,
                 int *out_n, int *out_idx, float *out_dist) {
    switch (dim) {
#define C(D) case D: kd_query_t<D>(npts, pts, nq, q, radius_sq, knn, out_n, out_idx, out_dist); return 0;
        C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16)
#undef C
    }
    return -1;
}

}  // extern "C"
