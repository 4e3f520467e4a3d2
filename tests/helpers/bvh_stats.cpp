// Host-only check + statistics of the BVH builders (host/accel.cpp): traverses the binary trees (Morton, SAH) and the
// four-wide tree the renderer uploads (CollapseToBvh4) like device/dscene.h:BvhIntersect (same slab test, same
// nearest-child-first order, same tie rule) on rays that resemble the
// renderer's (camera rays and cosine-distributed bounces off the hit points) and reports inner-node visits and triangle
// tests per ray.  Exit code 1 if the two trees disagree on any (triangle id, t).
// build: hipcc -x hip tests/helpers/bvh_stats.cpp <objs of the product> (see tests/test_host.py)
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../langevin-mcmc_amd/csrc/host/accel.h"
#include "../../langevin-mcmc_amd/csrc/host/scene.h"

using namespace lmcd;

struct Stats {
    long long rays = 0, nodes = 0, tris = 0, hits = 0, maxStack = 0, leaves = 0;
};

static bool Slab(const float *bmin, const float *bmax, V3 org, V3 invd, float tnear, float tfar, float &tEntry) {  // = dscene.h:SlabTest
    float ax = (bmin[0] - org.x) * invd.x, bx = (bmax[0] - org.x) * invd.x;
    float ay = (bmin[1] - org.y) * invd.y, by = (bmax[1] - org.y) * invd.y;
    float az = (bmin[2] - org.z) * invd.z, bz = (bmax[2] - org.z) * invd.z;
    float t0 = fmaxf(fmaxf(tnear, fminf(ax, bx)), fmaxf(fminf(ay, by), fminf(az, bz)));
    float t1 = fminf(fminf(tfar, fmaxf(ax, bx)), fminf(fmaxf(ay, by), fmaxf(az, bz)));
    tEntry = t0;
    return t0 * 0.9999996f <= t1 * 1.0000004f;
}

static int Traverse(const lmc::LbvhResult &B, V3 org, V3 dir, float tnear, float tfar, float &tHit, Stats &st) {
    V3 invd{1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z};
    int stack[BVH_STACK], sp = 0, best = -1, cur = 0;
    float bestT = tfar;
    bool alive = true;
    st.rays++;
    while (alive) {
        while (cur >= 0) {
            const BvhNode &nd = B.nodes[cur];
            st.nodes++;
            float tl, tr;
            const bool hl = Slab(nd.lmin, nd.lmax, org, invd, tnear, bestT, tl), hr = Slab(nd.rmin, nd.rmax, org, invd, tnear, bestT, tr);
            if (hl && hr) {
                int nearC = nd.left, farC = nd.right;
                if (tr < tl) nearC = nd.right, farC = nd.left;
                stack[sp++] = farC;
                if (sp > st.maxStack) st.maxStack = sp;
                cur = nearC;
            } else if (hl) cur = nd.left;
            else if (hr) cur = nd.right;
            else {
                if (!sp) {
                    alive = false;
                    break;
                }
                cur = stack[--sp];
            }
        }
        if (!alive) break;
        const unsigned code = (unsigned)~cur;
        const int first = (int)(code >> 3), cnt = (int)(code & 7u) + 1;
        st.leaves++;
        for (int i = 0; i < cnt; i++) {
            const LeafTri &tr = B.leafTris[first + i];
            st.tris++;
            float t;
            if (TriTest(tr.p0, tr.e1, tr.e2, org, dir, tnear, bestT, t))
                if (best < 0 || t < bestT || (t == bestT && tr.id < best)) bestT = t, best = tr.id;
        }
        if (!sp) break;
        cur = stack[--sp];
    }
    tHit = bestT;
    if (best >= 0) st.hits++;
    return best;
}

// device/dscene.h: VisitNode4<true> + BvhIntersect on the four-wide tree
static int Traverse4(const lmc::Bvh4Result &B, V3 org, V3 dir, float tnear, float tfar, float &tHit, Stats &st) {
    V3 invd{1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z};
    int stack[BVH_STACK], sp = 0, best = -1, cur = 0;
    float bestT = tfar;
    st.rays++;
    for (;;) {
        bool done = false;
        while (cur >= 0) {
            const BvhNode4 &nd = B.nodes[cur];
            st.nodes++;
            float tk[4];
            int ck[4];
            for (int k = 0; k < 4; k++) {
                float t;
                const bool h = nd.child[k] != BVH4_EMPTY && Slab(nd.bmin[k], nd.bmax[k], org, invd, tnear, bestT, t);
                tk[k] = h ? t : INFINITY, ck[k] = h ? nd.child[k] : BVH4_EMPTY;
            }
            auto cswap = [&](int a, int b) {
                if (tk[b] < tk[a]) std::swap(tk[a], tk[b]), std::swap(ck[a], ck[b]);
            };
            cswap(0, 1), cswap(2, 3), cswap(0, 2), cswap(1, 3), cswap(1, 2);
            for (int k = 3; k >= 1; k--)
                if (ck[k] != BVH4_EMPTY) stack[sp++] = ck[k];
            if (sp > st.maxStack) st.maxStack = sp;
            cur = ck[0];
            if (cur == BVH4_EMPTY) {
                if (!sp) {
                    done = true;
                    break;
                }
                cur = stack[--sp];
            }
        }
        if (done) break;
        const unsigned code = (unsigned)~cur;
        const int first = (int)(code >> 3), cnt = (int)(code & 7u) + 1;
        st.leaves++;
        for (int i = 0; i < cnt; i++) {
            const LeafTri &tr = B.leafTris[first + i];
            st.tris++;
            float t;
            if (TriTest(tr.p0, tr.e1, tr.e2, org, dir, tnear, bestT, t))
                if (best < 0 || t < bestT || (t == bestT && tr.id < best)) bestT = t, best = tr.id;
        }
        if (!sp) break;
        cur = stack[--sp];
    }
    tHit = bestT;
    if (best >= 0) st.hits++;
    return best;
}

// device/dscene.h: VisitNode4Q<true> on the quantised nodes (build option LMC_BVH_QUANT): same arithmetic (fused multiply-add of the 8-bit
// offsets), same widened comparison; the boxes only cull, so the hits must be those of the exact nodes
static int Traverse4Q(const lmc::Bvh4Result &B, V3 org, V3 dir, float tnear, float tfar, float &tHit, Stats &st) {
    V3 invd{1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z};
    int stack[BVH_STACK], sp = 0, best = -1, cur = 0;
    float bestT = tfar;
    st.rays++;
    for (;;) {
        bool done = false;
        while (cur >= 0) {
            const BvhNode4Q &nd = B.qnodes[cur];
            st.nodes++;
            const float A[3] = {(nd.org[0] - org.x) * invd.x, (nd.org[1] - org.y) * invd.y, (nd.org[2] - org.z) * invd.z};
            const float Bq[3] = {nd.scale[0] * invd.x, nd.scale[1] * invd.y, nd.scale[2] * invd.z};
            float tk[4];
            int ck[4];
            for (int k = 0; k < 4; k++) {
                const float ax = fmaf((float)nd.qmin[0][k], Bq[0], A[0]), bx = fmaf((float)nd.qmax[0][k], Bq[0], A[0]);
                const float ay = fmaf((float)nd.qmin[1][k], Bq[1], A[1]), by = fmaf((float)nd.qmax[1][k], Bq[1], A[1]);
                const float az = fmaf((float)nd.qmin[2][k], Bq[2], A[2]), bz = fmaf((float)nd.qmax[2][k], Bq[2], A[2]);
                const float t0 = fmaxf(fmaxf(tnear, fminf(ax, bx)), fmaxf(fminf(ay, by), fminf(az, bz)));
                const float t1 = fminf(fminf(bestT, fmaxf(ax, bx)), fminf(fmaxf(ay, by), fmaxf(az, bz)));
                const bool h = nd.child[k] != BVH4_EMPTY && t0 * 0.9999992f <= t1 * 1.0000008f;
                tk[k] = h ? t0 : INFINITY, ck[k] = h ? nd.child[k] : BVH4_EMPTY;
            }
            auto cswap = [&](int a, int b) {
                if (tk[b] < tk[a]) std::swap(tk[a], tk[b]), std::swap(ck[a], ck[b]);
            };
            cswap(0, 1), cswap(2, 3), cswap(0, 2), cswap(1, 3), cswap(1, 2);
            for (int k = 3; k >= 1; k--)
                if (ck[k] != BVH4_EMPTY) stack[sp++] = ck[k];
            if (sp > st.maxStack) st.maxStack = sp;
            cur = ck[0];
            if (cur == BVH4_EMPTY) {
                if (!sp) {
                    done = true;
                    break;
                }
                cur = stack[--sp];
            }
        }
        if (done) break;
        const unsigned code = (unsigned)~cur;
        const int first = (int)(code >> 3), cnt = (int)(code & 7u) + 1;
        st.leaves++;
        for (int i = 0; i < cnt; i++) {
            const LeafTri &tr = B.leafTris[first + i];
            st.tris++;
            float t;
            if (TriTest(tr.p0, tr.e1, tr.e2, org, dir, tnear, bestT, t))
                if (best < 0 || t < bestT || (t == bestT && tr.id < best)) bestT = t, best = tr.id;
        }
        if (!sp) break;
        cur = stack[--sp];
    }
    tHit = bestT;
    if (best >= 0) st.hits++;
    return best;
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    lmc::LoadOverrides ov;
    auto scene = lmc::ParseScene(argv[1], ov);
    const int nRays = argc > 2 ? atoi(argv[2]) : 100000;
    std::vector<TriData> tris;
    for (size_t mi = 0; mi < scene->meshes.size(); mi++) {
        const lmc::Mesh &m = scene->meshes[mi];
        for (size_t t = 0; t < m.numTris(); t++) {
            TriData T;
            memset(&T, 0, sizeof(T));
            uint32_t i0 = m.idx[3 * t], i1 = m.idx[3 * t + 1], i2 = m.idx[3 * t + 2];
            for (int k = 0; k < 3; k++) T.p0[k] = m.P[i0][k], T.e1[k] = m.P[i1][k] - m.P[i0][k], T.e2[k] = m.P[i2][k] - m.P[i0][k];
            tris.push_back(T);
        }
    }
    lmc::LbvhResult trees[2] = {lmc::BuildLbvh(tris), lmc::BuildSahBvh(tris, 4)};
    const lmc::Bvh4Result wide = lmc::CollapseToBvh4(trees[1]);
    const char *names[4] = {"lbvh", "sah", "sah_4wide", "sah_4wide_quantised"};
    Stats st[4];
    std::mt19937 gen(7);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    const lmc::Camera &cam = scene->camera;
    lmc::M4 tw = lmc::ToM4(cam.camToWorld);
    long long mismatches = 0;
    for (int r = 0; r < nRays; r++) {
        // camera ray through a random film point (camera.cpp:38-51), then up to 5 cosine bounces
        float sx = U(gen), sy = U(gen);
        const auto &m = cam.sampleToCam.m;
        float px = m[0][0] * sx + m[0][1] * sy + m[0][3], py = m[1][0] * sx + m[1][1] * sy + m[1][3], pz = m[2][0] * sx + m[2][1] * sy + m[2][3],
              pw = m[3][0] * sx + m[3][1] * sy + m[3][3];
        V3 dc{px / pw, py / pw, pz / pw};
        float len = sqrtf(dc.x * dc.x + dc.y * dc.y + dc.z * dc.z);
        dc = V3{dc.x / len, dc.y / len, dc.z / len};
        V3 org{tw.m[0][3], tw.m[1][3], tw.m[2][3]};
        V3 dir{tw.m[0][0] * dc.x + tw.m[0][1] * dc.y + tw.m[0][2] * dc.z, tw.m[1][0] * dc.x + tw.m[1][1] * dc.y + tw.m[1][2] * dc.z,
               tw.m[2][0] * dc.x + tw.m[2][1] * dc.y + tw.m[2][2] * dc.z};
        float tnear = 1e-3f;
        for (int bounce = 0; bounce < 6; bounce++) {
            float t[2];
            int id[2];
            for (int k = 0; k < 2; k++) id[k] = Traverse(trees[k], org, dir, tnear, INFINITY, t[k], st[k]);
            if (id[0] != id[1] || (id[0] >= 0 && t[0] != t[1])) mismatches++;
            float t4;
            const int id4 = Traverse4(wide, org, dir, tnear, INFINITY, t4, st[2]);
            if (id4 != id[1] || (id4 >= 0 && t4 != t[1])) mismatches++;
            float tq;
            const int idq = Traverse4Q(wide, org, dir, tnear, INFINITY, tq, st[3]);
            if (idq != id[1] || (idq >= 0 && tq != t[1])) mismatches++;
            if (id[0] < 0) break;
            const TriData &T = tris[id[0]];
            V3 e1{T.e1[0], T.e1[1], T.e1[2]}, e2{T.e2[0], T.e2[1], T.e2[2]};
            V3 n = Normalize(Cross(e1, e2));
            if (Dot(n, dir) > 0) n = -n;
            org = org + t[0] * dir;
            V3 b0, b1;
            CoordinateSystem(n, b0, b1);
            float u1 = U(gen), u2 = U(gen), rr = sqrtf(u1), phi = 6.2831853f * u2;
            float lx = rr * cosf(phi), ly = rr * sinf(phi), lz = sqrtf(fmaxf(0.f, 1.f - u1));
            dir = lx * b0 + ly * b1 + lz * n;
            tnear = 5e-4f;
        }
    }
    for (int k = 0; k < 4; k++)
        printf("{\"tree\": \"%s\", \"nodes\": %zu, \"depth\": %d, \"rays\": %lld, \"node_visits_per_ray\": %.2f, \"leaf_visits_per_ray\": %.2f, \"tri_tests_per_ray\": %.2f, \"max_stack\": %lld, \"stack_bound\": %d}\n",
               names[k], k < 2 ? trees[k].nodes.size() : wide.nodes.size(), k < 2 ? trees[k].depth : wide.depth, st[k].rays, (double)st[k].nodes / st[k].rays,
               (double)st[k].leaves / st[k].rays, (double)st[k].tris / st[k].rays, st[k].maxStack, k < 2 ? trees[k].depth : wide.stackNeed);
    printf("{\"thickened_flat_leaf_share\": %.4f}\n", lmc::ThickenedFlatLeafShare(wide));  // accel.h: what tells scenes the quantised nodes suit from the others
    if (st[2].maxStack > wide.stackNeed || st[3].maxStack > wide.stackNeed) mismatches++;  // the bound the host sizes the traversal stack with must hold
    printf("{\"mismatches\": %lld}\n", mismatches);
    return mismatches ? 1 : 0;
}
