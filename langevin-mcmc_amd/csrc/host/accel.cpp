// See accel.h.
#include "accel.h"

#include <cstdlib>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>

namespace lmc {

// ------------------------------------------------------------------------------------------------ LBVH
static inline uint64_t ExpandBits21(uint32_t v) {  // 21 bits -> every third bit of 63
    uint64_t x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

namespace {
struct Prim {
    uint64_t code;
    int id;
    float bmin[3], bmax[3];
};
struct Builder {
    std::vector<Prim> prims;
    LbvhResult out;
    const std::vector<lmcd::TriData> *tris;
    int maxLeaf = 4;  // triangles per leaf (<= 8 by the leaf encoding); LMC_BVH_LEAF overrides for experiments

    // returns encoded child (>= 0 inner node index, < 0 leaf) and its box
    int Build(int lo, int hi, float *bmin, float *bmax, int depth) {
        out.depth = std::max(out.depth, depth);
        const int n = hi - lo;
        if (n <= maxLeaf) {
            int first = (int)out.leafTris.size();
            for (int k = 0; k < 3; k++) bmin[k] = INFINITY, bmax[k] = -INFINITY;
            for (int i = lo; i < hi; i++) {
                const lmcd::TriData &T = (*tris)[prims[i].id];
                lmcd::LeafTri lt;
                memset(&lt, 0, sizeof(lt));
                memcpy(lt.p0, T.p0, 12), memcpy(lt.e1, T.e1, 12), memcpy(lt.e2, T.e2, 12);
                lt.id = prims[i].id;
                out.leafTris.push_back(lt);
                for (int k = 0; k < 3; k++) bmin[k] = std::min(bmin[k], prims[i].bmin[k]), bmax[k] = std::max(bmax[k], prims[i].bmax[k]);
            }
            return ~((first << 3) | (n - 1));
        }
        // split at the highest bit in which the first and last Morton code differ (binary radix tree)
        uint64_t a = prims[lo].code, b = prims[hi - 1].code;
        int mid;
        if (a == b) {
            mid = (lo + hi) / 2;
        } else {
            int bit = 63 - __builtin_clzll(a ^ b);
            uint64_t mask = 1ull << bit;
            // first index whose code has `bit` set (codes are sorted, prefix above `bit` is common)
            int l = lo, r = hi - 1;
            while (l < r) {
                int m = (l + r) / 2;
                if (prims[m].code & mask) r = m;
                else l = m + 1;
            }
            mid = l;
        }
        int ni = (int)out.nodes.size();
        out.nodes.push_back(lmcd::BvhNode());
        float lmin[3], lmax[3], rmin[3], rmax[3];
        int left = Build(lo, mid, lmin, lmax, depth + 1);
        int right = Build(mid, hi, rmin, rmax, depth + 1);
        lmcd::BvhNode &nd = out.nodes[ni];
        memset(&nd, 0, sizeof(nd));
        for (int k = 0; k < 3; k++) {
            nd.lmin[k] = lmin[k], nd.lmax[k] = lmax[k], nd.rmin[k] = rmin[k], nd.rmax[k] = rmax[k];
            bmin[k] = std::min(lmin[k], rmin[k]), bmax[k] = std::max(lmax[k], rmax[k]);
        }
        nd.left = left, nd.right = right;
        return ni;
    }
};
}  // namespace

// Top-down binned-SAH build over the same leaf / node format (the scene is static and built once on the host, so build
// quality is free): 16 centroid bins per axis, leaf when <= maxLeaf triangles and splitting does not pay, median split of
// the widest axis when every centroid falls into one bin.  Closest hits are tree independent (ties -> lowest triangle
// id), so the two builders are interchangeable; tests/test_host.py checks that on random rays.
namespace {
struct SahBuilder {
    std::vector<Prim> prims;
    std::vector<float> cen;  // 3 per prim (by position in prims)
    LbvhResult out;
    const std::vector<lmcd::TriData> *tris;
    int maxLeaf = 4;

    static float Area(const float *mn, const float *mx) {
        float d[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
        if (d[0] < 0 || d[1] < 0 || d[2] < 0) return 0.f;
        return 2.f * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]);
    }
    int MakeLeaf(int lo, int hi, float *bmin, float *bmax) {
        int first = (int)out.leafTris.size();
        for (int k = 0; k < 3; k++) bmin[k] = INFINITY, bmax[k] = -INFINITY;
        std::sort(prims.begin() + lo, prims.begin() + hi, [](const Prim &a, const Prim &b) { return a.id < b.id; });
        for (int i = lo; i < hi; i++) {
            const lmcd::TriData &T = (*tris)[prims[i].id];
            lmcd::LeafTri lt;
            memset(&lt, 0, sizeof(lt));
            memcpy(lt.p0, T.p0, 12), memcpy(lt.e1, T.e1, 12), memcpy(lt.e2, T.e2, 12);
            lt.id = prims[i].id;
            out.leafTris.push_back(lt);
            for (int k = 0; k < 3; k++) bmin[k] = std::min(bmin[k], prims[i].bmin[k]), bmax[k] = std::max(bmax[k], prims[i].bmax[k]);
        }
        return ~((first << 3) | (hi - lo - 1));
    }
    static float Centroid(const Prim &p, int k) { return 0.5f * (p.bmin[k] + p.bmax[k]); }
    int Build(int lo, int hi, float *bmin, float *bmax, int depth) {
        out.depth = std::max(out.depth, depth);
        const int n = hi - lo;
        if (n <= 1) return MakeLeaf(lo, hi, bmin, bmax);
        float bb[2][3] = {{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}}, cb[2][3] = {{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}};
        for (int i = lo; i < hi; i++)
            for (int k = 0; k < 3; k++) {
                bb[0][k] = std::min(bb[0][k], prims[i].bmin[k]), bb[1][k] = std::max(bb[1][k], prims[i].bmax[k]);
                const float c = Centroid(prims[i], k);
                cb[0][k] = std::min(cb[0][k], c), cb[1][k] = std::max(cb[1][k], c);
            }
        constexpr int NB = 16;
        float bestCost = INFINITY;
        int bestAxis = -1, bestBin = -1;
        for (int ax = 0; ax < 3; ax++) {
            const float ext = cb[1][ax] - cb[0][ax];
            if (!(ext > 0)) continue;
            int cnt[NB] = {0};
            float bmn[NB][3], bmx[NB][3];
            for (int b = 0; b < NB; b++)
                for (int k = 0; k < 3; k++) bmn[b][k] = INFINITY, bmx[b][k] = -INFINITY;
            for (int i = lo; i < hi; i++) {
                int b = std::min(NB - 1, (int)((Centroid(prims[i], ax) - cb[0][ax]) / ext * NB));
                cnt[b]++;
                for (int k = 0; k < 3; k++) bmn[b][k] = std::min(bmn[b][k], prims[i].bmin[k]), bmx[b][k] = std::max(bmx[b][k], prims[i].bmax[k]);
            }
            float rArea[NB];
            int rCnt[NB];
            float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
            int c = 0;
            for (int b = NB - 1; b > 0; b--) {
                for (int k = 0; k < 3; k++) mn[k] = std::min(mn[k], bmn[b][k]), mx[k] = std::max(mx[k], bmx[b][k]);
                c += cnt[b];
                rArea[b] = Area(mn, mx), rCnt[b] = c;
            }
            for (int k = 0; k < 3; k++) mn[k] = INFINITY, mx[k] = -INFINITY;
            c = 0;
            for (int b = 0; b < NB - 1; b++) {
                for (int k = 0; k < 3; k++) mn[k] = std::min(mn[k], bmn[b][k]), mx[k] = std::max(mx[k], bmx[b][k]);
                c += cnt[b];
                if (c == 0 || rCnt[b + 1] == 0) continue;
                // cost in units of one triangle test; leaves hold up to maxLeaf triangles fetched together
                const float cost = Area(mn, mx) * c + rArea[b + 1] * rCnt[b + 1];
                if (cost < bestCost) bestCost = cost, bestAxis = ax, bestBin = b;
            }
        }
        const float parentArea = Area(bb[0], bb[1]);
        if (n <= maxLeaf) {
            // a node visit costs about as much as two triangle tests (one 64 B fetch + two slab tests)
            const float leafCost = (float)n, splitCost = bestAxis >= 0 && parentArea > 0 ? 2.0f + bestCost / parentArea : INFINITY;
            if (!(splitCost < leafCost)) return MakeLeaf(lo, hi, bmin, bmax);
        }
        int mid;
        if (bestAxis < 0) {
            mid = (lo + hi) / 2;  // all centroids coincide
        } else {
            const float ext = cb[1][bestAxis] - cb[0][bestAxis];
            auto it = std::partition(prims.begin() + lo, prims.begin() + hi, [&](const Prim &p) {
                int b = std::min(NB - 1, (int)((Centroid(p, bestAxis) - cb[0][bestAxis]) / ext * NB));
                return b <= bestBin;
            });
            mid = (int)(it - prims.begin());
            if (mid == lo || mid == hi) mid = (lo + hi) / 2;
        }
        int ni = (int)out.nodes.size();
        out.nodes.push_back(lmcd::BvhNode());
        float lmin[3], lmax[3], rmin[3], rmax[3];
        int left = Build(lo, mid, lmin, lmax, depth + 1);
        int right = Build(mid, hi, rmin, rmax, depth + 1);
        lmcd::BvhNode &nd = out.nodes[ni];
        memset(&nd, 0, sizeof(nd));
        for (int k = 0; k < 3; k++) {
            nd.lmin[k] = lmin[k], nd.lmax[k] = lmax[k], nd.rmin[k] = rmin[k], nd.rmax[k] = rmax[k];
            bmin[k] = std::min(lmin[k], rmin[k]), bmax[k] = std::max(lmax[k], rmax[k]);
        }
        nd.left = left, nd.right = right;
        return ni;
    }
};
}  // namespace

static void PrimBounds(const std::vector<lmcd::TriData> &tris, std::vector<Prim> &prims) {
    prims.resize(tris.size());
    for (size_t i = 0; i < tris.size(); i++) {
        const lmcd::TriData &T = tris[i];
        for (int k = 0; k < 3; k++) {
            float p0 = T.p0[k], p1 = T.p0[k] + T.e1[k], p2 = T.p0[k] + T.e2[k];
            prims[i].bmin[k] = std::min(p0, std::min(p1, p2));
            prims[i].bmax[k] = std::max(p0, std::max(p1, p2));
        }
        prims[i].id = (int)i;
        prims[i].code = 0;
    }
}

static void WrapSingleLeaf(LbvhResult &out, int root, const float *bmin, const float *bmax) {
    if (root >= 0) return;  // the whole scene is one leaf: wrap it so that node 0 is an inner node
    lmcd::BvhNode nd;
    memset(&nd, 0, sizeof(nd));
    for (int k = 0; k < 3; k++) nd.lmin[k] = bmin[k], nd.lmax[k] = bmax[k], nd.rmin[k] = INFINITY, nd.rmax[k] = -INFINITY;
    nd.left = root;
    nd.right = root;
    out.nodes.push_back(nd);
}

LbvhResult BuildSahBvh(const std::vector<lmcd::TriData> &tris, int maxLeaf) {
    SahBuilder B;
    B.tris = &tris;
    B.maxLeaf = std::max(1, std::min(8, maxLeaf));
    if (tris.empty()) return B.out;
    PrimBounds(tris, B.prims);
    float bmin[3], bmax[3];
    int root = B.Build(0, (int)tris.size(), bmin, bmax, 1);
    WrapSingleLeaf(B.out, root, bmin, bmax);
    if (B.out.depth > lmcd::BVH_STACK) throw std::runtime_error("BVH deeper than the traversal stack");
    return B.out;
}

LbvhResult BuildSceneBvh(const std::vector<lmcd::TriData> &tris) {
    const char *mode = getenv("LMC_BVH");
    int leaf = 4;
    if (const char *e = getenv("LMC_BVH_LEAF")) leaf = std::max(1, std::min(8, atoi(e)));
    if (mode && std::string(mode) == "lbvh") return BuildLbvh(tris);
    return BuildSahBvh(tris, leaf);
}

namespace {
struct Child2 {
    int ref;  // >= 0 inner node of the binary tree, < 0 leaf code
    float bmin[3], bmax[3];
};
float HalfArea(const Child2 &c) {
    const float dx = c.bmax[0] - c.bmin[0], dy = c.bmax[1] - c.bmin[1], dz = c.bmax[2] - c.bmin[2];
    return dx * dy + dy * dz + dz * dx;
}
void ChildrenOf(const LbvhResult &bvh, int node, Child2 out[2]) {
    const lmcd::BvhNode &nd = bvh.nodes[node];
    out[0].ref = nd.left, out[1].ref = nd.right;
    for (int k = 0; k < 3; k++) out[0].bmin[k] = nd.lmin[k], out[0].bmax[k] = nd.lmax[k], out[1].bmin[k] = nd.rmin[k], out[1].bmax[k] = nd.rmax[k];
}
// returns the index of the wide node made from binary node `node`; `pending` = stack entries above this node's own visit
int Collapse(const LbvhResult &bvh, int node, Bvh4Result &out, int depth, int pending) {
    Child2 ch[4];
    int n = 2;
    ChildrenOf(bvh, node, ch);
    if (ch[1].ref == ch[0].ref && ch[0].ref < 0) n = 1;  // WrapSingleLeaf: the scene is one leaf, stored twice
    while (n < 4) {
        int pick = -1;
        for (int k = 0; k < n; k++)
            if (ch[k].ref >= 0 && (pick < 0 || HalfArea(ch[k]) > HalfArea(ch[pick]))) pick = k;
        if (pick < 0) break;
        Child2 two[2];
        ChildrenOf(bvh, ch[pick].ref, two);
        ch[pick] = two[0];
        ch[n++] = two[1];
    }
    const int me = (int)out.nodes.size();
    out.nodes.emplace_back();
    out.depth = std::max(out.depth, depth);
    out.stackNeed = std::max(out.stackNeed, pending + n - 1);
    lmcd::BvhNode4 nd;
    memset(&nd, 0, sizeof(nd));
    for (int k = 0; k < 4; k++) {
        nd.child[k] = lmcd::BVH4_EMPTY;
        for (int a = 0; a < 3; a++) nd.bmin[k][a] = INFINITY, nd.bmax[k][a] = -INFINITY;
    }
    for (int k = 0; k < n; k++) {
        for (int a = 0; a < 3; a++) nd.bmin[k][a] = ch[k].bmin[a], nd.bmax[k][a] = ch[k].bmax[a];
        // while child k is being walked, at most the n - 1 siblings are pending (fewer in practice: upper bound)
        nd.child[k] = ch[k].ref >= 0 ? Collapse(bvh, ch[k].ref, out, depth + 1, pending + n - 1) : ch[k].ref;
    }
    out.nodes[me] = nd;
    return me;
}
}  // namespace

// The first `topCount` nodes in breadth-first order (root, its children, their children ...), the rest in their depth-first
// order: the top of the tree is one contiguous block -- and that stays
// in a handful of L1 lines when it does not.  Only node numbers change; the children keep their order inside every node, so the
// traversal visits the same nodes in the same order.
static void TopLevelsFirst(Bvh4Result &t, int topCount) {
    const int n = (int)t.nodes.size();
    if (n <= 2) return;
    std::vector<int> order;
    order.reserve(n);
    std::vector<char> taken(n, 0);
    order.push_back(0), taken[0] = 1;
    for (size_t head = 0; head < order.size() && (int)order.size() < topCount; head++)
        for (int k = 0; k < 4; k++) {
            const int c = t.nodes[order[head]].child[k];
            if (c >= 0 && c != lmcd::BVH4_EMPTY && !taken[c] && (int)order.size() < topCount) order.push_back(c), taken[c] = 1;
        }
    for (int i = 0; i < n; i++)
        if (!taken[i]) order.push_back(i);
    std::vector<int> newIndex(n);
    for (int i = 0; i < n; i++) newIndex[order[i]] = i;
    std::vector<lmcd::BvhNode4> nodes(n);
    for (int i = 0; i < n; i++) {
        lmcd::BvhNode4 nd = t.nodes[order[i]];
        for (int k = 0; k < 4; k++)
            if (nd.child[k] >= 0 && nd.child[k] != lmcd::BVH4_EMPTY) nd.child[k] = newIndex[nd.child[k]];
        nodes[i] = nd;
    }
    t.nodes.swap(nodes);
}

// dscene.h BvhNode4Q: child boxes as 8-bit offsets inside the node's box, rounded outwards with a minimum margin (none at the node's own
// faces, where the offset is exact), every bound checked in double precision
static constexpr double QUANT_SLACK = 1.0 / 64;
void QuantizeBvh4(Bvh4Result &t) {
    t.qnodes.resize(t.nodes.size());
    for (size_t i = 0; i < t.nodes.size(); i++) {
        const lmcd::BvhNode4 &nd = t.nodes[i];
        lmcd::BvhNode4Q q;
        memset(&q, 0, sizeof(q));
        for (int a = 0; a < 3; a++) {
            float lo = INFINITY, hi = -INFINITY;
            for (int k = 0; k < 4; k++)
                if (nd.child[k] != lmcd::BVH4_EMPTY) lo = std::min(lo, nd.bmin[k][a]), hi = std::max(hi, nd.bmax[k][a]);
            if (!(lo <= hi)) lo = hi = 0.f;  // no child
            float scale = (float)(((double)hi - (double)lo) / 255.0);
            while ((double)lo + 255.0 * (double)scale < (double)hi) scale = std::nextafter(scale, INFINITY);
            // The frame is anchored at the node's lower OR upper face (org = hi, negative scale): offset 0 is exact there, so a FLAT child lying in
            // the anchored face stays flat.  The face that holds more flat-child area wins.  (A thickened flat child -- a wall, a floor -- is entered
            // again by every ray that leaves that surface.)  The device code is the same either way: the slab test takes min / max of the two bounds.
            double flatLo = 0, flatHi = 0;
            for (int k = 0; k < 4; k++) {
                if (nd.child[k] == lmcd::BVH4_EMPTY || nd.bmin[k][a] != nd.bmax[k][a]) continue;
                const int b = (a + 1) % 3, c = (a + 2) % 3;
                const double area = ((double)nd.bmax[k][b] - nd.bmin[k][b]) * ((double)nd.bmax[k][c] - nd.bmin[k][c]);
                if (nd.bmin[k][a] == lo) flatLo += area;
                else if (nd.bmin[k][a] == hi) flatHi += area;
            }
            const bool fromHi = flatHi > flatLo;
            const double org = fromHi ? (double)hi : (double)lo, sc = fromHi ? -(double)scale : (double)scale;
            q.org[a] = (float)org, q.scale[a] = (float)sc;
            for (int k = 0; k < 4; k++) {
                if (nd.child[k] == lmcd::BVH4_EMPTY) continue;
                // offsets of the child's two bounds, measured from the anchored face: u0 <= u1 in steps
                const double near = fromHi ? (double)hi - (double)nd.bmax[k][a] : (double)nd.bmin[k][a] - (double)lo;
                const double far = fromHi ? (double)hi - (double)nd.bmin[k][a] : (double)nd.bmax[k][a] - (double)lo;
                int a0 = 0, a1 = 255;
                if (scale > 0.f) {
                    // outwards, with at least QUANT_SLACK of a step between the offset and the exact bound (the device's slab distances carry
                    // rounding errors of ~1e-4 step at most); none in the anchored face, where the offset is exact
                    a0 = (int)std::floor(near / (double)scale - QUANT_SLACK), a1 = (int)std::ceil(far / (double)scale + QUANT_SLACK);
                    if (far == 0.0) a1 = 0;  // a flat child in the anchored face stays flat
                    a0 = std::max(0, std::min(255, a0)), a1 = std::max(0, std::min(255, a1));
                    while (a0 > 0 && a0 * (double)scale > near) a0--;
                    while (a1 < 255 && a1 * (double)scale < far) a1++;
                }
                // decoded bounds in world coordinates, checked in double precision
                const double d0 = org + a0 * sc, d1 = org + a1 * sc, dlo = std::min(d0, d1), dhi = std::max(d0, d1);
                if (dlo > (double)nd.bmin[k][a] || dhi < (double)nd.bmax[k][a]) {  // never seen; the scene then simply keeps the exact nodes
                    t.qnodes.clear();
                    return;
                }
                q.qmin[a][k] = (unsigned char)a0, q.qmax[a][k] = (unsigned char)a1;
            }
        }
        for (int k = 0; k < 4; k++) q.child[k] = nd.child[k];
        t.qnodes[i] = q;
    }
}

double ThickenedFlatLeafShare(const Bvh4Result &t) {
    double all = 0, thick = 0;
    if (t.qnodes.size() != t.nodes.size()) return 0.0;
    for (size_t i = 0; i < t.nodes.size(); i++) {
        const lmcd::BvhNode4 &nd = t.nodes[i];
        const lmcd::BvhNode4Q &q = t.qnodes[i];
        for (int k = 0; k < 4; k++) {
            if (nd.child[k] == lmcd::BVH4_EMPTY || nd.child[k] >= 0) continue;  // leaf children only
            double e[3];
            bool thickened = false;
            for (int a = 0; a < 3; a++) {
                e[a] = (double)nd.bmax[k][a] - (double)nd.bmin[k][a];
                const double g = ((double)q.qmax[a][k] - (double)q.qmin[a][k]) * (double)q.scale[a];
                if (e[a] < 0.25 * (double)q.scale[a] && g >= (double)q.scale[a] && g > 0) thickened = true;
            }
            const double sa = e[0] * e[1] + e[1] * e[2] + e[0] * e[2];
            all += sa;
            if (thickened) thick += sa;
        }
    }
    return all > 0 ? thick / all : 0.0;
}

Bvh4Result CollapseToBvh4(const LbvhResult &bvh) {
    Bvh4Result out;
    out.leafTris = bvh.leafTris;
    if (bvh.nodes.empty()) return out;
    Collapse(bvh, 0, out, 1, 0);
    if (out.stackNeed > lmcd::BVH_STACK) throw std::runtime_error("BVH needs a deeper traversal stack than BVH_STACK");
    TopLevelsFirst(out, lmcd::BVH_TOP_NODES);
    QuantizeBvh4(out);
    return out;
}

LbvhResult BuildLbvh(const std::vector<lmcd::TriData> &tris) {
    Builder B;
    B.tris = &tris;
    if (const char *e = getenv("LMC_BVH_LEAF")) B.maxLeaf = std::max(1, std::min(8, atoi(e)));
    const int n = (int)tris.size();
    if (n == 0) return B.out;
    B.prims.resize(n);
    float cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
    std::vector<float> cen((size_t)n * 3);
    for (int i = 0; i < n; i++) {
        const lmcd::TriData &T = tris[i];
        for (int k = 0; k < 3; k++) {
            float p0 = T.p0[k], p1 = T.p0[k] + T.e1[k], p2 = T.p0[k] + T.e2[k];
            B.prims[i].bmin[k] = std::min(p0, std::min(p1, p2));
            B.prims[i].bmax[k] = std::max(p0, std::max(p1, p2));
            cen[(size_t)i * 3 + k] = 0.5f * (B.prims[i].bmin[k] + B.prims[i].bmax[k]);
            cmin[k] = std::min(cmin[k], cen[(size_t)i * 3 + k]);
            cmax[k] = std::max(cmax[k], cen[(size_t)i * 3 + k]);
        }
        B.prims[i].id = i;
    }
    for (int i = 0; i < n; i++) {
        uint64_t code = 0;
        for (int k = 0; k < 3; k++) {
            float ext = cmax[k] - cmin[k];
            double f = ext > 0 ? (double)(cen[(size_t)i * 3 + k] - cmin[k]) / ext : 0.0;
            uint32_t q = (uint32_t)std::min(2097151.0, std::max(0.0, f * 2097152.0));
            code |= ExpandBits21(q) << (2 - k);
        }
        B.prims[i].code = code;
    }
    std::sort(B.prims.begin(), B.prims.end(), [](const Prim &a, const Prim &b) { return a.code != b.code ? a.code < b.code : a.id < b.id; });
    float bmin[3], bmax[3];
    int root = B.Build(0, n, bmin, bmax, 1);
    if (root < 0) {  // the whole scene is one leaf: wrap it so that node 0 is an inner node
        lmcd::BvhNode nd;
        memset(&nd, 0, sizeof(nd));
        for (int k = 0; k < 3; k++) nd.lmin[k] = bmin[k], nd.lmax[k] = bmax[k], nd.rmin[k] = INFINITY, nd.rmax[k] = -INFINITY;
        nd.left = root;
        nd.right = root;
        B.out.nodes.push_back(nd);
    }
    if (B.out.depth > lmcd::BVH_STACK) throw std::runtime_error("LBVH deeper than the traversal stack");
    return B.out;
}

// ------------------------------------------------------------------------------------------------ kd-tree
namespace {
struct Interval {
    float low, high;
};
struct KdBuilder {
    const float *pts;
    int dim;
    KdTreeResult out;

    float Get(int idx, int d) const { return pts[(size_t)idx * dim + d]; }
    void PlaneSplit(int *ind, int count, int cutfeat, float cutval, int &lim1, int &lim2) const {  // nanoflann.hpp:976-1011
        int left = 0, right = count - 1;
        for (;;) {
            while (left <= right && Get(ind[left], cutfeat) < cutval) ++left;
            while (right && left <= right && Get(ind[right], cutfeat) >= cutval) --right;
            if (left > right || !right) break;
            std::swap(ind[left], ind[right]);
            ++left;
            --right;
        }
        lim1 = left;
        right = count - 1;
        for (;;) {
            while (left <= right && Get(ind[left], cutfeat) <= cutval) ++left;
            while (right && left <= right && Get(ind[right], cutfeat) > cutval) --right;
            if (left > right || !right) break;
            std::swap(ind[left], ind[right]);
            ++left;
            --right;
        }
        lim2 = left;
    }
    // The spreads of ALL coordinates in one pass over the rows (each row is read once, contiguously) instead of one strided pass per candidate
    // coordinate: the same minima and maxima, and the build of a 3000-point tree takes half the time -- it sits between two steps of the
    // population when a gradient cache becomes ready (context.cpp CacheApplyFinish).
    static constexpr int MAX_DIM = 16;
    void MinMaxAll(const int *ind, int count, float *mn, float *mx) const {
        const float *p0 = pts + (size_t)ind[0] * dim;
        for (int d = 0; d < dim; ++d) mn[d] = mx[d] = p0[d];
        for (int i = 1; i < count; ++i) {
            const float *p = pts + (size_t)ind[i] * dim;
            for (int d = 0; d < dim; ++d) {
                const float v = p[d];
                mn[d] = v < mn[d] ? v : mn[d];
                mx[d] = v > mx[d] ? v : mx[d];
            }
        }
    }
    int Divide(int left, int right, Interval *bbox, int depth = 1) {  // nanoflann.hpp:867-917
        out.depth = std::max(out.depth, depth);
        int ni = (int)out.nodes.size();
        lmcd::KdNode nd;
        memset(&nd, 0, sizeof(nd));
        nd.child1 = nd.child2 = -1;
        out.nodes.push_back(nd);
        if ((right - left) <= 10) {
            out.nodes[ni].left = left, out.nodes[ni].right = right;
            float mn[MAX_DIM], mx[MAX_DIM];
            MinMaxAll(&out.vind[0] + left, right - left, mn, mx);
            for (int i = 0; i < dim; ++i) bbox[i].low = mn[i], bbox[i].high = mx[i];
            return ni;
        }
        int *ind = &out.vind[0] + left;
        const int count = right - left;
        const float EPS = 0.00001f;  // middleSplit_, nanoflann.hpp:919-966
        float max_span = bbox[0].high - bbox[0].low;
        for (int i = 1; i < dim; ++i) max_span = std::max(max_span, bbox[i].high - bbox[i].low);
        float max_spread = -1;
        int cutfeat = 0;
        float mnAll[MAX_DIM], mxAll[MAX_DIM];
        MinMaxAll(ind, count, mnAll, mxAll);
        for (int i = 0; i < dim; ++i) {
            float span = bbox[i].high - bbox[i].low;
            if (span > (1 - EPS) * max_span) {
                float spread = mxAll[i] - mnAll[i];
                if (spread > max_spread) cutfeat = i, max_spread = spread;
            }
        }
        float split_val = (bbox[cutfeat].low + bbox[cutfeat].high) / 2;
        const float mn = mnAll[cutfeat], mx = mxAll[cutfeat];
        float cutval = split_val < mn ? mn : (split_val > mx ? mx : split_val);
        int lim1, lim2;
        PlaneSplit(ind, count, cutfeat, cutval, lim1, lim2);
        int idx = lim1 > count / 2 ? lim1 : (lim2 < count / 2 ? lim2 : count / 2);
        out.nodes[ni].divfeat = cutfeat;
        Interval lb[MAX_DIM], rb[MAX_DIM];
        for (int i = 0; i < dim; ++i) lb[i] = rb[i] = bbox[i];
        lb[cutfeat].high = cutval;
        int c1 = Divide(left, left + idx, lb, depth + 1);
        rb[cutfeat].low = cutval;
        int c2 = Divide(left + idx, right, rb, depth + 1);
        out.nodes[ni].child1 = c1, out.nodes[ni].child2 = c2;
        out.nodes[ni].divlow = lb[cutfeat].high;
        out.nodes[ni].divhigh = rb[cutfeat].low;
        for (int i = 0; i < dim; ++i) bbox[i].low = std::min(lb[i].low, rb[i].low), bbox[i].high = std::max(lb[i].high, rb[i].high);
        return ni;
    }
};
}  // namespace

void ChooseGridCoords(const float *pts, int n, int dim, int m, int *coord) {
    using namespace lmcd;
    m = std::min(std::min(m, dim), 4);
    const char *e = getenv("LMC_GRID_COORDS");
    if (e && std::string(e) == "first") {
        for (int k = 0; k < m; k++) coord[k] = k;
        return;
    }
    const int G = CacheGridG(dim);
    std::vector<long long> score(dim, 0);
    std::vector<int> hist(G);
    for (int c = 0; c < dim; c++) {
        std::fill(hist.begin(), hist.end(), 0);
        for (int i = 0; i < n; i++) hist[CacheGridCell(pts[(size_t)i * dim + c], G)]++;
        for (int h : hist) score[c] += (long long)h * h;
    }
    std::vector<int> order(dim);
    for (int c = 0; c < dim; c++) order[c] = c;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return score[a] < score[b]; });
    std::sort(order.begin(), order.begin() + m);
    for (int k = 0; k < m; k++) coord[k] = order[k];
}

CacheGrid BuildCacheGrid(const float *pts, int n, int dim, int m, const int *coord) {
    using namespace lmcd;
    CacheGrid g;
    g.G = CacheGridG(dim), g.m = std::min(std::min(m, dim), 4);
    if (coord) {
        for (int k = 0; k < g.m; k++) g.coord[k] = coord[k];
    } else {
        ChooseGridCoords(pts, n, dim, g.m, g.coord);
    }
    const int G = g.G;
    size_t cells = 1;
    for (int k = 0; k < g.m; k++) cells *= G;
    int nbrs = 1;
    for (int k = 0; k < g.m; k++) nbrs *= 3;
    // (cell, point) pairs of the dilation, counting-sorted by cell
    std::vector<int> pairCell;
    pairCell.reserve((size_t)n * nbrs);
    g.start.assign(cells + 1, 0);
    auto forNeighbours = [&](const float *p, auto &&fn) {
        int c[8];
        for (int k = 0; k < g.m; k++) c[k] = CacheGridCell(p[g.coord[k]], G);
        for (int o = 0; o < nbrs; o++) {
            int cell = 0, t = o;
            bool inside = true;
            for (int k = 0; k < g.m; k++, t /= 3) {
                const int ck = c[k] + (t % 3) - 1;
                inside = inside && ck >= 0 && ck < G;
                cell = cell * G + ck;
            }
            if (inside) fn(cell);
        }
    };
    for (int i = 0; i < n; i++) forNeighbours(pts + (size_t)i * dim, [&](int cell) { g.start[cell + 1]++; });
    for (size_t c = 0; c < cells; c++) g.start[c + 1] += g.start[c];
    g.rows.resize((size_t)g.start[cells] * dim);
    std::vector<int> cursor(g.start.begin(), g.start.end() - 1);
    for (int i = 0; i < n; i++)
        forNeighbours(pts + (size_t)i * dim, [&](int cell) {
            std::copy(pts + (size_t)i * dim, pts + (size_t)(i + 1) * dim, g.rows.begin() + (size_t)cursor[cell]++ * dim);
        });
    return g;
}
bool CacheGrid::Exists(const float *q, int dim) const {
    using namespace lmcd;
    const float radiusSq = dim * (PSS_QUERY_DIST * PSS_QUERY_DIST);
    int cell = 0;
    for (int k = 0; k < m; k++) cell = cell * G + CacheGridCell(q[coord[k]], G);
    bool any = false;
    for (int j = start[cell]; j < start[cell + 1]; j++) {
        const float *p = rows.data() + (size_t)j * dim;
        float d = 0.f;
        for (int k = 0; k < dim; k++) {
            const float diff = q[k] - p[k];
            d += diff * diff;
        }
        any = any || d < radiusSq;
    }
    return any;
}

KdTreeResult BuildKdTree(const float *pts, int n, int dim) {
    KdBuilder B;
    B.pts = pts, B.dim = dim;
    B.out.vind.resize(n);
    for (int i = 0; i < n; i++) B.out.vind[i] = i;
    if (dim > KdBuilder::MAX_DIM) throw std::runtime_error("kd-tree of more than 16 coordinates");
    if (n <= 0) throw std::runtime_error("kd-tree of no points");
    Interval bbox[KdBuilder::MAX_DIM];
    {
        float mn[KdBuilder::MAX_DIM], mx[KdBuilder::MAX_DIM];
        B.MinMaxAll(B.out.vind.data(), n, mn, mx);
        for (int i = 0; i < dim; ++i) bbox[i].low = mn[i], bbox[i].high = mx[i];
    }
    B.out.nodes.reserve((size_t)n / 2 + 16);
    B.Divide(0, n, bbox);
    if (B.out.depth + 1 > lmcd::KD_STACK) throw std::runtime_error("global-cache kd-tree deeper than the in-kernel search stack");
    B.out.rootLow.resize(dim), B.out.rootHigh.resize(dim);
    for (int i = 0; i < dim; i++) B.out.rootLow[i] = bbox[i].low, B.out.rootHigh[i] = bbox[i].high;
    return B.out;
}

}  // namespace lmc
