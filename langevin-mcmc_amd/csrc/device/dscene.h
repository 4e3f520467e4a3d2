// Flattened scene in HBM (read-only during rendering; ~a few MB, L2/MALL resident) and the LBVH
// closest-hit / any-hit traversal that stands in for Embree's rtcIntersect1 / rtcOccluded1
// (/root/reference/src/scene.cpp:106-149).  Built by host/context.cpp from lmc::Scene.
#pragma once
#include "dmath.h"

namespace lmcd {

enum { BSDF_LAMBERTIAN = 0, BSDF_PHONG = 1, BSDF_ROUGHDIELECTRIC = 2 };
enum { LIGHT_POINT = 0, LIGHT_AREA = 1, LIGHT_ENV = 2 };

// LBVH node, 64 B = one cache line: both children's boxes live in the parent so one fetch feeds two
// slab tests.  child >= 0: inner node index; child < 0: leaf, ~child = (firstTri << 3) | (count - 1), count <= 4.
struct alignas(16) BvhNode {
    float lmin[3], lmax[3], rmin[3], rmax[3];
    int left, right;
    int pad[2];
};
// The tree the kernels walk: the binary tree collapsed to four children per node (host/accel.cpp CollapseToBvh4), 128 B = one
// fetch round for four slab tests.  A ray visits about half as many nodes as in the binary tree, and each visit is one
// dependent memory round trip, which is what the step kernels wait on (profiles/r02_h_lean_regions.json: 32 % of a wave's
// cycles inside the closest-hit traversal).  child: >= 0 inner node, < 0 leaf code as above, BVH4_EMPTY = no child.
constexpr int BVH4_EMPTY = 0x7fffffff;
struct alignas(16) BvhNode4 {
    float bmin[4][3], bmax[4][3];  // child k: bmin[k], bmax[k]
    int child[4];
    int pad[4];
};
static_assert(sizeof(BvhNode4) == 128, "one node = two cache lines of 64 B");
// The same node in 64 B (build option LMC_BVH_QUANT=1, off by default): the children's boxes as 8-bit offsets inside the node's own box,
//   child k, axis a:  [org[a] + qmin[a][k] * scale[a],  org[a] + qmax[a][k] * scale[a]]  contains  [bmin[k][a], bmax[k][a]]
// rounded outwards with a checked margin (host/accel.cpp QuantizeBvh4 verifies every bound in double precision; none in the face the node's
// frame is anchored at, where offset 0 is exact).  Why it was built: once the chains are grouped by technique (relocate.hip) the closest-hit traversal is 41 % of
// the lean kernel (profiles/r04_reloc_g_*), and what a node visit costs is the vector L1's work for 64 lanes x 128 B from 64 different lines --
// seven 16-byte loads per lane; this node takes four, from one 64 B half line.  Boxes only cull: a larger box means a visit more, never another
// hit, so every intersection result stays what the exact boxes give (tests/helpers/bvh_stats.cpp walks both forms with the device's arithmetic).
// What the format cannot represent is a FLAT child away from its node's lower face (a wall): thickened to a step or two, it is entered again by
// every ray that leaves that surface (veach-door: leaf visits per ray 1.22 -> 1.71, -9 % chain-steps/s; torus: 1.03 -> 1.04, +2.5 .. +6 %;
// accel.cpp ThickenedFlatLeafShare tells the two kinds of scene apart: 0.005 vs 0.55).  Anchoring a node's frame at whichever face holds more
// flat-child area (negative scale; same device code) brings the door to 1.55 -- interior flat children remain.  Both formats in one build, chosen per scene at load
// time, was measured too (profiles/r04_nodes_a_*): the kernels lose on the exact path what the torus gains (veach-door 181 -> 140 M), so the
// format stays a build option.
struct alignas(16) BvhNode4Q {
    float org[3], scale[3];
    unsigned char qmin[3][4], qmax[3][4];  // [axis][child]
    int child[4];
};
static_assert(sizeof(BvhNode4Q) == 64, "one node = one 64 B half line");
#ifndef LMC_BVH_QUANT
#define LMC_BVH_QUANT 0  // build option: 1 = EVERY kernel walks the quantised nodes (A/B; the product chooses per scene and per launch: LdsStackT::kQuant)
#endif
// triangle in BVH leaf order, 48 B: Moeller-Trumbore operands + global triangle id
struct alignas(16) LeafTri {
    float p0[3];
    int id;
    float e1[3];
    float pad0;
    float e2[3];
    float pad1;
};
// shading data by global triangle id, 112 B
struct alignas(16) TriData {
    float p0[3], e1[3], e2[3];
    float n0[3], n1[3], n2[3];
    float st[6];  // per-vertex st (valid iff hasST)
    int mesh;
    // copies of the mesh's fields: what a hit needs sits in the triangle's own record, one fetch instead of a chain of dependent ones
    int hasST, material, areaLight;
};
struct DMesh {
    int material, areaLight, hasST, triBase, numTris;
    float totalArea, invTotalArea;
    // PiecewiseConstant1D over triangle areas (area lights only): offsets into DScene::areaFunc / areaCdf
    int areaOff, areaCdfOff;
    float areaFuncInt;
};
// constant or bitmap texture reference (texture.h, constanttexture.h, bitmaptexture.h)
// LMC_TEX_INLINE (default): a TEXTURED reference holds what a look-up needs in the words a constant one uses for its value: `bitmap` = the word offset of
// the bitmap's texels in DScene::texPool, value[0..1] = the bits of its width / height, value[2] = its gamma (filled by host/context.cpp UploadScene).
// The look-up then goes material record -> texels; through DScene::bitmaps[bitmap] it was material record -> bitmap header -> texels, one more
// dependent round trip per textured BSDF evaluation.  0: the index form (A/B).
#ifndef LMC_TEX_INLINE
#define LMC_TEX_INLINE 1
#endif
struct DTexRef {
    int bitmap;  // -1 = constant; else LMC_TEX_INLINE ? word offset into DScene::texPool : index into DScene::bitmaps
    float value[3];
    float sScale, tScale;
};
struct DBitmap {
    const float *pix;  // W*H*3, row-major
    int W, H;
    float gamma;  // 2.2 for 8-bit files (bitmaptexture.h:135-144)
};
struct DMaterial {
    int type, twoSided;
    DTexRef Kd, Ks, Kt;  // Lambertian: Kd; Phong: Kd, Ks; RoughDielectric: Ks, Kt
    DTexRef expOrAlpha;  // Phong exponent / dielectric alpha (channel 0)
    float eta, invEta, KsWeight;
};
struct DLight {
    int type;
    float samplingWeight;
    float pos[3], intensity[3];  // point
    int mesh;                    // area
    float radiance[3];
};
struct DEnv {
    int W, H;
    const float *image;  // W*H*3
    const float *cdfRows, *cdfCols, *rowWeights;
    float normalization, pixelSize[2];
    float toWorld[16], toLight[16];
    float xformBlocks[30];  // the two 15-float AnimatedTransform blocks, as Light::Serialize writes them
};
struct DCamera {
    float sampleToCam[16], camToSample[16], toWorld[16], worldToCamera[16];
    int width, height;
    float nearClip, farClip, dist;
};
struct DOptions {
    int minDepth, maxDepth, mala, h2mc;
    float roughnessThreshold, largeStepProbability, largeStepProbScale;
    float malaGN, malaStepsize, malaStdDev, perturbStdDev, discreteStdDev, uniformMixingProbability;
    int seedOffset;
    int leanLightless;  // no state of this scene can have a light sub-path (its only emitter is the environment map, whose light sub-paths carry
                        // nothing): the lean launch runs the instantiation without the light-sub-path code; a state with l > 1, should one
                        // ever appear, takes the generic launch (dstep.h QueueNext)
    int sampleCache;    // samplecache with mala (LargeStepCache, mlt.cpp:71-73): every small step takes the generic launch, which keeps chain.path
    int useLightCoord;  // uselightcoordinatesampling (path.cpp:1339-1360, 1881-1951): every small step then takes the generic launch
};

struct DScene {
    const BvhNode4 *nodes;
    const BvhNode4Q *qnodes;  // the same tree, quantised boxes
    const LeafTri *leafTris;
    const TriData *tris;
    const DMesh *meshes;
    const DMaterial *materials;
    const DBitmap *bitmaps;
    const float *texPool;  // the texels of every bitmap (DTexRef::bitmap)
    const DLight *lights;
    const float *areaFunc, *areaCdf;
    const float *lightFunc, *lightCdf;
    float lightFuncInt, lightWeightSum;
    int numTris, numNodes, numMeshes, numLights, envLight, numMaterials;
    int glossy;  // any non-Lambertian BSDF: selects the kernel instantiations that carry the Phong / rough-dielectric code
    DEnv env;
    DCamera cam;
    DOptions opt;
    float bsCenter[3], bsRadius;
    float sceneParams[38];
};

// ---------------------------------------------------------------------------------------------- ray / tri
// Same arithmetic, in the same order, as the oracle's TriTest (oracle/scene_rt.cpp) which restates the
// reference's TriangleIntersect (trianglemesh.cpp:30-53): the build compiles with -ffp-contract=off so that
// the accepted (id, t) pairs are bit-identical between CPU and GPU.
LMC_HD bool TriTest(const float *p0, const float *e1, const float *e2, V3 org, V3 dir, float tnear, float tfar, float &t) {
    V3 E1{e1[0], e1[1], e1[2]}, E2{e2[0], e2[1], e2[2]};
    V3 s1 = Cross(dir, E2);
    float divisor = Dot(s1, E1);
    if (divisor == 0.0f) return false;
    float invDivisor = inverse(divisor);
    V3 s = org - V3{p0[0], p0[1], p0[2]};
    float u = Dot(s, s1) * invDivisor;
    V3 s2 = Cross(s, E1);
    float v = Dot(dir, s2) * invDivisor;
    if (!(u >= 0.0f && v >= 0.0f && u + v <= 1.0f)) return false;
    float tt = Dot(E2, s2) * invDivisor;
    if (!(tt >= tnear && tt <= tfar)) return false;
    t = tt;
    return true;
}

LMC_D bool SlabTest(const float *bmin, const float *bmax, V3 org, V3 invd, float tnear, float tfar, float &tEntry) {
    float ax = (bmin[0] - org.x) * invd.x, bx = (bmax[0] - org.x) * invd.x;
    float ay = (bmin[1] - org.y) * invd.y, by = (bmax[1] - org.y) * invd.y;
    float az = (bmin[2] - org.z) * invd.z, bz = (bmax[2] - org.z) * invd.z;
    float t0 = fmaxf(fmaxf(tnear, fminf(ax, bx)), fmaxf(fminf(ay, by), fminf(az, bz)));
    float t1 = fminf(fminf(tfar, fmaxf(ax, bx)), fminf(fmaxf(ay, by), fmaxf(az, bz)));
    tEntry = t0;
    return t0 * 0.9999996f <= t1 * 1.0000004f;  // 2*gamma(3) widening, as in the oracle
}

// The host numbers the top of the four-wide tree breadth first (accel.cpp TopLevelsFirst): nodes [0, BVH_TOP_NODES) are the root,
// its children and grandchildren ... (1 + 4 + 16 + 64 = 85 when every node is full), so that the top of the tree sits in a handful of
// cache lines.  (Round 3 staged those nodes in LDS: no gain, profiles/r03_b_ab_bvh_lds_top_rejected.jsonl; the code is gone.)
constexpr int BVH_TOP_NODES = 85;

constexpr int BVH_STACK = 64;      // host-checked bound on the LBVH depth
// entries of the per-thread LDS stack (the host routes deeper trees to the private-memory instantiations).  40: the four-wide tree of the
// veach-door scene needs 37 (tests/helpers/bvh_stats.cpp); with 32 that scene ran every launch on its scratch-stack fallback
// (profiles/r03_e_door_kernel_stats.csv: large steps 11.9 ms, lean 4.5 ms, the cache-filling launch on the round-1 generic kernel)
constexpr int BVH_LDS_STACK = 40;

// Traversal stack policies.  Private arrays indexed at run time live in scratch memory, which on gfx950 is HBM-backed
// and was the bottleneck of the first version of the step kernel (profiles/r01_a_*): the hot kernels keep the stack
// in LDS instead, laid out [entry][thread] so that a wave's accesses are conflict-free.
// Both policies also carry kGlossy: whether the scene has non-Lambertian BSDFs.  Every kernel is instantiated for both
// values so that Lambertian-only scenes (BASELINE.json configs[1]) do not pay registers / code for Phong and the rough
// dielectric.
// ... and kQuant (round 5): whether the launch walks the 64-byte quantised nodes (DScene::qnodes) or the exact 128-byte ones.  Chosen PER SCENE
// when the scene is loaded (host/context.cpp UploadScene: the quantised nodes suit a tree unless it is full of flat leaves away from their node's
// faces -- accel.h ThickenedFlatLeafShare) and compiled in as a template parameter of the two hot launches (the lean small steps, the large
// steps): a run-time branch inside the kernels cost more than the smaller nodes save (profiles/r04_nodes_a_*).
template <bool GLOSSY, bool QUANT = false>
struct LocalStackT {
    static constexpr bool kGlossy = GLOSSY;
    static constexpr bool kQuant = QUANT;
    int s[BVH_STACK];
    int sp = 0;
    LMC_D void Reset() { sp = 0; }
    LMC_D bool Empty() const { return sp == 0; }
    LMC_D void Push(int v) {
        if (sp < BVH_STACK) s[sp++] = v;
    }
    LMC_D int Pop() { return s[--sp]; }
};
template <bool GLOSSY, bool QUANT = false>
struct LdsStackT {
    static constexpr bool kGlossy = GLOSSY;
    static constexpr bool kQuant = QUANT;
    int *base;   // &lds[threadIdx.x]
    int stride;  // blockDim.x
    int sp;      // the top's word offset from base = entries x stride, advanced by additions: with an entry COUNT every push and pop paid a 32-bit
                 // integer multiply (v_mul_lo_u32, a quarter-rate instruction) for its address -- four per node visit of the closest-hit walk (hipcc -S)
    LMC_D void Reset() { sp = 0; }
    LMC_D bool Empty() const { return sp == 0; }
#ifdef LMC_STACK_MUL  // A/B build: the entry-count form
    LMC_D void Push(int v) {
        if (sp < BVH_LDS_STACK) base[sp * stride] = v, sp++;
    }
    LMC_D int Pop() {
        --sp;
        return base[sp * stride];
    }
#else
    LMC_D void Push(int v) {
        if (sp < BVH_LDS_STACK * stride) base[sp] = v, sp += stride;
    }
    LMC_D int Pop() {
        sp -= stride;
        return base[sp];
    }
#endif
};

// closest hit: smallest t in [tnear, tfar]; ties -> lower global triangle id (tree-independent answer).
// "while-while" traversal: all lanes of a wave first descend through inner nodes until each has reached a leaf (or
// finished), then all test their leaf's triangles together.  With the node test and the leaf test as two branches of
// one loop a wave executed both bodies on almost every iteration (profiles/r01_d: 5270 vector-memory instructions per
// wave-step for ~800 per lane); the visiting order, hence the result, is the same depth-first order.
// one inner-node visit: slab-tests the four children; returns the nearest hit child (or BVH4_EMPTY) and pushes the others,
// farthest first, so that they are popped nearest first (ORDERED = false: any order, for the any-hit query)
template <bool ORDERED, class Stk>
LMC_D int VisitNode4(const BvhNode4 &nd, V3 org, V3 invd, float tnear, float tfar, Stk &stk) {
    float t[4];
    bool h[4];
#pragma unroll
    for (int k = 0; k < 4; k++) h[k] = nd.child[k] != BVH4_EMPTY && SlabTest(nd.bmin[k], nd.bmax[k], org, invd, tnear, tfar, t[k]);
    int next = BVH4_EMPTY;
    if (!ORDERED) {
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (h[k]) {
                if (next == BVH4_EMPTY) next = nd.child[k];
                else
                    stk.Push(nd.child[k]);
            }
        return next;
    }
    // sort the (up to four) hits by entry distance with a 5-comparator network on (t, child) pairs; misses carry +inf
    float tk[4];
    int ck[4];
#pragma unroll
    for (int k = 0; k < 4; k++) tk[k] = h[k] ? t[k] : INFINITY, ck[k] = h[k] ? nd.child[k] : BVH4_EMPTY;
    auto cswap = [&](int a, int b) {
        if (tk[b] < tk[a]) {
            const float tt = tk[a];
            tk[a] = tk[b], tk[b] = tt;
            const int cc = ck[a];
            ck[a] = ck[b], ck[b] = cc;
        }
    };
    cswap(0, 1), cswap(2, 3), cswap(0, 2), cswap(1, 3), cswap(1, 2);
    if (ck[3] != BVH4_EMPTY) stk.Push(ck[3]);
    if (ck[2] != BVH4_EMPTY) stk.Push(ck[2]);
    if (ck[1] != BVH4_EMPTY) stk.Push(ck[1]);
    return ck[0];
}

// one inner-node visit on the quantised node: the slab distances straight from the 8-bit offsets,
//   t = ((org_n - o) * invd) + q * (scale * invd)      (one convert + one fused multiply-add per bound, as many instructions as the exact form)
// The rounding of the three products / sums is covered by the quantisation margin where the node's extent dominates (error <= 1e-6 steps)
// and by the widened comparison where the distance dominates.  A zero direction component gives inf / NaN slab distances, which fminf /
// fmaxf drop: the axis then culls nothing.
template <bool ORDERED, class Stk>
LMC_D int VisitNode4Q(const BvhNode4Q &nd, V3 org, V3 invd, float tnear, float tfar, Stk &stk) {
    const float A[3] = {(nd.org[0] - org.x) * invd.x, (nd.org[1] - org.y) * invd.y, (nd.org[2] - org.z) * invd.z};
    const float B[3] = {nd.scale[0] * invd.x, nd.scale[1] * invd.y, nd.scale[2] * invd.z};
    float t[4];
    bool h[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float ax = __builtin_fmaf((float)nd.qmin[0][k], B[0], A[0]), bx = __builtin_fmaf((float)nd.qmax[0][k], B[0], A[0]);
        const float ay = __builtin_fmaf((float)nd.qmin[1][k], B[1], A[1]), by = __builtin_fmaf((float)nd.qmax[1][k], B[1], A[1]);
        const float az = __builtin_fmaf((float)nd.qmin[2][k], B[2], A[2]), bz = __builtin_fmaf((float)nd.qmax[2][k], B[2], A[2]);
        const float t0 = fmaxf(fmaxf(tnear, fminf(ax, bx)), fmaxf(fminf(ay, by), fminf(az, bz)));
        const float t1 = fminf(fminf(tfar, fmaxf(ax, bx)), fminf(fmaxf(ay, by), fmaxf(az, bz)));
        t[k] = t0;
        h[k] = nd.child[k] != BVH4_EMPTY && t0 * 0.9999992f <= t1 * 1.0000008f;
    }
    int next = BVH4_EMPTY;
    if (!ORDERED) {
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (h[k]) {
                if (next == BVH4_EMPTY) next = nd.child[k];
                else
                    stk.Push(nd.child[k]);
            }
        return next;
    }
    float tk[4];
    int ck[4];
#pragma unroll
    for (int k = 0; k < 4; k++) tk[k] = h[k] ? t[k] : INFINITY, ck[k] = h[k] ? nd.child[k] : BVH4_EMPTY;
    auto cswap = [&](int a, int b) {
        if (tk[b] < tk[a]) {
            const float tt = tk[a];
            tk[a] = tk[b], tk[b] = tt;
            const int cc = ck[a];
            ck[a] = ck[b], ck[b] = cc;
        }
    };
    cswap(0, 1), cswap(2, 3), cswap(0, 2), cswap(1, 3), cswap(1, 2);
    if (ck[3] != BVH4_EMPTY) stk.Push(ck[3]);
    if (ck[2] != BVH4_EMPTY) stk.Push(ck[2]);
    if (ck[1] != BVH4_EMPTY) stk.Push(ck[1]);
    return ck[0];
}
// hipcc sinks a load to its first use: a node's child indices (needed after the slab arithmetic) and a leaf triangle's p0 (needed after the
// divisor test, inside a branch) were fetched there, each a dependent round trip through the vector L1 on a line the visit had already paid for
// (hipcc -S of the round-5 lean kernel: `global_load_dwordx4 ... offset:48` + `s_waitcnt vmcnt(0)` behind the min / max chain; one
// `global_load_dwordx3` per triangle behind `v_cmp_neq_f32 divisor`).  An empty asm that names the registers as in / out pins the loads where
// the source has them: all of a visit's loads are in flight together.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LMC_NO_LOAD_PIN)
#define LMC_PIN2(a, b) asm volatile("" : "+v"(a), "+v"(b))
#define LMC_PIN3(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c))
#define LMC_PIN5(a, b, c, d, e) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e))
#define LMC_PIN4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#else
#define LMC_PIN2(a, b) ((void)0)
#define LMC_PIN3(a, b, c) ((void)0)
#define LMC_PIN5(a, b, c, d, e) ((void)0)
#define LMC_PIN4(a, b, c, d) ((void)0)
#endif
// one visit of inner node `cur`: the nearest hit child (or BVH4_EMPTY), the others pushed
template <bool ORDERED, class Stk>
LMC_D int VisitInner(const DScene &S, int cur, V3 org, V3 invd, float tnear, float tfar, Stk &stk) {
    if constexpr (Stk::kQuant || LMC_BVH_QUANT) {
        BvhNode4Q nd = S.qnodes[cur];
        LMC_PIN4(nd.child[0], nd.child[1], nd.child[2], nd.child[3]);
        return VisitNode4Q<ORDERED>(nd, org, invd, tnear, tfar, stk);
    } else {
        BvhNode4 nd = S.nodes[cur];
        LMC_PIN4(nd.child[0], nd.child[1], nd.child[2], nd.child[3]);
        return VisitNode4<ORDERED>(nd, org, invd, tnear, tfar, stk);
    }
}

// closest hit: smallest t in [tnear, tfar]; ties -> lower global triangle id (tree-independent answer).
// "while-while" traversal: all lanes of a wave first descend through inner nodes until each has reached a leaf (or
// finished), then all test their leaf's triangles together.  With the node test and the leaf test as two branches of
// one loop a wave executed both bodies on almost every iteration (profiles/r01_d: 5270 vector-memory instructions per
// wave-step for ~800 per lane).  A leaf's triangles (up to four) are fetched in one round and tested in leaf order.
template <class Stk>
LMC_D int BvhIntersect(const DScene &S, V3 org, V3 dir, float tnear, float tfar, float &tHit, Stk &stk, int hint = -1, bool hintOnly = false) {  // hintOnly: measurement switch LMC_EXP_NOTRAV (dstep_params.h)
    if (S.numNodes == 0) return -1;
    V3 invd{1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z};
    stk.Reset();
    int best = -1;
    float bestT = tfar;
#ifndef LMC_NO_TRI_HINT  // (A/B switch of the build; profiles/r05_ac_ab_triangle_hint.jsonl)
    // hint: the triangle the re-traced state's vertex lies on.  A small step moves a path by little, so the perturbed ray mostly meets the same
    // triangle: tested first, its distance bounds the walk from the root on (lean kernel alone 1.50 -> 1.43 ms).  The answer stays the walk's own --
    // smallest t, ties to the lowest id: the hinted triangle is met again in its leaf and changes nothing.
    if (hint >= 0) {
        const TriData &T = S.tris[hint];
        float p0[3] = {T.p0[0], T.p0[1], T.p0[2]}, e1[3] = {T.e1[0], T.e1[1], T.e1[2]}, e2[3] = {T.e2[0], T.e2[1], T.e2[2]};
        LMC_PIN3(p0[0], p0[1], p0[2]);  // one round of loads (below: LMC_PIN)
        float t;
        if (TriTest(p0, e1, e2, org, dir, tnear, tfar, t)) best = hint, bestT = t;
    }
#endif
    if (hintOnly) {
        tHit = bestT;
        return best;
    }
    int cur = 0;  // root is an inner node
#if defined(LMC_TRAV_SPEC) && defined(__HIP_DEVICE_COMPILE__)
    // A/B build ("speculative" while-while): the wave leaves the inner-node loop as soon as fewer than LMC_TRAV_SPEC of its lanes
    // still hold an inner node -- the lanes waiting at their leaves test them while the stragglers keep their node for the next round -- instead of
    // when none does.  Same visits per ray, same answer; what changes is how many lanes idle in which loop.  cur == BVH4_EMPTY: this lane's walk is over.
    for (;;) {
        for (;;) {
            if (cur >= 0 && cur != BVH4_EMPTY) {
                cur = VisitInner<true>(S, cur, org, invd, tnear, bestT, stk);
                if (cur == BVH4_EMPTY && !stk.Empty()) cur = stk.Pop();
            }
            if (__popcll(__ballot(cur >= 0 && cur != BVH4_EMPTY)) < LMC_TRAV_SPEC) break;
        }
        if (cur < 0) {
            const unsigned code = (unsigned)~cur;
            const int first = (int)(code >> 3), cnt = (int)(code & 7u) + 1;
            for (int base = 0; base < cnt; base += 4) {
                LeafTri tr[4];
#pragma unroll
                for (int i = 0; i < 4; i++) tr[i] = S.leafTris[first + min(base + i, cnt - 1)];
#pragma unroll
                for (int i = 0; i < 4; i++) LMC_PIN3(tr[i].p0[0], tr[i].p0[1], tr[i].p0[2]);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    float t;
                    if (base + i < cnt && TriTest(tr[i].p0, tr[i].e1, tr[i].e2, org, dir, tnear, bestT, t)) {
                        if (best < 0 || t < bestT || (t == bestT && tr[i].id < best)) {
                            bestT = t;
                            best = tr[i].id;
                        }
                    }
                }
            }
            cur = stk.Empty() ? BVH4_EMPTY : stk.Pop();
        }
        if (__ballot(cur != BVH4_EMPTY) == 0ull) break;
    }
    tHit = bestT;
    return best;
#endif
    for (;;) {
        while (cur >= 0) {
            cur = VisitInner<true>(S, cur, org, invd, tnear, bestT, stk);
            if (cur == BVH4_EMPTY) {
                if (stk.Empty()) {
                    tHit = bestT;
                    return best;
                }
                cur = stk.Pop();
            }
        }
        const unsigned code = (unsigned)~cur;
        const int first = (int)(code >> 3), cnt = (int)(code & 7u) + 1;
        for (int base = 0; base < cnt; base += 4) {
            LeafTri tr[4];
#pragma unroll
            for (int i = 0; i < 4; i++) tr[i] = S.leafTris[first + min(base + i, cnt - 1)];
#pragma unroll
            for (int i = 0; i < 4; i++) LMC_PIN3(tr[i].p0[0], tr[i].p0[1], tr[i].p0[2]);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float t;
                if (base + i < cnt && TriTest(tr[i].p0, tr[i].e1, tr[i].e2, org, dir, tnear, bestT, t)) {
                    if (best < 0 || t < bestT || (t == bestT && tr[i].id < best)) {
                        bestT = t;
                        best = tr[i].id;
                    }
                }
            }
        }
        if (stk.Empty()) break;
        cur = stk.Pop();
    }
    tHit = bestT;
    return best;
}

template <class Stk>
LMC_D bool BvhOccluded(const DScene &S, V3 org, V3 dir, float tnear, float tfar, Stk &stk) {
    if (S.numNodes == 0) return false;
    V3 invd{1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z};
    stk.Reset();
    int cur = 0;
    for (;;) {
        while (cur >= 0) {
            cur = VisitInner<false>(S, cur, org, invd, tnear, tfar, stk);
            if (cur == BVH4_EMPTY) {
                if (stk.Empty()) return false;
                cur = stk.Pop();
            }
        }
        const unsigned code = (unsigned)~cur;
        const int first = (int)(code >> 3), cnt = (int)(code & 7u) + 1;
        bool hit = false;
        for (int base = 0; base < cnt; base += 4) {
            LeafTri tr[4];
#pragma unroll
            for (int i = 0; i < 4; i++) tr[i] = S.leafTris[first + min(base + i, cnt - 1)];
#pragma unroll
            for (int i = 0; i < 4; i++) LMC_PIN3(tr[i].p0[0], tr[i].p0[1], tr[i].p0[2]);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float t;
                if (base + i < cnt && TriTest(tr[i].p0, tr[i].e1, tr[i].e2, org, dir, tnear, tfar, t)) hit = true;
            }
        }
        if (hit) return true;
        if (stk.Empty()) return false;
        cur = stk.Pop();
    }
}

// scene.cpp:128-149
template <class Stk>
LMC_D bool Occluded(const DScene &S, V3 org, V3 dir, float dist, Stk &stk) {
    float maxT = (dist == INFINITY) ? INFINITY : (1.0f - c_ShadowEpsilon) * dist;
    return BvhOccluded(S, org, dir, c_IsectEpsilon, maxT, stk);
}

}  // namespace lmcd
