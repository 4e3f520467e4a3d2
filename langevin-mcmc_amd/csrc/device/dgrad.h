// In-kernel gradient hook of the MALA step: serialises the path into the reference's path-function ABI
// layout (path.cpp:2497-2586) inside a per-thread slice of an SoA work buffer in HBM and differentiates the
// path program (pathfunc.h) in forward mode.
#pragma once
#include "dpath.h"
#include "pathfunc.h"

namespace lmcd {

// per-thread slice of the serialisation buffer: word w of slot s lives at buf[w * stride + s]
struct GradWork {
    float *buf;
    size_t stride, slot;
};

// derivative programs exist for 1 <= c <= 9, 0 <= l <= 8, 3 <= c + l <= 9 (path.cpp:3955-3959, --max-derivatives-depth 8)
LMC_HD bool GradAvailable(int c, int l) { return c >= 1 && l >= 0 && c + l >= 3 && c + l - 1 <= 8; }

// Serialize(scene, path, ssubPath), path.cpp:2497-2586, into the strided work buffer
struct StridedOut {
    float *p;
    size_t stride;
    int n;
    LMC_D void Put(float v) {
        p[(size_t)n * stride] = v;
        n++;
    }
    LMC_D void Skip(int k) {
        for (int i = 0; i < k; i++) Put(0.f);
    }
};

// The serialisers copy scene records into a state's record word by word.  Through references every `o.Put(T.p0[k])` was "load, wait, store" --
// the store may alias the next load for all the compiler knows, so the next load was not even issued before it: ~450 dependent round trips
// per serialised state (scripts/isa_load_waits.py on k_h2_perturb_streamed: 36 inlined copies x 46 words, every load waited for on its own).
// The records are therefore copied by value first, their loads pinned together (dscene.h LMC_PIN), and written out afterwards.
LMC_D void SerializeTri(const DScene &S, int tri, StridedOut &o) {  // trianglemesh.cpp:145-187
    TriData T = S.tris[tri];
    LMC_PIN4(T.p0[0], T.e1[1], T.e2[2], T.n0[0]);
    LMC_PIN4(T.n1[1], T.n2[2], T.st[0], T.st[3]);
    LMC_PIN2(T.st[5], T.mesh);
    DMesh M = S.meshes[T.mesh];
    LMC_PIN2(M.hasST, M.invTotalArea);
    o.Put(0.f);  // ShapeType::TriangleMesh
    o.Put(0.f);  // isMoving
    for (int rep = 0; rep < 2; rep++) {
        for (int k = 0; k < 3; k++) o.Put(T.p0[k]);
        for (int k = 0; k < 3; k++) o.Put(T.e1[k]);
        for (int k = 0; k < 3; k++) o.Put(T.e2[k]);
        for (int k = 0; k < 3; k++) o.Put(T.n0[k]);
        for (int k = 0; k < 3; k++) o.Put(T.n1[k]);
        for (int k = 0; k < 3; k++) o.Put(T.n2[k]);
    }
    o.Put(M.hasST ? 0.f : 1.f);
    if (M.hasST)
        for (int k = 0; k < 6; k++) o.Put(T.st[k]);
    else
        o.Skip(6);
    o.Put(M.invTotalArea);
}

LMC_D void SerializeBSDF(const DScene &S, int tri, V2 st, StridedOut &o) {  // bsdf.cpp:7-11 (10-float slot)
    const DMaterial m = LoadMaterial<true>(S, tri);  // by value, one round of loads (dshade.h)
    int start = o.n;
    o.Put((float)m.type);
    if (m.type == BSDF_LAMBERTIAN) {
        V3 kd = EvalKd(S, m, st);
        o.Put(kd.x), o.Put(kd.y), o.Put(kd.z);
    } else if (m.type == BSDF_PHONG) {  // phong.cpp:14-20
        V3 kd = EvalKd(S, m, st), ks = EvalTex(S, m.Ks, st);
        o.Put(kd.x), o.Put(kd.y), o.Put(kd.z);
        o.Put(ks.x), o.Put(ks.y), o.Put(ks.z);
        o.Put(EvalTex(S, m.expOrAlpha, st).x);
        o.Put(m.KsWeight);
    } else {  // roughdielectric.cpp:13-20
        V3 ks = EvalTex(S, m.Ks, st), kt = EvalTex(S, m.Kt, st);
        o.Put(ks.x), o.Put(ks.y), o.Put(ks.z);
        o.Put(kt.x), o.Put(kt.y), o.Put(kt.z);
        o.Put(m.eta), o.Put(m.invEta), o.Put(EvalTex(S, m.expOrAlpha, st).x);
    }
    o.Skip(10 - (o.n - start));
}

LMC_D void SerializeLight(const DScene &S, int light, int lPrimID, StridedOut &o) {  // 56-float slot (light.cpp:7-10)
    DLight L = S.lights[light];
    LMC_PIN4(L.type, L.pos[0], L.pos[2], L.intensity[1]);
    LMC_PIN4(L.intensity[2], L.mesh, L.radiance[0], L.radiance[2]);
    int start = o.n;
    o.Put((float)L.type);
    if (L.type == LIGHT_POINT) {  // pointlight.cpp:14-18
        for (int k = 0; k < 3; k++) o.Put(L.pos[k]);
        for (int k = 0; k < 3; k++) o.Put(L.intensity[k]);
    } else if (L.type == LIGHT_AREA) {  // arealight.cpp:17-22
        SerializeTri(S, S.meshes[L.mesh].triBase + lPrimID, o);
        for (int k = 0; k < 3; k++) o.Put(L.radiance[k]);
    } else {  // envlight.cpp:65-118
        const DEnv &E = S.env;
        for (int k = 0; k < 30; k++) o.Put(E.xformBlocks[k]);
        int col = lPrimID % E.W, row = lPrimID / E.W;
        const float *cdfCol = E.cdfCols + (long)row * (E.W + 1);
        // every look-up first, pinned together, then the stores
        float cc0 = cdfCol[col], cc1 = cdfCol[col + 1], cr0 = E.cdfRows[row], cr1 = E.cdfRows[row + 1];
        V3 t00 = EnvRepAt(E, col, row), t10 = EnvRepAt(E, col + 1, row), t01 = EnvRepAt(E, col, row + 1), t11 = EnvRepAt(E, col + 1, row + 1);
        float rw0 = E.rowWeights[Clampi(row, 0, E.H - 1)], rw1 = E.rowWeights[Clampi(row + 1, 0, E.H - 1)];
        LMC_PIN4(cc0, cc1, cr0, cr1);
        LMC_PIN3(t00.x, t00.y, t00.z);
        LMC_PIN3(t10.x, t10.y, t10.z);
        LMC_PIN3(t01.x, t01.y, t01.z);
        LMC_PIN3(t11.x, t11.y, t11.z);
        LMC_PIN2(rw0, rw1);
        o.Put(cc0), o.Put(cc1);
        o.Put(cr0), o.Put(cr1);
        o.Put((float)col), o.Put((float)row);
        o.Put(E.pixelSize[0]), o.Put(E.pixelSize[1]);
        o.Put(t00.x), o.Put(t00.y), o.Put(t00.z), o.Put(t10.x), o.Put(t10.y), o.Put(t10.z);
        o.Put(t01.x), o.Put(t01.y), o.Put(t01.z), o.Put(t11.x), o.Put(t11.y), o.Put(t11.z);
        o.Put(rw0);
        o.Put(rw1);
        o.Put(E.normalization);
    }
    o.Skip(56 - (o.n - start));
}

// returns the number of vertParams words written; primary[0..2L] filled
LMC_D int SerializePath(const DScene &S, const DPath &path, float *primary, StridedOut &o) {
    int pi = 0;
    primary[pi++] = path.time;
    o.Put(path.lensPos0), o.Put(path.lensPos1), o.Put(0.f);  // lensVertexPos: unused by the static MALA programs
    if (path.lgtDepth > 1) {
        primary[pi++] = path.lgtPos0, primary[pi++] = path.lgtPos1, primary[pi++] = path.lgtDir0, primary[pi++] = path.lgtDir1;
        o.Put(PickLightProb(S, path.lgtLight));
        SerializeLight(S, path.lgtLight, path.lgtPrim, o);
        for (int d = 0; d < path.lgtCount; d++) {
            const DVertex &v = path.lgt[d];
            SerializeTri(S, v.tri, o);
            o.Put(v.bsdfDiscrete), o.Put(v.useAbs);
            SerializeBSDF(S, v.tri, V2{v.st0, v.st1}, o);
            if (d == path.lgtCount - 1 && path.camDepth == 1) return o.n;
            if (d == path.lgtCount - 1) break;
            primary[pi++] = v.rnd0, primary[pi++] = v.rnd1;
            o.Put(v.rrWeight);
        }
    }
    primary[pi++] = path.screen0, primary[pi++] = path.screen1;
    for (int d = 0; d < path.camCount; d++) {
        const DVertex &v = path.cam[d];
        if (v.tri >= 0) {
            SerializeTri(S, v.tri, o);
        } else {
            // escaped to the environment: the reference leaves a stale slot here (path.cpp:2546-2549); its value does
            // not influence the result as long as it is a non-degenerate triangle -- write a fixed unit one
            const float dummy[46] = {0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 1};
            for (int k = 0; k < 46; k++) o.Put(dummy[k]);
        }
        if (d == path.camCount - 1) {
            if (path.lgtDepth == 0) {
                if (v.tri < 0) {  // escaped: environment light
                    SerializeLight(S, S.envLight, path.envPrim, o);
                    o.Put(PickLightProb(S, S.envLight));
                } else {
                    const int al = S.meshes[S.tris[v.tri].mesh].areaLight;
                    SerializeLight(S, al, v.tri - S.meshes[S.tris[v.tri].mesh].triBase, o);
                    o.Put(PickLightProb(S, al));
                }
            } else if (path.lgtDepth == 1) {
                primary[pi++] = v.dirRnd0, primary[pi++] = v.dirRnd1;
                SerializeLight(S, v.dirLight, v.dirPrim, o);
                SerializeBSDF(S, v.tri, V2{v.st0, v.st1}, o);
                o.Put(PickLightProb(S, v.dirLight));
            } else {
                SerializeBSDF(S, v.tri, V2{v.st0, v.st1}, o);
            }
            return o.n;
        }
        primary[pi++] = v.rnd0, primary[pi++] = v.rnd1;
        o.Put(v.bsdfDiscrete), o.Put(v.useAbs);
        SerializeBSDF(S, v.tri, V2{v.st0, v.st1}, o);
        o.Put(v.rrWeight);
    }
    return o.n;
}

// d log ssScore / d pss through the path program (the reference calls evaluate_path_bidir_mala_<c>_<l>_static_derv,
// mutation_mala.h:101-107)
LMC_D void ComputeGradient(const DScene &S, const DPath &path, const Contrib &sp, float *grad, GradWork &gw) {
    float primary[2 * MAXD + 1];
    StridedOut o{gw.buf + gw.slot, gw.stride, 0};
    SerializePath(S, path, primary, o);
    StridedIn vin{gw.buf + gw.slot, gw.stride};
    float logLum;
    PathFuncGradUpTo12(path.camDepth, path.lgtDepth, primary, S.sceneParams, vin, &logLum, grad);
    (void)sp;
}

}  // namespace lmcd
