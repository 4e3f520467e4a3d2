// The streamed re-trace: PerturbPathBidir (/root/reference/src/path.cpp:1953-2160) with the path never held as a whole -- vertex by vertex from
// the chain's current SoA path buffer in HBM through registers into the chain's other buffer -- for the launches beside the lean kernel that
// re-trace a state per lane (the H2MC pipeline's k_h2_perturb, step_h2_phases.hip).  It is the walk of dsmall.h SmallStepLean (which keeps its own
// copy: that body is tuned register by register and also collects the new primary-sample vector on the way) and, like it, the same arithmetic in the
// same RNG order as the generic dpath.h PerturbPathBidir, which stays the form of `uselightcoordinatesampling` renders and of trees too deep for
// the LDS traversal stack.  Round 4's k_h2_perturb held the path as a private-memory DPath (1.2 KB per lane) from LoadPath to StorePath and
// H2Serialize: 5.5 GB of HBM traffic per launch at 2^20 chains on the veach-door scene (profiles/r04_final_h2mc_pmc_door.json).
//
// Also here: Serialize(scene, path) (path.cpp:2497-2586) reading the path through an accessor, so that a state's record for the derivative launches
// (dh2coop.h) is written straight from the SoA buffer the walk has just filled.
#pragma once
#include "dgrad.h"
#include "dsmall.h"

namespace lmcd {

// the proposal offsets of the step in the caller's LDS words [first, first + dim), consumed in PerturbPathBidir's order
struct LdsOffsets {
    const float *base;  // &lds[threadIdx.x]
    int stride, w;
    LMC_D float Pop() { return base[(w++) * stride]; }
};

// Re-traces chain i's current state (technique (c, l), record `cur`) with the offsets `off` into `prop`; the contribution of the re-traced path in
// `pc`.  Returns PerturbPathBidir's result, the shadow ray of the connection strategy included (cast after the walk: one any-hit site).
// On success `prop` holds the proposal as a sub-path (ToSubpath, path.cpp:1660-1669: counts, depths, envPrim); the lens words are not part of a
// re-trace (they are written by the large step into both buffers' head only through StorePath and read by nobody but Serialize's three pad words).
template <class Stk>
LMC_D bool PerturbPathStreamed(const DScene &S, const float *cur, float *prop, size_t N, int i, int c, int l, LdsOffsets off, Rng &rng, Stk &stk, Contrib &pc) {
    const int camCount = max(c - 1, 0), lgtCount = max(l - 1, 0);
    DVertex nextV = LoadVertex(cur, N, i, l > 1, 0);
    bool ok = false;
    DeferOcclusion occ;
    NormalDist normDist(0.0f, S.opt.discreteStdDev);
    const float time = Modulo1(LdS(&cur[(size_t)PW_TIME * N + i]) + normDist(rng));
    StS(&prop[(size_t)PW_TIME * N + i], time);
    StS(&prop[(size_t)PW_CAMDEPTH * N + i], __int_as_float(c)), StS(&prop[(size_t)PW_LGTDEPTH * N + i], __int_as_float(l));
    StS(&prop[(size_t)PW_CAMCOUNT * N + i], __int_as_float(camCount)), StS(&prop[(size_t)PW_LGTCOUNT * N + i], __int_as_float(lgtCount));
    StS(&prop[(size_t)PW_LENS0 * N + i], LdS(&cur[(size_t)PW_LENS0 * N + i])), StS(&prop[(size_t)PW_LENS1 * N + i], LdS(&cur[(size_t)PW_LENS1 * N + i]));
    int envPrim = (l == 0) ? __float_as_int(LdS(&cur[(size_t)PW_ENVPRIM * N + i])) : -1;
    BPS lps, cps;
    DVertex lastLgt;
    lastLgt.tri = -1;
    V3 org, dir;
    float tnear = c_IsectEpsilon, tfar = INFINITY;
    V2 screenPos{0.f, 0.f};
    int lgtLight = -1;
    bool lightPhase = false;
    auto BeginCamera = [&]() {  // EmitFromCamera with the perturbed screen position, path.cpp:2032-2038
        const float screen0 = Modulo1(LdS(&cur[(size_t)PW_SCREEN0 * N + i]) + off.Pop());
        const float screen1 = Modulo1(LdS(&cur[(size_t)PW_SCREEN1 * N + i]) + off.Pop());
        StS(&prop[(size_t)PW_SCREEN0 * N + i], screen0), StS(&prop[(size_t)PW_SCREEN1 * N + i], screen1);
        screenPos = V2{screen0, screen1};
        EmitFromCamera(S, screenPos, org, dir, cps);
        tnear = PrimaryMinT(S, screenPos, tfar);
        lightPhase = false;
    };
    if (l > 1) {
        lightPhase = true;
        lgtLight = __float_as_int(LdS(&cur[(size_t)PW_LGTLIGHT * N + i]));
        const float lightPickProb = PickLightProb(S, lgtLight);
        DPath hd;  // only the emitter fields are used by EmitFromLight
        hd.lgtPos0 = Modulo1(LdS(&cur[(size_t)PW_LGTPOS0 * N + i]) + off.Pop());
        hd.lgtPos1 = Modulo1(LdS(&cur[(size_t)PW_LGTPOS1 * N + i]) + off.Pop());
        hd.lgtDir0 = Modulo1(LdS(&cur[(size_t)PW_LGTDIR0 * N + i]) + off.Pop());
        hd.lgtDir1 = Modulo1(LdS(&cur[(size_t)PW_LGTDIR1 * N + i]) + off.Pop());
        hd.lgtLight = lgtLight;
        hd.lgtPrim = __float_as_int(LdS(&cur[(size_t)PW_LGTPRIM * N + i]));
        EmitFromLight(S, lightPickProb, hd, org, dir, lps);
        StS(&prop[(size_t)PW_LGTPOS0 * N + i], hd.lgtPos0), StS(&prop[(size_t)PW_LGTPOS1 * N + i], hd.lgtPos1);
        StS(&prop[(size_t)PW_LGTDIR0 * N + i], hd.lgtDir0), StS(&prop[(size_t)PW_LGTDIR1 * N + i], hd.lgtDir1);
        StS(&prop[(size_t)PW_LGTLIGHT * N + i], __int_as_float(lgtLight)), StS(&prop[(size_t)PW_LGTPRIM * N + i], __int_as_float(hd.lgtPrim));
    } else {
        // a state without a light sub-path keeps the emitter words of its record (Serialize does not read them; the reference copies the whole Path)
        BeginCamera();
    }
    int depth = 0;  // vertex index inside the current sub-path
    // every iteration = one path segment; `break` = the step's contribution is decided (ok) or the path died
    while (lightPhase || depth < camCount) {
        DVertex sv = nextV;
        {  // the record of the vertex the next iteration perturbs is requested now: its round trip runs behind this segment's traversal
            bool nl = lightPhase;
            int nd = depth + 1;
            if (lightPhase && depth == lgtCount - 1) nl = false, nd = 0;
            if (nl ? nd < lgtCount : nd < camCount) nextV = LoadVertex(cur, N, i, nl, nd);
        }
        SurfHit hit;
        hit.tri = -1;
        hit.st = V2{0.f, 0.f};
        Isect isect;
        isect.position = isect.shadingNormal = isect.geomNormal = V3{0.f, 0.f, 0.f};
        const bool hitSurface = IntersectSurface(S, org, dir, tnear, tfar, hit, isect, stk, sv.tri);  // the current state's triangle first (dscene.h)
        if (lightPhase) {
            if (!hitSurface) break;
            lps.isect = isect;
            sv.tri = hit.tri, sv.st0 = hit.st.x, sv.st1 = hit.st.y;
            lps.wi = -dir;
            sv.bsdfDiscrete = Modulo1(sv.bsdfDiscrete + normDist(rng));
            ConvertMIS(S, depth, lgtLight, org, dir, lps);
            if (depth == lgtCount - 1 && c == 1) {
                ok = ConnectToCamera(S, depth, lps, sv, pc, stk, occ);
                StoreVertex(prop, N, i, true, depth, sv);
                break;
            }
            if (depth == lgtCount - 1) {
                StoreVertex(prop, N, i, true, depth, sv);
                lastLgt = sv;
                BeginCamera();
                depth = 0;
                continue;
            }
            sv.rnd0 = Modulo1(sv.rnd0 + off.Pop());
            sv.rnd1 = Modulo1(sv.rnd1 + off.Pop());
            V3 bsdfContrib;
            if (!MAT_BSDF(true, true)(S, MAT_ARG lps, sv, lps, dir, bsdfContrib)) break;
            StoreVertex(prop, N, i, true, depth, sv);
            lps.throughput = lps.throughput * sv.rrWeight;
            org = lps.isect.position;
            depth++;
            continue;
        }
        if (hitSurface) cps.isect = isect;
        sv.tri = hit.tri, sv.st0 = hit.st.x, sv.st1 = hit.st.y;
        cps.wi = -dir;
        if (hitSurface) ConvertMIS(S, depth, -1, org, dir, cps);
        if (depth == camCount - 1 && l == 0) {
            const int light = HitLightOf(S, hitSurface, hit);
            if (light >= 0) ok = HandleHitLight(S, depth, light, hitSurface, dir, screenPos, cps, envPrim, pc);
            StoreVertex(prop, N, i, false, depth, sv);
            break;
        }
        if (!hitSurface) break;
        sv.bsdfDiscrete = Modulo1(sv.bsdfDiscrete + normDist(rng));
        if (depth == camCount - 1) {
            if (l == 1) {
                const float directLightPickProb = PickLightProb(S, sv.dirLight);
                sv.dirRnd0 = Modulo1(sv.dirRnd0 + off.Pop());
                sv.dirRnd1 = Modulo1(sv.dirRnd1 + off.Pop());
                ok = MAT_DIRECT(S, depth, cps, screenPos, directLightPickProb, sv, pc, stk, occ);
            } else {
                ok = ConnectVertex(S, depth, lgtCount - 1, lps, lastLgt, cps, sv, screenPos, pc, stk, occ);
            }
            StoreVertex(prop, N, i, false, depth, sv);
            break;
        }
        sv.rnd0 = Modulo1(sv.rnd0 + off.Pop());
        sv.rnd1 = Modulo1(sv.rnd1 + off.Pop());
        V3 bsdfContrib;
        if (!MAT_BSDF(false, true)(S, MAT_ARG cps, sv, cps, dir, bsdfContrib)) break;
        StoreVertex(prop, N, i, false, depth, sv);
        cps.throughput = cps.throughput * sv.rrWeight;
        org = cps.isect.position;
        tnear = c_IsectEpsilon;
        tfar = INFINITY;
        depth++;
    }
    StS(&prop[(size_t)PW_ENVPRIM * N + i], __int_as_float(envPrim));
    // the one shadow ray of the step (scene.cpp:128-149), cast after its strategy has been evaluated
    if (ok && occ.pending) ok = !Occluded(S, occ.org, occ.dir, occ.dist, stk);
    return ok;
}

// ---- Serialize(scene, path), path.cpp:2497-2586, over an SoA path record (same words, same order as dgrad.h SerializePath over a DPath)
struct SoAPathView {
    const float *buf;
    size_t N;
    int i;
    LMC_D float HeadF(int w) const { return buf[(size_t)w * N + i]; }
    LMC_D int HeadI(int w) const { return __float_as_int(buf[(size_t)w * N + i]); }
    LMC_D DVertex Vert(bool lgt, int d) const { return LoadVertex(buf, N, i, lgt, d); }
};

// rec: the chain's AoS record (dh2coop.h): [0, 2L + 1) primary | [H2_REC_C] c | [H2_REC_L] l | [H2_REC_VP ..) vertParams.  Returns the state's material
// signature (dpipe.h H2MaterialSignature: same hash, gathered on the way)
LMC_D unsigned H2SerializeStreamed(const DScene &S, const SoAPathView P, float *rec) {
    const int camDepth = P.HeadI(PW_CAMDEPTH), lgtDepth = P.HeadI(PW_LGTDEPTH), camCount = P.HeadI(PW_CAMCOUNT), lgtCount = P.HeadI(PW_LGTCOUNT);
    StridedOut o{rec + H2_REC_VP, 1, 0};
    unsigned sig = 0;
    auto Sig = [&](const DVertex &v) {
#ifndef LMC_H2_NOSIG
        if (v.tri >= 0) sig = sig * 7u + (unsigned)MaterialOfTri(S, v.tri).type + (v.useAbs != 0.0f ? 3u : 0u) + 1u;
#endif
    };
    int pi = 0;
    // the head words in one round of loads (a store to the record between two loads keeps the second from being issued: dgrad.h)
    float hTime = P.HeadF(PW_TIME), hLens0 = P.HeadF(PW_LENS0), hLens1 = P.HeadF(PW_LENS1), hScreen0 = P.HeadF(PW_SCREEN0), hScreen1 = P.HeadF(PW_SCREEN1);
    LMC_PIN5(hTime, hLens0, hLens1, hScreen0, hScreen1);
    rec[pi++] = hTime;
    rec[H2_REC_C] = __int_as_float(camDepth), rec[H2_REC_L] = __int_as_float(lgtDepth);
    o.Put(hLens0), o.Put(hLens1), o.Put(0.f);
    if (lgtDepth > 1) {
        const int lgtLight = P.HeadI(PW_LGTLIGHT);
        float lp0 = P.HeadF(PW_LGTPOS0), lp1 = P.HeadF(PW_LGTPOS1), ld0 = P.HeadF(PW_LGTDIR0), ld1 = P.HeadF(PW_LGTDIR1);
        LMC_PIN4(lp0, lp1, ld0, ld1);
        rec[pi++] = lp0, rec[pi++] = lp1, rec[pi++] = ld0, rec[pi++] = ld1;
        o.Put(PickLightProb(S, lgtLight));
        SerializeLight(S, lgtLight, P.HeadI(PW_LGTPRIM), o);
        for (int d = 0; d < lgtCount; d++) {
            const DVertex v = P.Vert(true, d);
            Sig(v);
            SerializeTri(S, v.tri, o);
            o.Put(v.bsdfDiscrete), o.Put(v.useAbs);
            SerializeBSDF(S, v.tri, V2{v.st0, v.st1}, o);
            if (d == lgtCount - 1 && camDepth == 1) return sig;
            if (d == lgtCount - 1) break;
            rec[pi++] = v.rnd0, rec[pi++] = v.rnd1;
            o.Put(v.rrWeight);
        }
    }
    rec[pi++] = hScreen0, rec[pi++] = hScreen1;
    for (int d = 0; d < camCount; d++) {
        const DVertex v = P.Vert(false, d);
        Sig(v);
        if (v.tri >= 0) {
            SerializeTri(S, v.tri, o);
        } else {  // escaped to the environment: a fixed non-degenerate triangle (dgrad.h SerializePath)
            const float dummy[46] = {0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 1};
            for (int k = 0; k < 46; k++) o.Put(dummy[k]);
        }
        if (d == camCount - 1) {
            if (lgtDepth == 0) {
                if (v.tri < 0) {
                    SerializeLight(S, S.envLight, P.HeadI(PW_ENVPRIM), o);
                    o.Put(PickLightProb(S, S.envLight));
                } else {
                    const int al = S.meshes[S.tris[v.tri].mesh].areaLight;
                    SerializeLight(S, al, v.tri - S.meshes[S.tris[v.tri].mesh].triBase, o);
                    o.Put(PickLightProb(S, al));
                }
            } else if (lgtDepth == 1) {
                rec[pi++] = v.dirRnd0, rec[pi++] = v.dirRnd1;
                SerializeLight(S, v.dirLight, v.dirPrim, o);
                SerializeBSDF(S, v.tri, V2{v.st0, v.st1}, o);
                o.Put(PickLightProb(S, v.dirLight));
            } else {
                SerializeBSDF(S, v.tri, V2{v.st0, v.st1}, o);
            }
            return sig;
        }
        rec[pi++] = v.rnd0, rec[pi++] = v.rnd1;
        o.Put(v.bsdfDiscrete), o.Put(v.useAbs);
        SerializeBSDF(S, v.tri, V2{v.st0, v.st1}, o);
        o.Put(v.rrWeight);
    }
    return sig;
}

}  // namespace lmcd
