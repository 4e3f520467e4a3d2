// GeneratePathBidir (/root/reference/src/path.cpp:1237-1449) with the vertex CONNECTIONS of a wave shared out over its 64 lanes.
//
// Why: in the large-step launch one lane generates a whole bidirectional path and, at every camera vertex, connects it to every vertex of its own light
// sub-path (ConnectVertex: a shadow ray + two BSDF evaluations each).  A lane with 7 camera and 7 light vertices runs ~30 connections, its neighbour whose
// path died at the second vertex none: the wave runs the longest lane's loop with a sixth to a quarter of its lanes active (profiles/lanes_by_workload.json:
// large steps 17 % torus, 10 % full-material torus, 25 % door) -- and on the scenes with real light sub-paths the large-step launch is as long as the lean one.
// Here the camera loop is wave-uniform (a lane whose path has ended stays in it, idle for the per-lane work) and at every camera depth the connections of ALL
// lanes form one task list in LDS that the 64 lanes work off together: a task = (owner lane, light depth); the worker reads the owner's camera state and that
// light state from a per-slot scratch in HBM (written by the owner, fenced), evaluates the SAME ConnectVertex on the SAME operands, and writes the result to a
// per-slot scratch from which the owner pushes the valid ones in light-depth order -- the sink sees exactly the sequence the reference's loop produces, the
// random numbers are drawn by the owner in the reference's order (connections draw none).  Same contributions bit for bit; tests: the chain-parity tests run
// the launch both ways (tests/test_gpu_parity.py, test_gpu_door.py: LMC_LARGE_COOP).
#pragma once
#include "dstep.h"

namespace lmcd {

constexpr int COOP_STATE_WORDS = 21;  // BPS (18) + the vertex's tri, st0, st1
constexpr int COOP_CAM_WORDS = COOP_STATE_WORDS + 2;  // + screenPos
constexpr int COOP_RES_WORDS = 10;    // valid flag + Contrib (9)
constexpr int COOP_MAX_TASKS = 64 * MAXD;

struct CoopScratch {
    float *lgt;  // [(depth * COOP_STATE_WORDS + w) * N + slot]: the light states of the slot's large step
    float *cam;  // [w * N + slot]: the camera state at the depth being connected
    float *res;  // [(lgtDepth * COOP_RES_WORDS + w) * N + slot]: results of the depth's connections
    size_t N;
};

LMC_D void CoopStoreState(float *p, size_t N, const BPS &s, const DVertex &v) {  // p = first word of the slot
    p[0 * N] = s.isect.position.x, p[1 * N] = s.isect.position.y, p[2 * N] = s.isect.position.z;
    p[3 * N] = s.isect.shadingNormal.x, p[4 * N] = s.isect.shadingNormal.y, p[5 * N] = s.isect.shadingNormal.z;
    p[6 * N] = s.isect.geomNormal.x, p[7 * N] = s.isect.geomNormal.y, p[8 * N] = s.isect.geomNormal.z;
    p[9 * N] = s.wi.x, p[10 * N] = s.wi.y, p[11 * N] = s.wi.z;
    p[12 * N] = s.accMISWPrev, p[13 * N] = s.accMISWThis;
    p[14 * N] = s.throughput.x, p[15 * N] = s.throughput.y, p[16 * N] = s.throughput.z;
    p[17 * N] = s.ssJacobian;
    p[18 * N] = __int_as_float(v.tri), p[19 * N] = v.st0, p[20 * N] = v.st1;
}
// The scratch is written by one lane and read by another lane of the same wave: the reads go to L2 (device-scope relaxed atomic loads: the vector L1 is
// bypassed, not invalidated -- an acquire fence per camera depth threw the scene's nodes and triangles out of L1 with it: door large steps 3.5 -> 5.9 ms),
// the writes are complete before them (CoopFence: a release at workgroup scope = wait for the wave's outstanding stores; the L1 is write-through)
LMC_D float CoopLd(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
LMC_D void CoopLoadState(const float *p, size_t N, BPS &s, DVertex &v) {
    float w[COOP_STATE_WORDS];
#pragma unroll
    for (int k = 0; k < COOP_STATE_WORDS; k++) w[k] = CoopLd(p + (size_t)k * N);
    s.isect.position = V3{w[0], w[1], w[2]};
    s.isect.shadingNormal = V3{w[3], w[4], w[5]};
    s.isect.geomNormal = V3{w[6], w[7], w[8]};
    s.wi = V3{w[9], w[10], w[11]};
    s.accMISWPrev = w[12], s.accMISWThis = w[13];
    s.throughput = V3{w[14], w[15], w[16]};
    s.ssJacobian = w[17];
    v.tri = __float_as_int(w[18]), v.st0 = w[19], v.st1 = w[20];
}
LMC_D void CoopFence() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }

// One wave (a 64-thread block); `active`: this lane generates a path (lanes past the end of the work list only work off connection tasks).
// taskList: COOP_MAX_TASKS unsigned shorts of LDS.  slot: the lane's chain slot (its index into the scratch), any valid slot for an inactive lane.
template <class Stk>
LMC_D void GeneratePathBidirCoop(const DScene &S, int minDepth, int maxDepth, DPath &path, ContribSink &sink, Rng &rng, Stk &stk, bool active, int slot, const CoopScratch &X,
                                 unsigned short *taskList) {
    TraceOcclusion trace;
    const int lane = threadIdx.x;
    const size_t N = X.N;
    path.camCount = path.lgtCount = 0;
    path.envPrim = -1;
    BPS lightStates[MAXD];
    int numLightStates = 0;
    V3 org{0.f, 0.f, 0.f}, dir{0.f, 0.f, 1.f};
    if (active) {  // ---- the light sub-path, as in GeneratePathBidir (dpath.h), lane by lane
        path.time = rng.Uniform();
        numLightStates = 1;
        float lightPickProb = 1.0f;
        {  // EmitFromLightInit, path.cpp:576-586
            V2 p = RndVec2(rng), d = RndVec2(rng);
            path.lgtPos0 = p.x, path.lgtPos1 = p.y, path.lgtDir0 = d.x, path.lgtDir1 = d.y;
            path.lgtLight = PickLight(S, rng.Uniform(), lightPickProb);
            path.lgtPrim = LightSampleDiscrete(S, path.lgtLight, rng.Uniform());
        }
        EmitFromLight(S, lightPickProb, path, org, dir, lightStates[0]);
        for (int lgtDepth = 0;; lgtDepth++) {
            DVertex &sv = path.lgt[lgtDepth];
            SurfHit hit;
            bool hitSurface = IntersectSurface(S, org, dir, c_IsectEpsilon, INFINITY, hit, lightStates[lgtDepth].isect, stk);
            if (!hitSurface) {
                numLightStates--;
                break;
            }
            path.lgtCount = lgtDepth + 1;
            sv.tri = hit.tri, sv.st0 = hit.st.x, sv.st1 = hit.st.y;
            sv.bsdfDiscrete = rng.Uniform();
            lightStates[lgtDepth].wi = -dir;
            ConvertMIS(S, lgtDepth, path.lgtLight, org, dir, lightStates[lgtDepth]);
            if (lgtDepth + 2 >= minDepth) {
                Contrib c;
                if (ConnectToCamera(S, lgtDepth, lightStates[lgtDepth], sv, c, stk, trace)) sink.Push(c);
            }
            if (maxDepth != -1 && lgtDepth + 2 >= maxDepth) break;
            if (lgtDepth + 1 >= MAXD) break;
            numLightStates++;
            V2 r = RndVec2(rng);
            sv.rnd0 = r.x, sv.rnd1 = r.y;
            V3 bsdfContrib;
            if (!BSDFSampling<true, false, Stk::kGlossy>(S, lightStates[lgtDepth], sv, lightStates[lgtDepth + 1], dir, bsdfContrib)) {
                numLightStates--;
                break;
            }
            if (sv.useAbs == 0.0f) lightStates[lgtDepth + 1].ssJacobian = 0.0f;
            if (!RussianRoulette(lgtDepth, bsdfContrib, sv.rrWeight, lightStates[lgtDepth + 1].throughput, rng)) {
                numLightStates--;
                break;
            }
            org = lightStates[lgtDepth].isect.position;
        }
        // the light states the camera vertices will be connected to, where any lane can read them
        for (int d = 0; d < numLightStates; d++) CoopStoreState(X.lgt + (size_t)d * COOP_STATE_WORDS * N + slot, N, lightStates[d], path.lgt[d]);
    }

    // ---- the camera sub-path: wave-uniform loop, connections shared out
    BPS cps;
    V2 screenPos{0.f, 0.f};
    float tnear = c_IsectEpsilon, tfar = INFINITY;
    float lcJac = 0.0f;
    bool alive = active;
    if (active) {
        V2 s = RndVec2(rng);  // EmitFromCameraInit with screenPosi = (-1,-1)
        path.screen0 = s.x, path.screen1 = s.y;
        screenPos = V2{path.screen0, path.screen1};
        EmitFromCamera(S, screenPos, org, dir, cps);
        tnear = PrimaryMinT(S, screenPos, tfar);
    }
    for (int camDepth = 0; camDepth < MAXD; camDepth++) {
        if (__ballot(alive) == 0ull) break;
        int nConn = 0, lgt0 = 0;
        SurfHit hit;
        hit.tri = -1;
        if (alive) {
            DVertex &sv = path.cam[camDepth];
            path.camCount = camDepth + 1;
            bool hitSurface = IntersectSurface(S, org, dir, tnear, tfar, hit, cps.isect, stk);
            sv.tri = hit.tri, sv.st0 = hit.st.x, sv.st1 = hit.st.y;
            cps.wi = -dir;
            if (hitSurface) ConvertMIS(S, camDepth, -1, org, dir, cps);
            if (camDepth + 1 >= minDepth) {
                int light = HitLightOf(S, hitSurface, hit);
                if (light >= 0) {
                    if (S.opt.useLightCoord && camDepth > 1 && S.lights[light].type == LIGHT_AREA) {  // path.cpp:1339-1360
                        DVertex &prev = path.cam[camDepth - 1];
                        const V2 sp = TriangleSampleParam(S, hit.tri, cps.isect.position);
                        prev.rnd0 = sp.x, prev.rnd1 = sp.y;
                        V3 dirToPrev = cps.isect.position - org;
                        const float distSq = LengthSquared(dirToPrev);
                        const float invDistSq = inverse(distSq);
                        const float invDist = sqrtf(invDistSq);
                        dirToPrev = dirToPrev * invDist;
                        cps.ssJacobian *= fabsf(Dot(dirToPrev, cps.isect.shadingNormal) * invDistSq) * (lcJac * S.meshes[S.tris[hit.tri].mesh].invTotalArea);
                    }
                    Contrib c;
                    if (HandleHitLight(S, camDepth, light, hitSurface, dir, screenPos, cps, path.envPrim, c)) sink.Push(c);
                    alive = false;  // `return` in the reference
                }
            }
            if (alive && (!hitSurface || (maxDepth != -1 && camDepth + 1 >= maxDepth))) alive = false;
            if (alive) {
                sv.bsdfDiscrete = rng.Uniform();
                if (camDepth + 2 >= minDepth) {
                    float directLightPickProb = 1.0f;
                    sv.dirLight = PickLight(S, rng.Uniform(), directLightPickProb);  // DirectLightingInit, path.cpp:184-193
                    V2 r = RndVec2(rng);
                    sv.dirRnd0 = r.x, sv.dirRnd1 = r.y;
                    sv.dirPrim = LightSampleDiscrete(S, sv.dirLight, rng.Uniform());
                    Contrib c;
                    if (MAT_DIRECT(S, camDepth, cps, screenPos, directLightPickProb, sv, c, stk, trace)) sink.Push(c);
                }
                const int maxLgtDepth = maxDepth == -1 ? (numLightStates - 1) : min(maxDepth - camDepth - 3, numLightStates - 1);
                lgt0 = max(0, minDepth - camDepth - 3);  // the reference's test `camDepth + lgtDepth + 3 >= minDepth` is monotone in lgtDepth
                nConn = max(0, maxLgtDepth - lgt0 + 1);
                if (nConn > 0) {  // the camera state the connections read, published
                    float *p = X.cam + slot;
                    CoopStoreState(p, N, cps, sv);
                    p[(size_t)COOP_STATE_WORDS * N] = screenPos.x, p[(size_t)(COOP_STATE_WORDS + 1) * N] = screenPos.y;
                }
            }
        }
        // ---- this depth's connections of the whole wave, worked off by all 64 lanes
        int incl = nConn;
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off);
            if (lane >= off) incl += o;
        }
        const int total = __shfl(incl, 63);
        if (total > 0) {
            const int start = incl - nConn;
            for (int k = 0; k < nConn; k++) taskList[start + k] = (unsigned short)((lane << 4) | (lgt0 + k));
            CoopFence();
            __builtin_amdgcn_wave_barrier();
            for (int t0 = 0; t0 < total; t0 += 64) {
                const int t = t0 + lane;
                const unsigned e = taskList[min(t, total - 1)];
                const int owner = (int)(e >> 4), ld = (int)(e & 15u);
                const int ownerSlot = __shfl(slot, owner);
                if (t < total) {
                    BPS lps, ocps;
                    DVertex lv, cv;
                    CoopLoadState(X.lgt + (size_t)ld * COOP_STATE_WORDS * N + ownerSlot, N, lps, lv);
                    const float *pc = X.cam + ownerSlot;
                    CoopLoadState(pc, N, ocps, cv);
                    const V2 osp{CoopLd(pc + (size_t)COOP_STATE_WORDS * N), CoopLd(pc + (size_t)(COOP_STATE_WORDS + 1) * N)};
                    Contrib c;
                    const bool ok = ConnectVertex(S, camDepth, ld, lps, lv, ocps, cv, osp, c, stk, trace);
                    float *r = X.res + (size_t)ld * COOP_RES_WORDS * N + ownerSlot;
                    r[0] = ok ? 1.0f : 0.0f;
                    if (ok) {
                        r[1 * N] = __int_as_float(c.camDepth), r[2 * N] = __int_as_float(c.lightDepth), r[3 * N] = c.screenPos.x, r[4 * N] = c.screenPos.y;
                        r[5 * N] = c.contrib.x, r[6 * N] = c.contrib.y, r[7 * N] = c.contrib.z, r[8 * N] = c.lsScore, r[9 * N] = c.ssScore;
                    }
                }
            }
            CoopFence();
            __builtin_amdgcn_wave_barrier();
            for (int k = 0; k < nConn; k++) {  // the owner pushes its valid results in light-depth order: the reference's sequence
                const float *r = X.res + (size_t)(lgt0 + k) * COOP_RES_WORDS * N + slot;
                if (CoopLd(r) != 0.0f) {
                    Contrib c;
                    c.camDepth = __float_as_int(CoopLd(r + 1 * N)), c.lightDepth = __float_as_int(CoopLd(r + 2 * N));
                    c.screenPos = V2{CoopLd(r + 3 * N), CoopLd(r + 4 * N)};
                    c.contrib = V3{CoopLd(r + 5 * N), CoopLd(r + 6 * N), CoopLd(r + 7 * N)};
                    c.lsScore = CoopLd(r + 8 * N), c.ssScore = CoopLd(r + 9 * N);
                    sink.Push(c);
                }
            }
        }
        if (alive) {
            DVertex &sv = path.cam[camDepth];
            V2 r = RndVec2(rng);
            sv.rnd0 = r.x, sv.rnd1 = r.y;
            V3 bsdfContrib;
            if (!MAT_BSDF(false, false)(S, MAT_ARG cps, sv, cps, dir, bsdfContrib, &lcJac)) alive = false;
            else if (!RussianRoulette(camDepth, bsdfContrib, sv.rrWeight, cps.throughput, rng)) alive = false;
            else {
                org = cps.isect.position;
                tnear = c_IsectEpsilon;
                tfar = INFINITY;
            }
        }
    }
}

}  // namespace lmcd
