// Large steps with the lanes of a wave RE-FILLED: LargeStep::Mutate (mutation_large.h:31-128) over GeneratePathBidir (path.cpp:1237-1449),
// one lane per chain as in k_step<large> (step_kernel.h), but a lane whose path has ended does not idle until the longest path of its wave
// has: the walk is a state machine of path SEGMENTS (one closest-hit traversal + the work at the vertex it finds), light and camera sub-paths
// through the same traversal site, every shadow ray of the connection strategies through ONE any-hit site (DeferOcclusion: the strategies draw
// no random numbers), and the lanes whose path is complete run the step's tail (technique selection, splats, accept / reject, the next step's
// kind) TOGETHER once enough of them wait, then take the next chains of the work list from a cursor shared by the launch.
//
// A chain's arithmetic, its random draws and their order are those of GeneratePathBidir / StepChain (dpath.h, dstep.h): which lane of which wave
// walks a chain, and beside whom, is all that changes -- the chain-exact parity tests of the large-step launch hold for this form as they stand.
// Why: lengths of freshly generated paths differ far more than those of a wave of re-traced states (a camera ray that leaves the scene ends the
// path at its first segment, Russian roulette from the fourth on): 17 % of the lanes of the plain launch were active on the torus, 10-25 % on the
// glossy scenes (profiles/r05_bi_*, r05_bk_*).
#pragma once
#include "dstep.h"

namespace lmcd {

enum : int { LP_NEED = 0, LP_LIGHT = 1, LP_CAM = 2, LP_DONE = 3, LP_EXIT = 4 };

// What StepChain<true, false, false, 0> does behind GeneratePathBidir (dstep.h: selection mutation_large.h:60-112, splats mlt.cpp:103-112,
// accept / reject mlt.cpp:113-170), on a path and a contribution list that are complete.
LMC_D void LargeStepFinish(const DScene &S, const DCache &cache, const ChainArrays &A, const Film &film, const StepParams &P, int i, DPath &prop,
                           const ContribSink &sink, Rng &rng, StepStats &st) {
    const size_t N = A.N;
    int flags = A.flags[i];
    const bool curValid = flags & F_VALID;
    const Contrib cur = LoadContrib(A.curContrib, A.N, i);
    Contrib pc;
    pc.camDepth = pc.lightDepth = 0;
    pc.lsScore = pc.ssScore = 0.f;
    float a = 1.0f;
    float propScoreSum = 0.f;
    if (sink.count > 0) {
        float scoreSum = 0.f;
        for (int k = 0; k < sink.count; k++) scoreSum += sink.LsScore(k);  // contribCdf.back()
        const float invSc = inverse(scoreSum);
        const float u = rng.Uniform();
        int pos = sink.count + 1;  // std::upper_bound(cdf * invSc, u): first element > u; contribId = clamp(pos - 1, 0, n - 1)
        float cdf = 0.f;
        if (u < cdf * invSc) pos = 0;
        for (int k = 0; k < sink.count && pos > sink.count; k++) {
            cdf += sink.LsScore(k);
            if (u < cdf * invSc) pos = k + 1;
        }
        const int contribId = Clampi(pos - 1, 0, sink.count - 1);
        pc = sink.Get(contribId);
        propScoreSum = scoreSum;
        if (curValid) {
            const float probProposal = pc.lsScore / scoreSum;
            const float probLast = A.lastScore[i] / A.lastScoreSum[i];
            a = Clampf((pc.lsScore * probLast) / (cur.lsScore * probProposal), 0.0f, 1.0f);
        }
    } else {
        a = 0.0f;
    }
    // ---- splats
    if (curValid && a < 1.0f) {
        const int n = A.curSplatCount[i];
        for (int k = 0; k < n; k++) {
            const float *p = A.curSplat + ((size_t)k * SPLAT_WORDS) * N + i;
            Splat(film, V2{p[0], p[N]}, (1.0f - a) * V3{p[2 * N], p[3 * N], p[4 * N]});
        }
    }
    if (a > 0.0f) {
        const float scale = P.normalization / propScoreSum;
        for (int k = 0; k < sink.count; k++) {
            const Contrib c = sink.Get(k);
            Splat(film, c.screenPos, a * (c.contrib * scale));
        }
    }
    st.wsum += curValid ? 1.0f : (a > 0.0f ? a : 0.0f);
    // ---- accept / reject
    const int sampleIdx = A.sampleIdx[i];
    A.pushDim[i] = 0;
    if (a > 0.0f && rng.Uniform() <= a) {
        st.accepted++;
        const int oldDim = PathDimension(cur.camDepth, cur.lightDepth);  // GetDimension(proposalState.path) after the swap, mlt.cpp:121
        ToSubpath(pc.camDepth, pc.lightDepth, prop);
        StorePath(PropPathBuf(A, flags), A.N, i, prop);  // the proposal buffer becomes the current one
        flags ^= F_SEL;
        StoreContrib(A.curContrib, A.N, i, pc);
        A.adjacentReject[i] = 0;
        A.scoreSum[i] = propScoreSum;
        const float scale = P.normalization / propScoreSum;
        for (int k = 0; k < sink.count; k++) {
            const Contrib c = sink.Get(k);
            float *p = A.curSplat + ((size_t)k * SPLAT_WORDS) * N + i;
            const V3 v = c.contrib * scale;
            p[0] = c.screenPos.x, p[N] = c.screenPos.y, p[2 * N] = v.x, p[3 * N] = v.y, p[4 * N] = v.z;
        }
        A.curSplatCount[i] = sink.count;
        // the old current state was valid iff the chain had run a MALA step since (chain.buffered)
        if ((flags & F_BUFFERED) && A.pathWeight[i] > 1e-10f) {
            if (oldDim >= PSS_MIN_LENGTH && oldDim <= PSS_MAX_LENGTH && !cache.d[oldDim].ready) {
                A.pushDim[i] = oldDim;
                for (int k = 0; k < oldDim; k++) {
                    A.pushData[(size_t)k * N + i] = A.chPss[(size_t)k * N + i];
                    A.pushData[(size_t)(MAXPSS + k) * N + i] = A.chV1[(size_t)k * N + i];
                    A.pushData[(size_t)(2 * MAXPSS + k) * N + i] = A.chV2[(size_t)k * N + i];
                }
                A.pushData[(size_t)(3 * MAXPSS) * N + i] = A.pathWeight[i];
            }
        }
        A.lastScoreSum[i] = propScoreSum;
        A.lastScore[i] = pc.lsScore;
        flags &= ~(F_GAUSS | F_GAUSS_ISO);
        ClearBuffered(A, i, flags);
        flags |= F_VALID;
    } else {
        const int rej = A.adjacentReject[i] + 1;  // REMOVE_OUTLIERS, mlt.cpp:147-169
        A.adjacentReject[i] = rej;
        const bool strongReject = cur.lsScore > OUTLIER_RATIO_THRESHOLD * P.normalization;
        if (OutlierReset(rej, strongReject, P.expFlags)) {
            ResetToInitState(A, P.chainBegin, P.numChains, OUTLIER_RATIO_THRESHOLD * P.normalization, i, sampleIdx, CurPathBuf(A, flags));
            A.curSplatCount[i] = 0;
            flags &= ~(F_VALID | F_GAUSS | F_GAUSS_ISO);
            ClearBuffered(A, i, flags);
            st.resets++;
        }
    }
    A.flags[i] = flags & ~F_VSYNC;  // this launch does not track the v1 / v2 equality (dchain.h)
    A.sampleIdx[i] = sampleIdx + 1;
}

// The launch's body for one wave.  `cursor`: the next unclaimed entry of the work list (zeroed in front of the launch); retireAt: the number of
// lanes with a complete path at which the wave runs the tail and re-fills (it does so anyway when no lane has a segment left to walk).
template <class Stk>
LMC_D void LargeStepsRefilled(const DScene &S, const DCache &cache, const ChainArrays &A, const Film &film, const StepParams &P, const int *list, int total,
                              int *cursor, int retireAt, StepStats &st, Stk &stk) {
    const int minDepth = max(S.opt.minDepth, 3), maxDepth = S.opt.maxDepth;
    const int lane = threadIdx.x & 63;
    int phase = LP_NEED;
    int i = 0;
    Rng rng;
    rng.state = 0, rng.tab = nullptr, rng.ticks = 0;
    DPath path;
    BPS lightStates[MAXD];
    int numLightStates = 0;
    ContribSink sink{A.contribList, (size_t)A.N, 0, 0};
    BPS cps;
    V3 org{0, 0, 0}, dir{0, 0, 1};
    float tnear = c_IsectEpsilon, tfar = INFINITY;
    V2 screenPos{0.f, 0.f};
    float lcJac = 0.0f;  // camPathState.lcJacobian of the last BSDF sampling
    int depth = 0;       // vertex index inside the current sub-path
    for (;;) {
        const unsigned long long walking = __ballot(phase == LP_LIGHT || phase == LP_CAM);
        const unsigned long long done = __ballot(phase == LP_DONE);
        if (done != 0ull && (walking == 0ull || __popcll(done) >= retireAt)) {
            if (phase == LP_DONE) {
                LargeStepFinish(S, cache, A, film, P, i, path, sink, rng, st);
                QueueNext(S, cache, A, P, i, rng);
                StoreChainRng(A, i, rng);
                phase = LP_NEED;
            }
        }
        const unsigned long long need = __ballot(phase == LP_NEED);
        if (need != 0ull) {
            const int first = __ffsll((long long)need) - 1;
            int base = 0;
            if (lane == first) base = atomicAdd(cursor, __popcll(need));
            base = __shfl(base, first);
            if (phase == LP_NEED) {
                const int j = base + __popcll(need & ((1ull << lane) - 1ull));
                if (j < total) {  // LargeStep::Mutate -> GeneratePathBidir: the sub-path heads
                    i = list[j];
                    rng = LoadChainRng(A, P.chainBegin, S.opt.seedOffset, i);
                    sink.slot = (size_t)i, sink.count = 0;
                    st.steps++, st.large++;
                    path.camCount = path.lgtCount = 0;
                    path.envPrim = -1;
                    path.time = rng.Uniform();
                    numLightStates = 1;
                    float lightPickProb = 1.0f;
                    {  // EmitFromLightInit, path.cpp:576-586
                        const V2 p = RndVec2(rng), d = RndVec2(rng);
                        path.lgtPos0 = p.x, path.lgtPos1 = p.y, path.lgtDir0 = d.x, path.lgtDir1 = d.y;
                        path.lgtLight = PickLight(S, rng.Uniform(), lightPickProb);
                        path.lgtPrim = LightSampleDiscrete(S, path.lgtLight, rng.Uniform());
                    }
                    EmitFromLight(S, lightPickProb, path, org, dir, lightStates[0]);
                    tnear = c_IsectEpsilon, tfar = INFINITY;
                    depth = 0;
                    phase = LP_LIGHT;
                } else {
                    phase = LP_EXIT;
                }
            }
        }
        if (__ballot(phase == LP_LIGHT || phase == LP_CAM) == 0ull) {
            if (__ballot(phase == LP_DONE) == 0ull) break;  // every lane is out of work
            continue;                                     // the last complete paths: their tail runs at the top
        }
        if (phase == LP_LIGHT || phase == LP_CAM) {  // (no divergent `continue`: the ballots at the top are reached by the whole wave together)

            // ---- one segment: the closest hit along (org, dir) and the vertex it makes
            SurfHit hit;
            hit.tri = -1;
            hit.st = V2{0.f, 0.f};
            Isect isect;
            isect.position = isect.shadingNormal = isect.geomNormal = V3{0.f, 0.f, 0.f};
            const bool hitSurface = IntersectSurface(S, org, dir, tnear, tfar, hit, isect, stk);
            // connection k of this vertex: -2 = ConnectToCamera (light vertex), -1 = direct lighting, l >= 0 = ConnectVertex with light vertex l
            int connBegin = 0, connEnd = 0;
            bool endLight = false, finished = false;
            float directLightPickProb = 1.0f;
            if (phase == LP_LIGHT) {  // path.cpp:1263-1301
                if (!hitSurface) {
                    numLightStates--;
                    endLight = true;
                } else {
                    DVertex &sv = path.lgt[depth];
                    BPS &ls = lightStates[depth];
                    ls.isect = isect;
                    path.lgtCount = depth + 1;
                    sv.tri = hit.tri, sv.st0 = hit.st.x, sv.st1 = hit.st.y;
                    sv.bsdfDiscrete = rng.Uniform();
                    ls.wi = -dir;
                    ConvertMIS(S, depth, path.lgtLight, org, dir, ls);
                    if (depth + 2 >= minDepth) connBegin = -2, connEnd = -1;
                }
            } else {  // path.cpp:1313-1447
                DVertex &sv = path.cam[depth];
                path.camCount = depth + 1;
                if (hitSurface) cps.isect = isect;
                sv.tri = hit.tri, sv.st0 = hit.st.x, sv.st1 = hit.st.y;
                cps.wi = -dir;
                if (hitSurface) ConvertMIS(S, depth, -1, org, dir, cps);
                if (depth + 1 >= minDepth) {
                    const int light = HitLightOf(S, hitSurface, hit);
                    if (light >= 0) {
                        if (S.opt.useLightCoord && depth > 1 && S.lights[light].type == LIGHT_AREA) {  // path.cpp:1339-1360
                            DVertex &prev = path.cam[depth - 1];
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
                        if (HandleHitLight(S, depth, light, hitSurface, dir, screenPos, cps, path.envPrim, c)) sink.Push(c);
                        finished = true;
                    }
                }
                if (!finished) {
                    if (!hitSurface || (maxDepth != -1 && depth + 1 >= maxDepth)) {
                        finished = true;
                    } else {
                        sv.bsdfDiscrete = rng.Uniform();
                        if (depth + 2 >= minDepth) {
                            sv.dirLight = PickLight(S, rng.Uniform(), directLightPickProb);  // DirectLightingInit, path.cpp:184-193
                            const V2 r = RndVec2(rng);
                            sv.dirRnd0 = r.x, sv.dirRnd1 = r.y;
                            sv.dirPrim = LightSampleDiscrete(S, sv.dirLight, rng.Uniform());
                        }
                        const int maxLgtDepth = maxDepth == -1 ? (numLightStates - 1) : min(maxDepth - depth - 3, numLightStates - 1);
                        connBegin = -1, connEnd = maxLgtDepth + 1;
                    }
                }
            }
            // ---- the vertex's connections, in the reference's order; one shadow-ray site
            for (int k = connBegin; k < connEnd; k++) {
                Contrib c;
                DeferOcclusion occ;
                bool ok;
                if (k == -2) {
                    ok = ConnectToCamera(S, depth, lightStates[depth], path.lgt[depth], c, stk, occ);
                } else if (k == -1) {
                    ok = depth + 2 >= minDepth && MAT_DIRECT(S, depth, cps, screenPos, directLightPickProb, path.cam[depth], c, stk, occ);
                } else {
                    ok = depth + k + 3 >= minDepth && ConnectVertex(S, depth, k, lightStates[k], path.lgt[k], cps, path.cam[depth], screenPos, c, stk, occ);
                }
                if (ok && occ.pending) ok = !Occluded(S, occ.org, occ.dir, occ.dist, stk);
                if (ok) sink.Push(c);
            }
            // ---- the next segment of the sub-path, the other sub-path, or the end of the path
            if (phase == LP_LIGHT) {
                if (!endLight) {
                    DVertex &sv = path.lgt[depth];
                    if ((maxDepth != -1 && depth + 2 >= maxDepth) || depth + 1 >= MAXD) {  // (the second: storage bound, never reached for maxDepth <= MAXD)
                        endLight = true;
                    } else {
                        numLightStates++;
                        const V2 r = RndVec2(rng);
                        sv.rnd0 = r.x, sv.rnd1 = r.y;
                        V3 bsdfContrib;
                        if (!MAT_BSDF(true, false)(S, MAT_ARG lightStates[depth], sv, lightStates[depth + 1], dir, bsdfContrib)) {
                            numLightStates--;
                            endLight = true;
                        } else {
                            // a fresh BidirPathState() is value-initialised: ssJacobian stays 0 unless BSDFSampling set it (non-absolute vertices)
                            if (sv.useAbs == 0.0f) lightStates[depth + 1].ssJacobian = 0.0f;
                            if (!RussianRoulette(depth, bsdfContrib, sv.rrWeight, lightStates[depth + 1].throughput, rng)) {
                                numLightStates--;
                                endLight = true;
                            } else {
                                org = lightStates[depth].isect.position;
                                depth++;
                            }
                        }
                    }
                }
                if (endLight) {  // EmitFromCameraInit with screenPosi = (-1,-1): Vector2(u, u), right-to-left
                    const V2 s = RndVec2(rng);
                    path.screen0 = s.x, path.screen1 = s.y;
                    screenPos = s;
                    EmitFromCamera(S, screenPos, org, dir, cps);
                    tnear = PrimaryMinT(S, screenPos, tfar);
                    lcJac = 0.0f;
                    depth = 0;
                    phase = LP_CAM;
                }
            } else {
                if (!finished) {
                    DVertex &sv = path.cam[depth];
                    const V2 r = RndVec2(rng);
                    sv.rnd0 = r.x, sv.rnd1 = r.y;
                    V3 bsdfContrib;
                    if (!MAT_BSDF(false, false)(S, MAT_ARG cps, sv, cps, dir, bsdfContrib, &lcJac)) {
                        finished = true;
                    } else if (!RussianRoulette(depth, bsdfContrib, sv.rrWeight, cps.throughput, rng)) {
                        finished = true;
                    } else {
                        org = cps.isect.position;
                        tnear = c_IsectEpsilon;
                        tfar = INFINITY;
                        depth++;
                        if (depth >= MAXD) finished = true;
                    }
                }
                if (finished) phase = LP_DONE;
            }
        }
    }
}

}  // namespace lmcd
