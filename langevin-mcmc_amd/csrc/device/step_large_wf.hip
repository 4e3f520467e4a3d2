// Large steps as a WAVEFRONT of launches: LargeStep::Mutate (mutation_large.h:31-128) over GeneratePathBidir (path.cpp:1237-1449), a chain's arithmetic,
// draws and their order exactly those of k_step<large> (step_kernel.h, dpath.h GeneratePathBidir, dstep.h StepChain), cut at the segments of the camera
// sub-path.  k_large_segment<.., FIRST> runs, per large-step chain: time, the light sub-path (its vertices streamed to the chain's PROPOSAL path buffer, its
// states to a scratch the camera vertices connect to), the camera ray and the FIRST camera segment; k_large_segment<.., false> at camera depth d = 1, 2, ...
// runs one segment of the chains whose path is still alive at depth d -- a list the launch of depth d - 1 appended to, so its waves are FULL whatever the
// paths' lengths: trace, emitter hit, direct lighting and the connections to the light vertices (one shadow-ray site), BSDF sampling, Russian roulette; the
// vertex goes to the proposal buffer, survivors (with a 20-word walk state) to the next list, and a lane whose path ENDS runs the step's tail on the spot
// (technique selection, splats, accept / reject -- an accepted path IS in the proposal buffer: the head words of the chosen technique are written and F_SEL
// flips --, the next step's kind).  The step's large-step list is cut into `parts` ranges, each a chain of launches on a stream of its own: a launch is placed
// at the rate at which the hot launch's waves retire, and between two launches of one chain the slots go back to the hot launch -- with several chains in
// flight a finished wave's slot goes to another part's pending launch instead.
// Why: one lane of k_step<large> walks a whole path and every connection of it; the lengths of freshly generated paths differ widely (a camera ray that leaves the
// scene ends the path at its first segment, Russian roulette from the fourth on), the wave waits for its longest path with 17 % (torus) / 10-25 % (glossy
// scenes) of its lanes active (profiles/r05_bi_*, r05_bk_*), and what a launch costs the step is the wave slots it holds meanwhile.  Re-filling the idle lanes
// inside a wave does not help (a launch of 3 277 wave-tasks on 2 048 slots has nothing to re-fill from, profiles/r06_av_*): compaction ACROSS waves at every
// segment does.  The random-number stream of a chain lives in its slot between launches as it does between steps (dchain.h LoadChainRng / StoreChainRng).
#ifndef LMC_NO_RNG_JUMP_LDS
#define LMC_RNG_JUMP_LDS  // drng.h: the PCG jump constants of these launches live in LDS
#endif
#include "step_kernel.h"
#include "dsmall.h"

using namespace lmcd;

namespace lmcd {

// walk state, SoA over the position j in the step's large-step list (stride = number of slots): words 0-8 the camera state's isect (position / shading normal /
// geometric normal), 9-11 the ray direction, 12 accMISWPrev, 13 accMISWThis, 14-16
// throughput, 17 ssJacobian, 18 lcJacobian, 19 contributions so far | light states << 16
constexpr int WF_STATE_WORDS = 20;
constexpr int WF_LGT_WORDS = 18;  // one BPS of the light sub-path: isect 9, wi 3, accMISWPrev / This, throughput 3, ssJacobian

LMC_D void WfStoreLightState(float *lgt, size_t N, int j, int d, const BPS &s) {
    float *p = lgt + ((size_t)d * WF_LGT_WORDS) * N + j;
    const float w[WF_LGT_WORDS] = {s.isect.position.x, s.isect.position.y, s.isect.position.z, s.isect.shadingNormal.x, s.isect.shadingNormal.y, s.isect.shadingNormal.z,
                                   s.isect.geomNormal.x, s.isect.geomNormal.y, s.isect.geomNormal.z, s.wi.x, s.wi.y, s.wi.z, s.accMISWPrev, s.accMISWThis,
                                   s.throughput.x, s.throughput.y, s.throughput.z, s.ssJacobian};
#pragma unroll
    for (int k = 0; k < WF_LGT_WORDS; k++) p[(size_t)k * N] = w[k];
}
LMC_D BPS WfLoadLightState(const float *lgt, size_t N, int j, int d) {
    const float *p = lgt + ((size_t)d * WF_LGT_WORDS) * N + j;
    float w[WF_LGT_WORDS];
#pragma unroll
    for (int k = 0; k < WF_LGT_WORDS; k++) w[k] = p[(size_t)k * N];
    BPS s;
    s.isect.position = V3{w[0], w[1], w[2]}, s.isect.shadingNormal = V3{w[3], w[4], w[5]}, s.isect.geomNormal = V3{w[6], w[7], w[8]};
    s.wi = V3{w[9], w[10], w[11]};
    s.accMISWPrev = w[12], s.accMISWThis = w[13];
    s.throughput = V3{w[14], w[15], w[16]};
    s.ssJacobian = w[17];
    return s;
}
LMC_D void WfStoreState(float *st, size_t N, int j, const BPS &cps, V3 dir, float lcJac, int sinkCount, int numLightStates) {
    float *p = st + j;
    const float w[WF_STATE_WORDS] = {cps.isect.position.x, cps.isect.position.y, cps.isect.position.z, cps.isect.shadingNormal.x, cps.isect.shadingNormal.y,
                                     cps.isect.shadingNormal.z, cps.isect.geomNormal.x, cps.isect.geomNormal.y, cps.isect.geomNormal.z, dir.x, dir.y, dir.z,
                                     cps.accMISWPrev, cps.accMISWThis, cps.throughput.x, cps.throughput.y, cps.throughput.z, cps.ssJacobian, lcJac,
                                     __int_as_float(sinkCount | (numLightStates << 16))};
#pragma unroll
    for (int k = 0; k < WF_STATE_WORDS; k++) p[(size_t)k * N] = w[k];
}

struct WfBuffers {
    float *state;     // WF_STATE_WORDS x N
    float *lgt;       // maxLightStates x WF_LGT_WORDS x N
    int *alive[2];    // per part, aliveStride entries each: the list positions alive at an even / odd camera depth (depth 0: the part's whole range)
    int aliveStride;
    int *count;       // [part * (MAXD + 2) + d] the number of entries alive at camera depth d (d >= 1)
};

#ifndef LMC_WF_WAVES
#define LMC_WF_WAVES 3
#endif
// FIRST: camera depth 0 behind the head of GeneratePathBidir (path.cpp:1237-1311) for every position of the part's range; else one camera segment
// (path.cpp:1313-1447, one iteration of the camera loop) of the positions alive at `depth`
template <bool GLOSSY, bool QUANT, bool FIRST>
__global__ void __launch_bounds__(64, LMC_WF_WAVES) k_large_segment(DScene S, const DCache *cachePtr, ChainArrays A, Film film, StepParams P, const int *list, const int *listCount, WfBuffers W,
                                                                    int stage, int depthBegin, int depthEnd, int part, int parts) {
    const int total = *listCount;
    const int jBegin = (int)((long long)total * part / parts), jEnd = (int)((long long)total * (part + 1) / parts);
    int *counts = W.count + part * (MAXD + 2);
    const int count = FIRST ? jEnd - jBegin : counts[stage];
    if ((int)(blockIdx.x * 64) >= count) return;
    LMC_RNG_JUMP_INIT();
    LMC_MAT_LDS_INIT(S);
    extern __shared__ int ldsStack[];
    typedef LdsStackT<GLOSSY, QUANT> Stk;
    Stk stk{ldsStack + threadIdx.x, 64, 0};
    const DCache &cache = *cachePtr;
    const size_t N = A.N;
    const int minDepth = max(S.opt.minDepth, 3), maxDepth = S.opt.maxDepth;
    const int *aliveIn = W.alive[stage & 1] + (size_t)part * W.aliveStride;
    int *aliveOut = W.alive[(stage + 1) & 1] + (size_t)part * W.aliveStride;
    StepStats st;
    const int rounds = (count + (int)gridDim.x * 64 - 1) / ((int)gridDim.x * 64);
    for (int r = 0; r < rounds; r++) {
        const int k = (r * gridDim.x + blockIdx.x) * 64 + threadIdx.x;
        bool survives = false;
        int j = 0;
        if (k < count) {
            j = FIRST ? jBegin + k : aliveIn[k];
            const int i = list[j];
            int flags = A.flags[i];
            float *prop = PropPathBuf(A, flags);
            Rng rng = LoadChainRng(A, P.chainBegin, S.opt.seedOffset, i);
            ContribSink sink{A.contribList, N, (size_t)i, 0};
            BPS cps;
            V3 org, dir;
            V2 screenPos;
            float lcJac = 0.0f;  // camPathState.lcJacobian of the last BSDF sampling
            int numLightStates;
            float tnear = c_IsectEpsilon, tfar = INFINITY;
            if constexpr (FIRST) {
                TraceOcclusion trace;
                DPath hd;  // the emitter fields only
                const float time = rng.Uniform();
                numLightStates = 1;
                float lightPickProb = 1.0f;
                {  // EmitFromLightInit, path.cpp:576-586
                    const V2 p = RndVec2(rng), d = RndVec2(rng);
                    hd.lgtPos0 = p.x, hd.lgtPos1 = p.y, hd.lgtDir0 = d.x, hd.lgtDir1 = d.y;
                    hd.lgtLight = PickLight(S, rng.Uniform(), lightPickProb);
                    hd.lgtPrim = LightSampleDiscrete(S, hd.lgtLight, rng.Uniform());
                }
                StS(&prop[(size_t)PW_TIME * N + i], time);
                StS(&prop[(size_t)PW_LGTPOS0 * N + i], hd.lgtPos0), StS(&prop[(size_t)PW_LGTPOS1 * N + i], hd.lgtPos1);
                StS(&prop[(size_t)PW_LGTDIR0 * N + i], hd.lgtDir0), StS(&prop[(size_t)PW_LGTDIR1 * N + i], hd.lgtDir1);
                StS(&prop[(size_t)PW_LGTLIGHT * N + i], __int_as_float(hd.lgtLight)), StS(&prop[(size_t)PW_LGTPRIM * N + i], __int_as_float(hd.lgtPrim));
                StS(&prop[(size_t)PW_ENVPRIM * N + i], __int_as_float(-1));
                StS(&prop[(size_t)PW_LENS0 * N + i], 0.0f), StS(&prop[(size_t)PW_LENS1 * N + i], 0.0f);
                BPS ls;
                ls.isect.position = ls.isect.shadingNormal = ls.isect.geomNormal = V3{0.f, 0.f, 0.f};
                ls.wi = V3{0.f, 0.f, 0.f};
                EmitFromLight(S, lightPickProb, hd, org, dir, ls);
                for (int lgtDepth = 0;; lgtDepth++) {
                    DVertex sv;
                    sv.tri = -1, sv.st0 = sv.st1 = sv.rnd0 = sv.rnd1 = sv.bsdfDiscrete = sv.useAbs = sv.rrWeight = sv.dirRnd0 = sv.dirRnd1 = 0.f, sv.dirLight = sv.dirPrim = 0;
                    SurfHit hit;
                    const bool hitSurface = IntersectSurface(S, org, dir, c_IsectEpsilon, INFINITY, hit, ls.isect, stk);
                    if (!hitSurface) {
                        numLightStates--;
                        break;
                    }
                    sv.tri = hit.tri, sv.st0 = hit.st.x, sv.st1 = hit.st.y;
                    sv.bsdfDiscrete = rng.Uniform();
                    ls.wi = -dir;
                    ConvertMIS(S, lgtDepth, hd.lgtLight, org, dir, ls);
                    if (lgtDepth + 2 >= minDepth) {
                        Contrib c;
                        if (ConnectToCamera(S, lgtDepth, ls, sv, c, stk, trace)) sink.Push(c);
                    }
                    WfStoreLightState(W.lgt, N, j, lgtDepth, ls);  // lightStates[lgtDepth] is final: the camera vertices connect to it
                    bool goOn = !(maxDepth != -1 && lgtDepth + 2 >= maxDepth) && lgtDepth + 1 < MAXD;
                    if (goOn) {
                        numLightStates++;
                        const V2 rr = RndVec2(rng);
                        sv.rnd0 = rr.x, sv.rnd1 = rr.y;
                        V3 bsdfContrib;
                        BPS next;
                        next.isect = ls.isect, next.wi = ls.wi;
                        next.ssJacobian = 0.0f;  // a fresh BidirPathState() is value-initialised: ssJacobian stays 0 unless BSDFSampling set it
                        if (!MAT_BSDF(true, false)(S, MAT_ARG ls, sv, next, dir, bsdfContrib)) {
                            numLightStates--;
                            goOn = false;
                        } else {
                            if (sv.useAbs == 0.0f) next.ssJacobian = 0.0f;
                            if (!RussianRoulette(lgtDepth, bsdfContrib, sv.rrWeight, next.throughput, rng)) {
                                numLightStates--;
                                goOn = false;
                            } else {
                                org = ls.isect.position;
                                ls = next;
                            }
                        }
                    }
                    StoreVertex(prop, N, i, true, lgtDepth, sv);
                    if (!goOn) break;
                }
                // EmitFromCameraInit with screenPosi = (-1,-1): Vector2(u, u), right-to-left
                screenPos = RndVec2(rng);
                StS(&prop[(size_t)PW_SCREEN0 * N + i], screenPos.x), StS(&prop[(size_t)PW_SCREEN1 * N + i], screenPos.y);
                cps.isect.position = cps.isect.shadingNormal = cps.isect.geomNormal = V3{0.f, 0.f, 0.f};
                EmitFromCamera(S, screenPos, org, dir, cps);
                tnear = PrimaryMinT(S, screenPos, tfar);
            } else {
                const float *sp = W.state + j;
                float w[WF_STATE_WORDS];
#pragma unroll
                for (int q = 0; q < WF_STATE_WORDS; q++) w[q] = sp[(size_t)q * N];
                screenPos = V2{LdS(&prop[(size_t)PW_SCREEN0 * N + i]), LdS(&prop[(size_t)PW_SCREEN1 * N + i])};
                cps.isect.position = V3{w[0], w[1], w[2]}, cps.isect.shadingNormal = V3{w[3], w[4], w[5]}, cps.isect.geomNormal = V3{w[6], w[7], w[8]};
                dir = V3{w[9], w[10], w[11]};
                cps.accMISWPrev = w[12], cps.accMISWThis = w[13];
                cps.throughput = V3{w[14], w[15], w[16]};
                cps.ssJacobian = w[17];
                lcJac = w[18];
                const int packed = __float_as_int(w[19]);
                numLightStates = packed >> 16;
                sink.count = packed & 0xffff;
                org = cps.isect.position;
            }

            // ---- the camera segments of this launch: depthBegin .. depthEnd - 1, while the path lives
            bool finished = false;
            for (int depth = depthBegin; depth < depthEnd && !finished; depth++) {
                if (depth != depthBegin) org = cps.isect.position, tnear = c_IsectEpsilon, tfar = INFINITY;
                DVertex sv;
                sv.tri = -1, sv.st0 = sv.st1 = sv.rnd0 = sv.rnd1 = sv.bsdfDiscrete = sv.useAbs = sv.rrWeight = sv.dirRnd0 = sv.dirRnd1 = 0.f, sv.dirLight = sv.dirPrim = 0;
                SurfHit hit;
                hit.tri = -1;
                hit.st = V2{0.f, 0.f};
                Isect isect;
                isect.position = isect.shadingNormal = isect.geomNormal = V3{0.f, 0.f, 0.f};
                const bool hitSurface = IntersectSurface(S, org, dir, tnear, tfar, hit, isect, stk);
                if (hitSurface) cps.isect = isect;
                sv.tri = hit.tri, sv.st0 = hit.st.x, sv.st1 = hit.st.y;
                cps.wi = -dir;
                if (hitSurface) ConvertMIS(S, depth, -1, org, dir, cps);
                if (depth + 1 >= minDepth) {
                    const int light = HitLightOf(S, hitSurface, hit);
                    if (light >= 0) {
                        if (S.opt.useLightCoord && depth > 1 && S.lights[light].type == LIGHT_AREA) {  // path.cpp:1339-1360
                            // area light: the BSDF sampling coordinates of the previous vertex become the light's direct sampling coordinates
                            const V2 tp = TriangleSampleParam(S, hit.tri, cps.isect.position);
                            StS(&prop[(size_t)VertWord(false, depth - 1, 3) * N + i], tp.x), StS(&prop[(size_t)VertWord(false, depth - 1, 4) * N + i], tp.y);
                            V3 dirToPrev = cps.isect.position - org;
                            const float distSq = LengthSquared(dirToPrev);
                            const float invDistSq = inverse(distSq);
                            const float invDist = sqrtf(invDistSq);
                            dirToPrev = dirToPrev * invDist;
                            cps.ssJacobian *= fabsf(Dot(dirToPrev, cps.isect.shadingNormal) * invDistSq) * (lcJac * S.meshes[S.tris[hit.tri].mesh].invTotalArea);
                        }
                        Contrib c;
                        int envPrim = -1;
                        if (HandleHitLight(S, depth, light, hitSurface, dir, screenPos, cps, envPrim, c)) sink.Push(c);
                        if (envPrim != -1) StS(&prop[(size_t)PW_ENVPRIM * N + i], __int_as_float(envPrim));
                        finished = true;
                    }
                }
                if (!finished && (!hitSurface || (maxDepth != -1 && depth + 1 >= maxDepth))) finished = true;
                if (!finished) {
                    sv.bsdfDiscrete = rng.Uniform();
                    float directLightPickProb = 1.0f;
                    const bool direct = depth + 2 >= minDepth;
                    if (direct) {
                        sv.dirLight = PickLight(S, rng.Uniform(), directLightPickProb);  // DirectLightingInit, path.cpp:184-193
                        const V2 rr = RndVec2(rng);
                        sv.dirRnd0 = rr.x, sv.dirRnd1 = rr.y;
                        sv.dirPrim = LightSampleDiscrete(S, sv.dirLight, rng.Uniform());
                    }
                    // the vertex's connections in the reference's order -- direct lighting (kc = -1), then the light vertices kc = 0 .. maxLgtDepth -- through ONE shadow-ray
                    // site: the strategies draw no random numbers, so they are evaluated first and the ray is cast behind them (dpath.h DeferOcclusion)
                    const int maxLgtDepth = maxDepth == -1 ? (numLightStates - 1) : min(maxDepth - depth - 3, numLightStates - 1);
                    for (int kc = direct ? -1 : 0; kc <= maxLgtDepth; kc++) {
                        Contrib c;
                        DeferOcclusion occ;
                        bool ok;
                        if (kc < 0) {
                            ok = MAT_DIRECT(S, depth, cps, screenPos, directLightPickProb, sv, c, stk, occ);
                        } else if (depth + kc + 3 >= minDepth) {
                            const BPS lps = WfLoadLightState(W.lgt, N, j, kc);
                            DVertex lv;
                            lv.tri = __float_as_int(LdS(&prop[(size_t)VertWord(true, kc, 0) * N + i]));
                            lv.st0 = LdS(&prop[(size_t)VertWord(true, kc, 1) * N + i]), lv.st1 = LdS(&prop[(size_t)VertWord(true, kc, 2) * N + i]);
                            ok = ConnectVertex(S, depth, kc, lps, lv, cps, sv, screenPos, c, stk, occ);
                        } else {
                            ok = false;
                        }
                        if (ok && occ.pending) ok = !Occluded(S, occ.org, occ.dir, occ.dist, stk);
                        if (ok) sink.Push(c);
                    }
                    const V2 rr = RndVec2(rng);
                    sv.rnd0 = rr.x, sv.rnd1 = rr.y;
                    V3 bsdfContrib;
                    if (!MAT_BSDF(false, false)(S, MAT_ARG cps, sv, cps, dir, bsdfContrib, &lcJac)) finished = true;
                    else if (!RussianRoulette(depth, bsdfContrib, sv.rrWeight, cps.throughput, rng))
                        finished = true;
                    else if (depth + 1 >= MAXD)
                        finished = true;
                }
                StoreVertex(prop, N, i, false, depth, sv);
                if (finished) {
                    // ---- the path is complete: what StepChain<true, false, false, 0> does behind GeneratePathBidir (dstep.h: selection mutation_large.h:60-112, splats
                    // mlt.cpp:103-112, accept / reject mlt.cpp:113-170), on the contribution list and the path in the proposal buffer
                    const bool curValid = flags & F_VALID;
                    const Contrib cur = LoadContrib(A.curContrib, A.N, i);
                    st.steps++, st.large++;
                    Contrib pc;
                    pc.camDepth = pc.lightDepth = 0;
                    pc.lsScore = pc.ssScore = 0.f;
                    float a = 1.0f;
                    float propScoreSum = 0.f;
                    if (sink.count > 0) {
                        float scoreSum = 0.f;
                        for (int q = 0; q < sink.count; q++) scoreSum += sink.LsScore(q);  // contribCdf.back()
                        const float invSc = inverse(scoreSum);
                        const float u = rng.Uniform();
                        int pos = sink.count + 1;  // std::upper_bound(cdf * invSc, u): first element > u; contribId = clamp(pos - 1, 0, n - 1)
                        float cdf = 0.f;
                        if (u < cdf * invSc) pos = 0;
                        for (int q = 0; q < sink.count && pos > sink.count; q++) {
                            cdf += sink.LsScore(q);
                            if (u < cdf * invSc) pos = q + 1;
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
                    if (curValid && a < 1.0f) {  // splats
                        const int n = A.curSplatCount[i];
                        for (int q = 0; q < n; q++) {
                            const float *p = A.curSplat + ((size_t)q * SPLAT_WORDS) * N + i;
                            Splat(film, V2{p[0], p[N]}, (1.0f - a) * V3{p[2 * N], p[3 * N], p[4 * N]});
                        }
                    }
                    if (a > 0.0f) {
                        const float scale = P.normalization / propScoreSum;
                        for (int q = 0; q < sink.count; q++) {
                            const Contrib c = sink.Get(q);
                            Splat(film, c.screenPos, a * (c.contrib * scale));
                        }
                    }
                    st.wsum += curValid ? 1.0f : (a > 0.0f ? a : 0.0f);
                    const int sampleIdx = A.sampleIdx[i];
                    A.pushDim[i] = 0;
                    if (a > 0.0f && rng.Uniform() <= a) {  // accept
                        st.accepted++;
                        const int oldDim = PathDimension(cur.camDepth, cur.lightDepth);  // GetDimension(proposalState.path) after the swap, mlt.cpp:121
                        // ToSubpath (path.cpp:1660-1669) on the path in the proposal buffer, which becomes the current one
                        StS(&prop[(size_t)PW_CAMDEPTH * N + i], __int_as_float(pc.camDepth)), StS(&prop[(size_t)PW_LGTDEPTH * N + i], __int_as_float(pc.lightDepth));
                        StS(&prop[(size_t)PW_CAMCOUNT * N + i], __int_as_float(max(pc.camDepth - 1, 0))), StS(&prop[(size_t)PW_LGTCOUNT * N + i], __int_as_float(max(pc.lightDepth - 1, 0)));
                        if (pc.lightDepth != 0) StS(&prop[(size_t)PW_ENVPRIM * N + i], __int_as_float(-1));
                        flags ^= F_SEL;
                        StoreContrib(A.curContrib, A.N, i, pc);
                        A.adjacentReject[i] = 0;
                        A.scoreSum[i] = propScoreSum;
                        const float scale = P.normalization / propScoreSum;
                        for (int q = 0; q < sink.count; q++) {
                            const Contrib c = sink.Get(q);
                            float *p = A.curSplat + ((size_t)q * SPLAT_WORDS) * N + i;
                            const V3 v = c.contrib * scale;
                            p[0] = c.screenPos.x, p[N] = c.screenPos.y, p[2 * N] = v.x, p[3 * N] = v.y, p[4 * N] = v.z;
                        }
                        A.curSplatCount[i] = sink.count;
                        // the old current state was valid iff the chain had run a MALA step since (chain.buffered)
                        if ((flags & F_BUFFERED) && A.pathWeight[i] > 1e-10f) {
                            if (oldDim >= PSS_MIN_LENGTH && oldDim <= PSS_MAX_LENGTH && !cache.d[oldDim].ready) {
                                A.pushDim[i] = oldDim;
                                for (int q = 0; q < oldDim; q++) {
                                    A.pushData[(size_t)q * N + i] = A.chPss[(size_t)q * N + i];
                                    A.pushData[(size_t)(MAXPSS + q) * N + i] = A.chV1[(size_t)q * N + i];
                                    A.pushData[(size_t)(2 * MAXPSS + q) * N + i] = A.chV2[(size_t)q * N + i];
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
                    QueueNext(S, cache, A, P, i, rng);
                }
            }
            if (!finished) {  // the path goes on: its state to the next launch
                WfStoreState(W.state, N, j, cps, dir, lcJac, sink.count, numLightStates);
                survives = true;
            }
            StoreChainRng(A, i, rng);
        }
        // append the survivors of this wave to the next launch's list (one atomic per wave)
        const unsigned long long m = __ballot(survives);
        if (m != 0ull) {
            const int lane = threadIdx.x & 63;
            const int first = __ffsll((long long)m) - 1;
            int base = 0;
            if (lane == first) base = atomicAdd(&counts[stage + 1], __popcll(m));
            base = __shfl(base, first);
            if (survives) aliveOut[base + __popcll(m & ((1ull << lane) - 1ull))] = j;
        }
    }
    __shared__ int sStats[9];
    BlockReduceStats(st, A.counters, A.weightSum, sStats);
}

}  // namespace lmcd

size_t LargeWavefrontStateWords() { return WF_STATE_WORDS; }
size_t LargeWavefrontLightWords() { return WF_LGT_WORDS; }
int LargeWavefrontCountWords(int parts) { return parts * (MAXD + 2); }

// one part's chain of launches on stream s (the caller forks / joins the parts' streams and zeroes `counts` in front of them).  cuts: the camera depths at which the
// chains still alive are compacted into a new list (ascending, at most MAXD; none: one launch walks whole paths)
bool LaunchStepLargeWavefrontPart(const DScene &S, const DCache *cache, const ChainArrays &A, const Film &film, const StepParams &P, const int *list, const int *listCount,
                                  float *state, float *lgt, int *alive0, int *alive1, int aliveStride, int *counts, bool glossy, int gridBlocks, int bvhStackNeed, int part, int parts,
                                  const int *cuts, int numCuts, hipStream_t s) {
    if (bvhStackNeed > BVH_LDS_STACK) return false;  // a tree too deep for the LDS stack: the single launch (LaunchStepLarge)
    RequireJumpLdsBlock(64);
    const size_t ldsBytes = (size_t)64 * ((bvhStackNeed + 7) / 8 * 8) * sizeof(int);
    const bool quant = S.qnodes != nullptr;
    const WfBuffers W{state, lgt, {alive0, alive1}, aliveStride, counts};
    const int segments = S.opt.maxDepth == -1 ? MAXD : min(S.opt.maxDepth, MAXD);
    const int grid = (gridBlocks + parts - 1) / parts;
    int bounds[MAXD + 2], nb = 0;
    bounds[nb++] = 0;
    for (int k = 0; k < numCuts && nb <= MAXD; k++)
        if (cuts[k] > bounds[nb - 1] && cuts[k] < segments) bounds[nb++] = cuts[k];
    bounds[nb] = segments;
#define LMC_WF_LAUNCH(G, Q)                                                                                                                                                          \
    do {                                                                                                                                                                             \
        hipLaunchKernelGGL((k_large_segment<G, Q, true>), dim3(grid), dim3(64), ldsBytes, s, S, cache, A, film, P, list, listCount, W, 0, bounds[0], bounds[1], part, parts);        \
        for (int g = 1; g < nb; g++)                                                                                                                                                 \
            hipLaunchKernelGGL((k_large_segment<G, Q, false>), dim3(grid), dim3(64), ldsBytes, s, S, cache, A, film, P, list, listCount, W, g, bounds[g], bounds[g + 1], part, parts); \
    } while (0)
    if (glossy && quant) LMC_WF_LAUNCH(true, true);
    else if (glossy) LMC_WF_LAUNCH(true, false);
    else if (quant) LMC_WF_LAUNCH(false, true);
    else
        LMC_WF_LAUNCH(false, false);
#undef LMC_WF_LAUNCH
    return true;
}
