// Per-launch parameters of the chain-step kernel (shared by host launch glue and device code).
#pragma once
namespace lmcd {

constexpr int LENGTH_DIST_MAX = 20;  // path lengths c + l - 1 < this (host-checked)
struct StepParams {
    float normalization;
    int numChains;   // all chains of the job (stride of the init-state arrays)
    int chainBegin;  // global id of this rank's chain 0
    int useGradient;  // 0: derivative library "absent" (isotropic until the cache is ready, path.cpp:4042-4053), 1: in-kernel gradient
    int expFlags;      // measurement aids, never set in production: bit 0 = skip the film splats of the lean kernel (LMC_EXP_NOSPLAT), bit 1 = skip its cache queries, bit 2 = skip the gradient program of the generic kernel (LMC_EXP_NOGRAD), bit 3 = no statistics reduction in the lean kernel, bits 4 / 5 = H2MC without the Hessian program / without the eigen-solve, bit 7 = outlier reset after 2 / 6 adjacent rejections (tests: dchain.h OutlierReset)
    // lengthDist of the multiplexed large step (mlt.h:99, mutation_large.h:45-47,90-101): PiecewiseConstant1D over the per-length
    // score sums of MLTInit (distribution.h:8-60); lengthCount = 0 unless `largestepmultiplexed` is set
    int lengthCount;
    float lengthFuncInt;
    float lengthFunc[LENGTH_DIST_MAX], lengthCdf[LENGTH_DIST_MAX + 1];
    int maxDervDepth;  // --max-derivatives-depth (main.cpp:46,59-60): derivative programs exist for path lengths c + l - 1 <= this (path.cpp:4030-4037)
};

// Work lists of the next step: chains are appended by the launch that finishes their current step.
struct NextLists {
    int *large, *smallGrad, *smallPlain;  // chain indices
    int *counts;                          // [0] large, [1] smallGrad, [2] smallPlain
};

enum StepKernelKind { STEP_LARGE = 0, STEP_SMALL_GRAD = 1, STEP_SMALL_PLAIN = 2 };

}  // namespace lmcd
