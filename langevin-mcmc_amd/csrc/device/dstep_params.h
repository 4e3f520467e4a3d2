// Per-launch parameters of the chain-step kernel (shared by host launch glue and device code).
#pragma once
namespace lmcd {

constexpr int LENGTH_DIST_MAX = 20;  // path lengths c + l - 1 < this (host-checked)
struct StepParams {
    float normalization;
    int numChains;   // all chains of the job (stride of the init-state arrays)
    int chainBegin;  // global id of this rank's chain 0
    int useGradient;  // 0: derivative library "absent" (isotropic until the cache is ready, path.cpp:4042-4053), 1: in-kernel gradient
    // bit 7 (128, LMC_EXP_OUTLIER_TEST: a TEST hook of every build) = outlier reset after 2 / 6 adjacent rejections (dchain.h OutlierReset).
    // Every other bit is a WORK-SKIPPING measurement switch and exists only in builds compiled with -DLMC_EXP_SWITCHES (scripts/build_exp.sh;
    // read through LMC_EXP below, which is the constant false in the shipped library -- host/context.cpp then refuses to create a context while
    // such a variable is set): 1 = no film splats in the lean kernel (LMC_EXP_NOSPLAT), 2 = no cache queries (LMC_EXP_NOQUERY), 256 / 512 = the
    // query cut short before its cell / after its occupancy word (LMC_EXP_QUERY_STOP=1|2), 4 = no gradient program (LMC_EXP_NOGRAD), 8 = no
    // statistics reduction (LMC_EXP_NOSTATS), 16 / 32 / 64 = H2MC without the Hessian program / the eigen-solve / the Hessian launch, 1024 = the lean kernel's rays
    // test the state's own triangle only: no tree walk, no shadow ray (LMC_EXP_NOTRAV), 2048 = ... does not store the proposal's vertices (LMC_EXP_NOSTORE), 4096 = a neighbour the cache query found (or a re-used one) is not used: isotropic (LMC_EXP_NOREUSE)
    int expFlags;
    // lengthDist of the multiplexed large step (mlt.h:99, mutation_large.h:45-47,90-101): PiecewiseConstant1D over the per-length
    // score sums of MLTInit (distribution.h:8-60); lengthCount = 0 unless `largestepmultiplexed` is set
    int lengthCount;
    float lengthFuncInt;
    float lengthFunc[LENGTH_DIST_MAX], lengthCdf[LENGTH_DIST_MAX + 1];
    int maxDervDepth;  // --max-derivatives-depth (main.cpp:46,59-60): derivative programs exist for path lengths c + l - 1 <= this (path.cpp:4030-4037)
};

#ifdef LMC_EXP_SWITCHES
#define LMC_EXP(flags, bit) (((flags) & (bit)) != 0)
#else
#define LMC_EXP(flags, bit) false
#endif

// Work lists of the next step: chains are appended by the launch that finishes their current step.
struct NextLists {
    int *large, *smallGrad, *smallPlain;  // chain indices
    int *counts;                          // [0] large, [1] smallGrad, [2] smallPlain
};

enum StepKernelKind { STEP_LARGE = 0, STEP_SMALL_GRAD = 1, STEP_SMALL_PLAIN = 2 };

}  // namespace lmcd
