// Per-launch parameters of the chain-step kernel (shared by host launch glue and device code).
#pragma once
namespace lmcd {

struct StepParams {
    float normalization;
    int numChains;
    int useGradient;  // 0: derivative library "absent" (isotropic until the cache is ready, path.cpp:4042-4053), 1: in-kernel gradient
};

}  // namespace lmcd
