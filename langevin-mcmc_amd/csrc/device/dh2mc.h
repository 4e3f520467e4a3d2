// H2MC proposal parameters: /root/reference/src/h2mc.h:9-29 (H2MCParam).  The Gaussian itself (h2mc.cpp:3-142: symmetric
// eigen-decomposition of the Hessian, per-eigenvalue variance / offset remap, isotropic prior, dense mean / covL / invCov / logDet) is
// built on the device by h2gauss.hip (16 lanes per state).  The CPU oracle has its own serial restatement, oracle/h2mc_serial.h; the two
// share nothing but these constants and follow the same rotation sequence by design, not by source.  Host- and device-compilable.
#pragma once
#include "dmath.h"

namespace lmcd {

constexpr int H2_MAXDIM = 16;  // derivative programs exist up to path length 8 (--max-derivatives-depth 8)

struct H2MCParam {
    float sigma, posScaleFactor, posOffsetFactor, negScaleFactor, negOffsetFactor, L;
};
LMC_HD H2MCParam MakeH2MCParam(float sigma) {  // h2mc.h:10-16, L = pi/2
    H2MCParam p;
    p.sigma = sigma;
    p.L = float(3.14159265358979323846 / 2.0);
    // exp of this file is the deterministic float routine of dtrans.h (the CPU oracle and the device agree to the bit; the
    // reference's float libm calls are within 1-2 ulp of these); sin / cos of the one constant L in double, rounded once
    const float eL = lexpf(p.L), emL = lexpf(-p.L);
    p.posScaleFactor = 0.5f * (eL - emL) * 0.5f * (eL - emL);
    p.posOffsetFactor = 0.5f * (eL + emL - 1.0f);
    const float sL = (float)sin((double)p.L), cL = (float)cos((double)p.L);
    p.negScaleFactor = sL * sL;
    p.negOffsetFactor = -(cL - 1.0f);
    return p;
}

}  // namespace lmcd
