// TEST INFRASTRUCTURE (CPU oracle): the serial H2MC Gaussian -- /root/reference/src/h2mc.cpp:3-142 (ComputeGaussian) and gaussian.cpp:24-55
// (dense GaussianLogPdf / GenerateSample).  The reference calls Eigen::SelfAdjointEigenSolver (third party, not vendored: parity unpinned,
// SURVEY.md 8c); here a cyclic Jacobi solver (eigenvector sign and order are a convention: mean, invCov, covL covL^T and logDet do not depend on
// it; tests/test_h2mc.py checks those against numpy.linalg.eigh).  The device builds the same Gaussian with its own 16-lane solver
// (langevin-mcmc_amd/csrc/device/h2gauss.hip), which follows the same rotation sequence and conventions so that both sides draw the same
// samples up to rounding; this header is NOT compiled into the product (it was shared with the device until round 4).
#pragma once
#include "../langevin-mcmc_amd/csrc/device/dh2mc.h"  // H2MCParam / MakeH2MCParam: the constants of h2mc.h:10-16

namespace lmcd {

// Symmetric eigen-decomposition by cyclic Jacobi rotations.  A (n x n, row-major, stride n) is destroyed; on return w holds
// the eigenvalues in ascending order (Eigen's convention) and column j of V (row-major, stride n) the unit eigenvector of w[j].
// MA / MV: anything indexable as a flat n x n array (float *, MatRef): the device keeps A in LDS (dh2step.h), same arithmetic
// Rotation order (round 5): the ROUND-ROBIN cyclic order -- n - 1 rounds of n / 2 disjoint pairs per sweep ("chess tournament": player m - 1
// stays, the others rotate; m = n rounded up to even, pairs with the bye skipped) -- instead of the row-cyclic (0,1), (0,2), ... order.  The
// rotations of a round touch disjoint rows / columns, so a round is ONE similarity transform A <- J^T A J with J = the product of its rotations,
// evaluated as: all angles from the matrix at the start of the round, then A <- A J (every pair's two columns), then A <- J^T A (every pair's two
// rows), V <- V J.  The device (h2gauss.hip) runs exactly this with the lanes = rows / columns: 3 barriers per round instead of 4 per rotation.
// Eigenvalues are those of the matrix either way; the eigenvector SIGNS follow the order, which is why both sides use the same one.
LMC_HD void JacobiRoundPair(int m, int r, int j, int &p, int &q) {  // pair j (0 <= j < m / 2) of round r (0 <= r < m - 1)
    int a, b;
    if (j == 0) a = m - 1, b = r;
    else
        a = (r + j) % (m - 1), b = (r - j + (m - 1)) % (m - 1);
    p = a < b ? a : b, q = a < b ? b : a;
}
template <class MA, class MV>
LMC_HD void JacobiEigenSymT(int n, MA A, MV V, float *w) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) V[i * n + j] = (i == j) ? 1.0f : 0.0f;
    const int m = (n + 1) & ~1;
    for (int sweep = 0; sweep < 30; sweep++) {
        float off = 0.f, diag = 0.f;
        for (int i = 0; i < n; i++) {
            diag += A[i * n + i] * A[i * n + i];
            for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j];
        }
        if (!(off > 1e-14f * (diag + off))) break;  // also leaves on NaN
        for (int r = 0; r < m - 1; r++) {
            float cs[16], sn[16];
            bool rot[16];
            for (int j = 0; j < m / 2; j++) {  // the angles of the round, all from the matrix as the round finds it
                int p, q;
                JacobiRoundPair(m, r, j, p, q);
                rot[j] = false;
                if (q >= n) continue;  // the bye of an odd n
                const float apq = A[p * n + q];
                if (apq == 0.0f) continue;
                const float app = A[p * n + p], aqq = A[q * n + q];
                const float theta = (aqq - app) / (2.0f * apq);
                const float t = (theta >= 0.0f ? 1.0f : -1.0f) / (fabsf(theta) + sqrtf(theta * theta + 1.0f));
                cs[j] = 1.0f / sqrtf(t * t + 1.0f), sn[j] = t * cs[j];
                rot[j] = true;
            }
            for (int j = 0; j < m / 2; j++) {  // A <- A J, V <- V J (columns p, q of every pair)
                if (!rot[j]) continue;
                int p, q;
                JacobiRoundPair(m, r, j, p, q);
                const float c = cs[j], s = sn[j];
                for (int k = 0; k < n; k++) {
                    const float akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                    const float vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
            for (int j = 0; j < m / 2; j++) {  // A <- J^T A (rows p, q of every pair)
                if (!rot[j]) continue;
                int p, q;
                JacobiRoundPair(m, r, j, p, q);
                const float c = cs[j], s = sn[j];
                for (int k = 0; k < n; k++) {
                    const float apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
            }
        }
    }
    for (int i = 0; i < n; i++) w[i] = A[i * n + i];
    for (int i = 0; i < n - 1; i++) {  // ascending order (selection sort; ties keep their order)
        int m = i;
        for (int j = i + 1; j < n; j++)
            if (w[j] < w[m]) m = j;
        if (m != i) {
            const float tw = w[i];
            w[i] = w[m], w[m] = tw;
            for (int k = 0; k < n; k++) {
                const float tv = V[k * n + i];
                V[k * n + i] = V[k * n + m], V[k * n + m] = tv;
            }
        }
    }
}
LMC_HD void JacobiEigenSym(int n, float *A, float *V, float *w) { JacobiEigenSymT(n, A, V, w); }

// An n x n matrix (row-major, entry (i,j) = word i*n+j) behind a stride: contiguous on the CPU, one word per chain-stride in the
// device's SoA arrays -- the device keeps covL / invCov in HBM and never holds a dense matrix of the Gaussian in private memory.
struct MatRef {
    float *p;
    size_t stride;
    LMC_HD float &operator[](int k) const { return p[(size_t)k * stride]; }
};

// Dense Gaussian of one state: mean[n], covL, invCov (n x n), logDet.
// `hess` is the n x n matrix as the derivative program delivers it (row i at hess[i*n]); its upper triangle is mirrored IN PLACE and it is destroyed
// by the eigen-solve; `work` needs n*n + 4*n floats.
template <class MA>
LMC_HD void ComputeGaussianH2MCT(const H2MCParam &param, int n, float sc, const float *grad, MA hess, float *mean, MatRef covL, MatRef invCov,
                                 float &logDet, float *work) {
    const float sigma = param.sigma, invSigmaSq = 1.0f / (sigma * sigma);
    float hnorm = 0.f;
    for (int i = 0; i < n * n; i++) hnorm += hess[i] * hess[i];
    hnorm = sqrtf(hnorm);
    if (sc <= 1e-15f || hnorm < 0.5f / (sigma * sigma) || !(hnorm == hnorm)) {  // h2mc.cpp:84-92 (NaN cannot occur: the caller zeroes non-finite input)
        for (int i = 0; i < n; i++) {
            mean[i] = 0.f;
            for (int j = 0; j < n; j++) covL[i * n + j] = (i == j) ? sigma : 0.f, invCov[i * n + j] = (i == j) ? invSigmaSq : 0.f;
        }
        logDet = 0.f;
        for (int i = 0; i < n; i++) logDet += llogf(invSigmaSq);
        return;
    }
    MA A = hess;
    float *V = work, *w = work + n * n, *eigenBuff = w + n, *offsetBuff = w + 2 * n, *post = w + 3 * n;
    // Eigen maps the row-major program output as a COLUMN-major matrix (h2mc.cpp:78) and SelfAdjointEigenSolver reads its lower
    // triangle only: entry (r, c), r >= c, of that view is hess[c * n + r], i.e. the UPPER triangle of the rows as delivered.
    for (int i = 0; i < n; i++)
        for (int j = i + 1; j < n; j++) A[j * n + i] = hess[i * n + j];
    JacobiEigenSymT(n, A, V, w);
    for (int i = 0; i < n; i++) eigenBuff[i] = fabsf(w[i]) > 1e-10f ? 1.0f / fabsf(w[i]) : 0.0f;
    for (int i = 0; i < n; i++) {  // offsetBuff = diag(eigenBuff) (V^T grad)
        float dot = 0.f;
        for (int k = 0; k < n; k++) dot += V[k * n + i] * grad[k];
        offsetBuff[i] = eigenBuff[i] * dot;
    }
    for (int i = 0; i < n; i++) {
        float s2 = 1.0f, o = 0.0f;
        if (fabsf(w[i]) > 1e-10f) {
            o = offsetBuff[i];
            if (w[i] > 0.0f) s2 = param.posScaleFactor, o *= param.posOffsetFactor;
            else
                s2 = param.negScaleFactor, o *= param.negOffsetFactor;
        } else {
            s2 = param.L * param.L;
            o = 0.5f * offsetBuff[i] * param.L * param.L;
        }
        eigenBuff[i] *= s2;
        eigenBuff[i] = eigenBuff[i] > 1e-10f ? 1.0f / eigenBuff[i] : 0.0f;
        offsetBuff[i] = o;
    }
    for (int i = 0; i < n; i++) post[i] = eigenBuff[i] + invSigmaSq;
    for (int i = 0; i < n; i++) {
        float m = 0.f;
        for (int k = 0; k < n; k++) m += V[i * n + k] * ((eigenBuff[k] / post[k]) * offsetBuff[k]);
        mean[i] = m;
        for (int j = 0; j < n; j++) {
            float ic = 0.f;
            for (int k = 0; k < n; k++) ic += V[i * n + k] * post[k] * V[j * n + k];
            invCov[i * n + j] = ic;
            covL[i * n + j] = V[i * n + j] * sqrtf(1.0f / post[j]);
        }
    }
    logDet = 0.f;
    for (int i = 0; i < n; i++) logDet += llogf(post[i]);
}

LMC_HD void ComputeGaussianH2MC(const H2MCParam &param, int n, float sc, const float *grad, float *hess, float *mean, MatRef covL, MatRef invCov,
                                float &logDet, float *work) {
    ComputeGaussianH2MCT(param, n, sc, grad, hess, mean, covL, invCov, logDet, work);
}

// gaussian.cpp:24-36 / :38-55, dense branch
LMC_HD float DenseGaussianLogPdf(int n, const float *offset, bool negate, const float *mean, MatRef invCov, float logDet) {
    float logPdf = n * (-0.9189385332046727f);
    logPdf += 0.5f * logDet;
    float q = 0.f;
    for (int i = 0; i < n; i++) {
        float r = 0.f;
        for (int j = 0; j < n; j++) r += invCov[i * n + j] * ((negate ? -offset[j] : offset[j]) - mean[j]);
        q += ((negate ? -offset[i] : offset[i]) - mean[i]) * r;
    }
    logPdf -= 0.5f * q;
    return logPdf;
}
LMC_HD void DenseGaussianMap(int n, const float *z, const float *mean, MatRef covL, float *x) {  // x = covL z + mean
    for (int i = 0; i < n; i++) {
        float r = 0.f;
        for (int j = 0; j < n; j++) r += covL[i * n + j] * z[j];
        x[i] = r + mean[i];
    }
}

}  // namespace lmcd
