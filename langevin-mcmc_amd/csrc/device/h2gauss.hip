// k_h2_gauss, k_h2_sample: the dense proposal Gaussian of an H2MC state and the offset drawn from it, 16 lanes per state (layout: dh2coop.h).
// No contraction (a * b + c stays two roundings, like everywhere outside the Hessian program); division and square root are the hardware's
// approximate reciprocal / rsqrt sequences (H2GAUSS_FLAGS in the Makefile) -- the rotation SEQUENCE and every convention (ascending
// eigenvalues, the order of every sum) are the CPU oracle's serial solver's, so both sides produce the same eigenvectors up to rounding.
#include "dh2coop.h"
#include "dh2mc.h"
#include "kernels.h"

using namespace lmcd;

// ---------------------------------------------------------------------------------------------------------------------------------
// k_h2_gauss: ComputeGaussian(H2MCParam, ...) of /root/reference/src/h2mc.cpp:3-142 for the states of a stage, 16 lanes per state
// (four states of one technique per wave): the symmetric eigen-decomposition (the reference calls Eigen::SelfAdjointEigenSolver --
// third party, absent: parity unpinned) is the cyclic Jacobi iteration (the oracle's serial form: oracle/h2mc_serial.h JacobiEigenSymT) with every rotation's
// 3 n element updates spread over the lanes: same rotation sequence, same arithmetic per element, hence the same eigenvectors (sign
// and order included) as the serial form the CPU oracle runs.  The matrices live in LDS (row stride 17: lanes k = 0..15 reading
// [k][p] hit 16 banks).  Only the two reductions (Frobenius norm of the early-out, off-diagonal norm of the convergence test) are
// summed in another order than the serial code.
// Deliberate deviation: the finite test (mutation_h2mc.h:80-85 IsFinite over the whole vHess) and the Frobenius-norm early-out
// (h2mc.cpp:84 hess.norm()) are evaluated on the matrix SYMMETRISED FROM THE TRIANGLE EIGEN READS -- the only entries h2hess.hip delivers.  The
// reference's programs deliver the full matrix, whose other triangle can differ where chad's adjoint overwrite makes the "Hessian" asymmetric
// (DESIGN.md "chad's adjoint overwrite"): a non-finite or asymmetric entry there would move the reference's iso / dense decision and not ours.
// The oracle (oracle/h2mc_serial.h) keeps the same convention so that the two sides stay comparable; the size of the effect is what
// test_h2mc_per_step_agreement measures (0.6 - 3 % of the chains part per 6 mutations, all causes together).
// Writes the state's Gaussian (AoS, dh2coop.h) into the chain's current (stage 0) or proposal (stage 1) buffer; stage 1 also
// returns px = GaussianLogPdf(-offset, proposalGaussian) (mutation_h2mc.h:104, gaussian.cpp:24-36).
namespace {
constexpr int GS = 17, GW = 16 * GS;
constexpr int G_LDS = 2 * GW + 6 * 16 + 16;  // (the pad holds the round's angles: 8 cos | 8 sin; the pairs live in a table of their own)  // A | V | w | eb | ob | post | grad | tmp | pad: 656 = 16 (mod 32), so the four groups of a wave sit on disjoint LDS banks (640 put all four on the same 16: profiles/r04_f: 250 M conflict cycles per launch)
__device__ __forceinline__ float GroupSum(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    return v;
}

// pair j of round r of an n-player tournament (n even): player n - 1 stays, the others rotate (oracle/h2mc_serial.h JacobiRoundPair)
template <int NN>
__device__ __forceinline__ void RoundPair(int r, int j, int &p, int &q) {
    int a, b;
    if (j == 0) a = NN - 1, b = r;
    else {
        a = r + j, b = r - j + (NN - 1);
        a = a >= NN - 1 ? a - (NN - 1) : a, b = b >= NN - 1 ? b - (NN - 1) : b;
    }
    p = min(a, b), q = max(a, b);
}
// The sweeps of one wave's (up to) four matrices, dimension NN at compile time.  A round = NN / 2 disjoint rotations = one similarity transform:
// lane j < NN / 2 of a group works out pair j's angle from the matrix as the round finds it; then every lane updates ITS row of A and of V
// (columns p, q of every pair); then ITS column of A (rows p, q).  The kernel is bound by its chain of dependent LDS round trips (2.6 k
// instructions per wave-task in 100 k cycles, profiles/r05_x_h2mc_pmc_door.json), so every phase issues ALL its loads before the first use:
// three round trips per round where the row-cyclic form (one rotation at a time, four barriers each) had about four per ROTATION.
template <int NN>
__device__ __forceinline__ void JacobiSweeps(float *A, float *V, float *rc, float *rs, int *rp, int k, bool act, bool run) {
    constexpr int H = NN / 2;
    for (int sweep = 0; sweep < 30; sweep++) {
        float offP = 0.f, diagP = 0.f;
        if (act) {
            diagP = A[k * GS + k] * A[k * GS + k];
            for (int c = k + 1; c < NN; c++) offP += A[k * GS + c] * A[k * GS + c];
        }
        const float off = GroupSum(offP), diag = GroupSum(diagP);
        if (!(off > 1e-14f * (diag + off))) run = false;  // also leaves on NaN
        if (!__any(run)) break;
#pragma unroll 1
        for (int r = 0; r < NN - 1; r++) {
            if (k < H) {
                int p, q;
                RoundPair<NN>(r, k, p, q);
                const float apq = A[p * GS + q], app = A[p * GS + p], aqq = A[q * GS + q];
                const float theta = (aqq - app) / (2.0f * apq);
                const float tt = (theta >= 0.0f ? 1.0f : -1.0f) / (fabsf(theta) + sqrtf(theta * theta + 1.0f));
                const float cs = 1.0f / sqrtf(tt * tt + 1.0f);
                rc[k] = cs, rs[k] = tt * cs, rp[k] = (run && apq != 0.0f) ? 1 : 0;
            }
            __syncthreads();
            const int kk = act ? k : 0;  // idle lanes read row 0 and write nothing
            float cs[H], sn[H], ap[H], aq[H], vp[H], vq[H];
            bool rot[H];
#pragma unroll
            for (int j = 0; j < H; j++) {
                int p, q;
                RoundPair<NN>(r, j, p, q);
                cs[j] = rc[j], sn[j] = rs[j], rot[j] = rp[j] != 0;
                ap[j] = A[kk * GS + p], aq[j] = A[kk * GS + q], vp[j] = V[kk * GS + p], vq[j] = V[kk * GS + q];
            }
            if (act) {
#pragma unroll
                for (int j = 0; j < H; j++) {
                    int p, q;
                    RoundPair<NN>(r, j, p, q);
                    A[k * GS + p] = rot[j] ? cs[j] * ap[j] - sn[j] * aq[j] : ap[j];
                    A[k * GS + q] = rot[j] ? sn[j] * ap[j] + cs[j] * aq[j] : aq[j];
                    V[k * GS + p] = rot[j] ? cs[j] * vp[j] - sn[j] * vq[j] : vp[j];
                    V[k * GS + q] = rot[j] ? sn[j] * vp[j] + cs[j] * vq[j] : vq[j];
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < H; j++) {
                int p, q;
                RoundPair<NN>(r, j, p, q);
                ap[j] = A[p * GS + kk], aq[j] = A[q * GS + kk];
            }
            if (act) {
#pragma unroll
                for (int j = 0; j < H; j++) {
                    int p, q;
                    RoundPair<NN>(r, j, p, q);
                    A[p * GS + k] = rot[j] ? cs[j] * ap[j] - sn[j] * aq[j] : ap[j];
                    A[q * GS + k] = rot[j] ? sn[j] * ap[j] + cs[j] * aq[j] : aq[j];
                }
            }
            __syncthreads();
        }
    }
}

// mean, invCov, covL of the dense Gaussian (h2mc.cpp:130-141) for lane k = column k; returns mean[k].  Dimension at compile time: the lane's own row of V
// and the posterior eigenvalues sit in registers, so a multiply-add of the n x n x n product reads ONE LDS word (V[i][c], the same for the lanes of a
// group: a broadcast) where the run-time loop read three.  Same operations in the same order.
template <int NN>
__device__ __forceinline__ float WriteDense(float *A, const float *V, const float *eb, const float *ob, const float *post, int k, float *G) {
    float pc[NN], vk[NN], wm[NN];
#pragma unroll
    for (int c = 0; c < NN; c++) pc[c] = post[c], vk[c] = V[k * GS + c], wm[c] = (eb[c] / post[c]) * ob[c];
    float m = 0.f;
#pragma unroll
    for (int c = 0; c < NN; c++) m += vk[c] * wm[c];
    G[k] = m;
    const float sq1 = sqrtf(1.0f / post[k]);
#pragma unroll 2
    for (int i = 0; i < NN; i++) {  // lane k = column k of invCov / covL: a row of the AoS record is written by consecutive lanes
        float ic = 0.f;
#pragma unroll
        for (int c = 0; c < NN; c++) ic += V[i * GS + c] * pc[c] * vk[c];
        G[H2_GAUSS_INVCOV + i * NN + k] = ic;
        G[H2_GAUSS_COVL + i * NN + k] = V[i * GS + k] * sq1;
        A[i * GS + k] = ic;  // kept for px
    }
    return m;
}
}  // namespace

__global__ void __launch_bounds__(64) k_h2_gauss(H2Bins bins, int N, const float *__restrict__ hout, H2MCParam param, int expFlags, const int *__restrict__ chainFlags,
                                                  int stage, float *__restrict__ gaussBuf, const float *__restrict__ offsetSoA, float *__restrict__ px) {
    __shared__ float lds[4 * G_LDS];
    __shared__ int pairTab[4 * 8];
    const int lane = threadIdx.x, g = lane >> 4, k = lane & 15;
    float *A = lds + g * G_LDS, *V = A + GW, *w = V + GW, *eb = w + 16, *ob = eb + 16, *post = ob + 16, *grad = post + 16, *tmp = grad + 16;
    float *rc = tmp + 16, *rs = rc + 8;
    int *rp = pairTab + g * 8;
    __shared__ int taskIncl[H2_NBINS];
    const int total = H2BuildTaskTable(bins.count, taskIncl, [](int) { return 4; });
    const float sigma = param.sigma, invSigmaSq = 1.0f / (sigma * sigma);
    for (int wk = blockIdx.x; wk < total; wk += gridDim.x) {
        const int wr = total - 1 - wk;
        const int bin = H2BinOfTask(taskIncl, wr);
        const int t = bin / H2_NSIG;
        const int j = wr - (bin ? taskIncl[bin - 1] : 0);
        const int tCnt = bins.count[bin];
        const int n = H2TechDim(t);
        const int first = 4 * j, nItems = min(4, tCnt - first);
        const bool has = g < nItems;
        const int item = has ? bins.items[bins.start[bin] + first + g] : 0;
        const bool act = has && k < n;
        const float *o = hout + (size_t)item * H2_OUT_WORDS;
        // the triangle Eigen reads (h2mc.cpp:78: the program's rows as a column-major matrix, lower triangle = the UPPER triangle of the rows)
        bool fin = true;
        float sq = 0.f;
        if (act) {
            const float gk = o[k];
            // the row of the symmetrised matrix: all its loads in flight together (a loop of "load, wait, store to LDS" was n dependent round trips)
            float hrow[16];
#pragma unroll
            for (int c = 0; c < 16; c++) {
                const int cc = min(c, n - 1);
                hrow[c] = cc >= k ? o[H2_OUT_HESS + k * n + cc] : o[H2_OUT_HESS + cc * n + k];
            }
            fin = isfinite(gk);
            grad[k] = gk;
#pragma unroll
            for (int c = 0; c < 16; c++)
                if (c < n) {
                    const float h = hrow[c];
                    fin = fin && isfinite(h);
                    A[k * GS + c] = h;
                    V[k * GS + c] = (k == c) ? 1.0f : 0.0f;
                    sq += h * h;
                }
        }
        const unsigned long long finMask = __ballot(fin || !act);
        const bool allFinite = ((finMask >> (16 * g)) & 0xffffull) == 0xffffull;  // mutation_h2mc.h:80-85: any non-finite entry zeroes gradient and Hessian
        const float hnorm = sqrtf(GroupSum(act ? sq : 0.f));
        const bool iso = !allFinite || LMC_EXP(expFlags, 32) || hnorm < 0.5f / (sigma * sigma) || !(hnorm == hnorm);  // h2mc.cpp:84-92
        bool run = has && !iso;
        __syncthreads();
        // ---- cyclic Jacobi, round-robin order: the rotation sequence of oracle/h2mc_serial.h JacobiEigenSymT
        switch (n) {
            case 4: JacobiSweeps<4>(A, V, rc, rs, rp, k, act, run); break;
            case 6: JacobiSweeps<6>(A, V, rc, rs, rp, k, act, run); break;
            case 8: JacobiSweeps<8>(A, V, rc, rs, rp, k, act, run); break;
            case 10: JacobiSweeps<10>(A, V, rc, rs, rp, k, act, run); break;
            case 12: JacobiSweeps<12>(A, V, rc, rs, rp, k, act, run); break;
            case 14: JacobiSweeps<14>(A, V, rc, rs, rp, k, act, run); break;
            default: JacobiSweeps<16>(A, V, rc, rs, rp, k, act, run); break;
        }
        float *G = gaussBuf + ((((chainFlags[item] & F_GSEL) != 0) != (stage != 0)) ? (size_t)N * H2_GAUSS_AOS : 0) + (size_t)item * H2_GAUSS_AOS;
        float logDet = 0.f, meanK = 0.f;
        if (iso) {
            for (int i = 0; i < n; i++) logDet += llogf(invSigmaSq);
            if (act) G[k] = 0.f;
            if (has && k == 0) G[H2_GAUSS_LOGDET] = logDet, G[H2_GAUSS_LOGDET + 1] = (float)H2K_ISO_EARLYOUT;
        }
        // eigenvalues in ascending order (selection sort; ties keep their order), eigenvectors = the columns of V
        if (act) w[k] = A[k * GS + k];
        __syncthreads();
        for (int i = 0; i < n - 1; i++) {
            int m = i;
            for (int c = i + 1; c < n; c++)
                if (w[c] < w[m]) m = c;
            const float wi = w[i], wm = w[m];
            const float vi = V[k * GS + i], vm = V[k * GS + m];
            __syncthreads();
            if (m != i && act) {
                if (k == 0) w[i] = wm, w[m] = wi;
                V[k * GS + i] = vm, V[k * GS + m] = vi;
            }
            __syncthreads();
        }
        if (act) {  // per eigenvalue: variance / offset remap, h2mc.cpp:100-128
            const float wk_ = w[k];
            float e = fabsf(wk_) > 1e-10f ? 1.0f / fabsf(wk_) : 0.0f;
            float dot = 0.f;
            for (int c = 0; c < n; c++) dot += V[c * GS + k] * grad[c];
            const float ofs = e * dot;
            float s2 = 1.0f, oo = 0.0f;
            if (fabsf(wk_) > 1e-10f) {
                oo = ofs;
                if (wk_ > 0.0f) s2 = param.posScaleFactor, oo *= param.posOffsetFactor;
                else
                    s2 = param.negScaleFactor, oo *= param.negOffsetFactor;
            } else {
                s2 = param.L * param.L;
                oo = 0.5f * ofs * param.L * param.L;
            }
            e *= s2;
            e = e > 1e-10f ? 1.0f / e : 0.0f;
            eb[k] = e, ob[k] = oo, post[k] = e + invSigmaSq;
            tmp[k] = llogf(e + invSigmaSq);
        }
        __syncthreads();
        if (!iso) {
            for (int i = 0; i < n; i++) logDet += tmp[i];
            if (act) {
                switch (n) {
                    case 4: meanK = WriteDense<4>(A, V, eb, ob, post, k, G); break;
                    case 6: meanK = WriteDense<6>(A, V, eb, ob, post, k, G); break;
                    case 8: meanK = WriteDense<8>(A, V, eb, ob, post, k, G); break;
                    case 10: meanK = WriteDense<10>(A, V, eb, ob, post, k, G); break;
                    case 12: meanK = WriteDense<12>(A, V, eb, ob, post, k, G); break;
                    case 14: meanK = WriteDense<14>(A, V, eb, ob, post, k, G); break;
                    default: meanK = WriteDense<16>(A, V, eb, ob, post, k, G); break;
                }
                if (k == 0) G[H2_GAUSS_LOGDET] = logDet, G[H2_GAUSS_LOGDET + 1] = (float)H2K_DENSE;
            }
        }
        __syncthreads();
        if (stage != 0) {  // px = GaussianLogPdf(-offset, this Gaussian), gaussian.cpp:24-36
            if (act) tmp[k] = -offsetSoA[(size_t)k * N + item] - meanK;
            __syncthreads();
            if (act) {
                float r = 0.f;
                if (iso) r = invSigmaSq * tmp[k];
                else
                    for (int c = 0; c < n; c++) r += A[k * GS + c] * tmp[c];
                w[k] = r;
            }
            __syncthreads();
            if (has && k == 0) {
                float qf = 0.f;
                for (int i = 0; i < n; i++) qf += tmp[i] * w[i];
                float logPdf = n * (-0.9189385332046727f);
                logPdf += 0.5f * logDet;
                logPdf -= 0.5f * qf;
                px[item] = logPdf;
            }
        }
        __syncthreads();
    }
}

// k_h2_sample: the proposal offset of an H2MC step and its density, GenerateSample / GaussianLogPdf (gaussian.cpp:24-55) with the dense
// current Gaussian: offset = covL z + mean, py = log N(offset; mean, invCov^-1).  16 lanes per chain (lane = row), over the whole
// small-step list; chains that take another kind of step this time (uniform mixing, isotropic Gaussian, no derivative program) were
// given their offset by k_h2_begin and are skipped (stepKind != 1).
__global__ void __launch_bounds__(64) k_h2_sample(const int *__restrict__ list, const int *__restrict__ listCount, int N, const int *__restrict__ chainFlags,
                                                   const unsigned char *__restrict__ stepKind, const float *__restrict__ curContrib, const float *__restrict__ gaussBuf,
                                                   float sigma, float *__restrict__ offsetSoA /* in: z, out: offset */, float *__restrict__ py) {
    __shared__ float lds[4 * (2 * 256 + 48)];
    const int lane = threadIdx.x, g = lane >> 4, k = lane & 15;
    float *CL = lds + g * (2 * 256 + 48), *IC = CL + 256, *z = IC + 256, *x = z + 16, *mean = x + 16;
    const int total = *listCount;
    const float invSigmaSq = 1.0f / (sigma * sigma);
    for (int base = blockIdx.x * 4; base < total; base += gridDim.x * 4) {
        const int e = base + g;
        const int i = e < total ? list[e] : -1;
        const bool has = i >= 0 && stepKind[i] == 1;
        int n = 0;
        bool iso = false;
        const float *G = gaussBuf;
        if (has) {
            const int c = __float_as_int(curContrib[i]), l = __float_as_int(curContrib[(size_t)N + i]);
            n = 2 * max(c + l - 1, 2);
            G = gaussBuf + ((chainFlags[i] & F_GSEL) ? (size_t)N * H2_GAUSS_AOS : 0) + (size_t)i * H2_GAUSS_AOS;
            iso = G[H2_GAUSS_LOGDET + 1] != 0.0f;  // an isotropic record (H2K_ISO_*): sigma I, mean 0, only logDet is stored
            if (!iso)
                for (int q = k; q < n * n; q += 16) CL[q] = G[H2_GAUSS_COVL + q], IC[q] = G[H2_GAUSS_INVCOV + q];
            if (k < n) z[k] = offsetSoA[(size_t)k * N + i], mean[k] = iso ? 0.0f : G[k];
        }
        __syncthreads();
        const bool act = has && k < n;
        if (act) {
            float r = 0.f;
            if (iso) r = sigma * z[k];
            else
                for (int c = 0; c < n; c++) r += CL[k * n + c] * z[c];
            const float xk = r + mean[k];
            x[k] = xk;
            offsetSoA[(size_t)k * N + i] = xk;
        }
        __syncthreads();
        if (act) {
            float r = 0.f;
            if (iso) r = invSigmaSq * (x[k] - mean[k]);
            else
                for (int c = 0; c < n; c++) r += IC[k * n + c] * (x[c] - mean[c]);
            z[k] = r;
        }
        __syncthreads();
        if (has && k == 0) {
            float qf = 0.f;
            for (int c = 0; c < n; c++) qf += (x[c] - mean[c]) * z[c];
            float logPdf = n * (-0.9189385332046727f);
            logPdf += 0.5f * G[H2_GAUSS_LOGDET];
            logPdf -= 0.5f * qf;
            py[i] = logPdf;
        }
        __syncthreads();
    }
}

void LaunchH2Gauss(const H2Bins &bins, int N, const float *hout, const H2MCParam &param, int expFlags, const int *chainFlags, int stage, float *gaussBuf,
                   const float *offsetSoA, float *px, int gridBlocks, hipStream_t s) {
    hipLaunchKernelGGL(k_h2_gauss, dim3(gridBlocks), dim3(64), 0, s, bins, N, hout, param, expFlags, chainFlags, stage, gaussBuf, offsetSoA, px);
}
void LaunchH2Sample(const int *list, const int *listCount, int N, const int *chainFlags, const unsigned char *stepKind, const float *curContrib, const float *gaussBuf,
                    float sigma, float *offsetSoA, float *py, int gridBlocks, hipStream_t s) {
    hipLaunchKernelGGL(k_h2_sample, dim3(gridBlocks), dim3(64), 0, s, list, listCount, N, chainFlags, stepKind, curContrib, gaussBuf, sigma, offsetSoA, py);
}
