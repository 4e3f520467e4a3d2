// debugging aid: launches the second-order path program on synthetic inputs and reports the HIP status
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "pathfunc.h"
using namespace lmcd;
template <class In>
__device__ __noinline__ void HessDev(int c, int l, const float *primary, const float *scene, const In &vp, float *logLum, float *grad, float *hess) {
    PathFuncHessN<NN>(c, l, primary, scene, vp, logLum, grad, hess);
}
__global__ void k(int c, int l, int n, const float *primarySoA, const float *scene, const float *vertSoA, float *hessSoA) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float primary[17];
    for (int k2 = 0; k2 < 17; k2++) primary[k2] = primarySoA[(size_t)k2 * n + i];
    StridedIn vin{vertSoA + i, (size_t)n};
    float ll, g[16], h[256];
    for (int k2 = 0; k2 < 256; k2++) h[k2] = 0.f;
    HessDev(c, l, primary, scene, vin, &ll, g, h);
    for (int k2 = 0; k2 < 256; k2++) hessSoA[(size_t)k2 * n + i] = h[k2];
}
// private-memory probe: W floats of scratch per lane, indexed at run time
template <int W>
__global__ void big(const int *idx, float *out) {
    float a[W];
    for (int k = 0; k < W; k++) a[k] = (float)(k + threadIdx.x);
    float s = 0.f;
    for (int k = 0; k < 64; k++) s += a[(idx[k] + k * 37) % W];
    out[threadIdx.x] = s;
}
template <int W>
void probe(const int *didx, float *dout) {
    hipLaunchKernelGGL(big<W>, dim3(4), dim3(64), 0, 0, didx, dout);
    hipError_t e = hipDeviceSynchronize();
    printf("scratch %d bytes/lane -> %s\n", W * 4, hipGetErrorString(e));
    fflush(stdout);
}
int main() {
    {
        int *didx;
        float *dout;
        hipMalloc(&didx, 64 * 4), hipMalloc(&dout, 64 * 4);
        hipMemset(didx, 0, 64 * 4);
        probe<2048>(didx, dout), probe<3500>(didx, dout), probe<4000>(didx, dout), probe<4200>(didx, dout), probe<5000>(didx, dout), probe<8000>(didx, dout);
    }
    const int n = 64, V = 1000;
    std::vector<float> prim(17 * n), scene(38, 0.f), vert((size_t)V * n);
    for (auto &x : prim) x = 0.1f + 0.8f * (rand() / (float)RAND_MAX);
    for (auto &x : vert) x = 0.1f + 0.8f * (rand() / (float)RAND_MAX);
    for (int k2 = 0; k2 < 38; k2++) scene[k2] = 0.3f + 0.01f * k2;
    float *dp, *ds, *dv, *dh;
    hipMalloc(&dp, prim.size() * 4), hipMalloc(&ds, 38 * 4), hipMalloc(&dv, vert.size() * 4), hipMalloc(&dh, 256 * n * 4);
    hipMemcpy(dp, prim.data(), prim.size() * 4, hipMemcpyHostToDevice), hipMemcpy(ds, scene.data(), 38 * 4, hipMemcpyHostToDevice);
    hipMemcpy(dv, vert.data(), vert.size() * 4, hipMemcpyHostToDevice);
    for (int c = 2; c <= 7; c++)
        for (int l = 0; l <= 2; l++) {
            if (c + l < 3 || c + l > 9) continue;
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, c, l, n, dp, ds, dv, dh);
            hipError_t e = hipDeviceSynchronize();
            printf("(%d,%d) -> %s\n", c, l, hipGetErrorString(e));
            fflush(stdout);
        }
    return 0;
}
