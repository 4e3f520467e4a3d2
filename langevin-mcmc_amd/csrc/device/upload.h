// Small host -> device transfers as a KERNEL that reads pinned host memory, for transfers that sit between kernels of a stream inside a step.
// A hipMemcpyAsync there goes to a DMA engine: every kernel -> copy -> kernel edge of the stream is then a hand-over between engines through the
// runtime's signal handler, 0.1 ms apiece -- the four tree uploads and the struct refresh of the step in which two gradient caches become ready
// (context.cpp CacheApplyFinish) kept the next step waiting 0.5 ms behind the last grid kernel (profiles/r06_aq_fill_phase_trace.txt, step 22).
// As a kernel the transfer is one more packet of the same queue.  (kernels.hip; not part of kernels.h, which the step kernels include.)
#pragma once
#include <hip/hip_runtime.h>

namespace lmcd {
constexpr int UPLOAD_MAX_SEGMENTS = 8;
struct UploadSegments {
    int count = 0;
    unsigned *dst[UPLOAD_MAX_SEGMENTS];        // device
    const unsigned *src[UPLOAD_MAX_SEGMENTS];  // pinned, mapped host memory (hipHostMalloc)
    int words[UPLOAD_MAX_SEGMENTS];            // 32-bit words
    void Add(void *d, const void *s, size_t bytes) {
        dst[count] = static_cast<unsigned *>(d), src[count] = static_cast<const unsigned *>(s), words[count] = (int)(bytes / 4);
        count++;
    }
};
}  // namespace lmcd
void LaunchUploadSegments(const lmcd::UploadSegments &U, hipStream_t s);
