// TEST HELPER: the six RCCL entry points the library binds (host/context.cpp GetRccl), implemented over POSIX shared memory between PROCESSES of one host:
// every collective drains its stream, copies the rank's send buffer to its slot of the segment, meets the other ranks at a barrier, and reads the slots in
// rank order.  Not a transport -- a stand-in that lets the multi-process rank code path of the library (sharded MLTInit's all-gathers, the per-step all-gather
// of the cache pushes, the film all-reduce, the driver's scalar all-reduces and barriers) run with several ranks on the ONE GPU of the test tier, where real
// RCCL refuses two ranks per device.  Selected with LMC_RCCL_LIB=<this library>.  Reductions run in rank order on the host (float / double, sum / max / min).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {
constexpr size_t SLOT = (size_t)256 << 20;  // bytes per rank (sparse until touched)
struct Header {
    std::atomic<int> arrived, gen;
};
struct Comm {
    int rank, n;
    Header *hd;
    char *data;
    size_t bytes;
    std::string name;
};
void Barrier(Comm *c) {
    const int g = c->hd->gen.load();
    if (c->hd->arrived.fetch_add(1) + 1 == c->n) {
        c->hd->arrived.store(0);
        c->hd->gen.fetch_add(1);
    } else {
        while (c->hd->gen.load() == g) sched_yield();
    }
}
size_t SizeOf(ncclDataType_t t) { return t == ncclFloat64 || t == ncclInt64 || t == ncclUint64 ? 8 : t == ncclFloat32 || t == ncclInt32 || t == ncclUint32 ? 4 : t == ncclFloat16 ? 2 : 1; }
template <class T>
void Reduce(T *acc, const T *x, size_t n, ncclRedOp_t op) {
    for (size_t i = 0; i < n; i++) acc[i] = op == ncclSum ? acc[i] + x[i] : op == ncclMax ? (x[i] > acc[i] ? x[i] : acc[i]) : (x[i] < acc[i] ? x[i] : acc[i]);
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    memset(id, 0, sizeof(*id));
    FILE *f = fopen("/dev/urandom", "rb");
    if (!f || fread(id->internal, 1, 16, f) != 16) return ncclSystemError;
    fclose(f);
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    char name[64] = "/lmc_rccl_stub_";
    for (int k = 0; k < 16; k++) snprintf(name + strlen(name), 3, "%02x", (unsigned char)id.internal[k]);
    const size_t bytes = 4096 + SLOT * (size_t)nranks;
    int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) return ncclSystemError;
    void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return ncclSystemError;
    Comm *c = new Comm{rank, nranks, reinterpret_cast<Header *>(m), reinterpret_cast<char *>(m) + 4096, bytes, name};  // a fresh segment is zero-filled: arrived = gen = 0
    *comm = reinterpret_cast<ncclComm_t>(c);
    Barrier(c);
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c) return ncclSuccess;
    if (c->rank == 0) shm_unlink(c->name.c_str());
    munmap(c->hd, c->bytes);
    delete c;
    return ncclSuccess;
}
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "rccl_stub error (shared memory, size or HIP)"; }
ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t stream) {
    Comm *c = reinterpret_cast<Comm *>(comm);
    const size_t bytes = count * SizeOf(t);
    if (bytes > SLOT) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(c->data + SLOT * c->rank, send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    Barrier(c);
    for (int r = 0; r < c->n; r++)
        if (hipMemcpy((char *)recv + bytes * r, c->data + SLOT * r, bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    Barrier(c);  // nobody overwrites its slot before everybody has read it
    return ncclSuccess;
}
ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    Comm *c = reinterpret_cast<Comm *>(comm);
    const size_t bytes = count * SizeOf(t);
    if (bytes > SLOT || (t != ncclFloat32 && t != ncclFloat64) || (op != ncclSum && op != ncclMax && op != ncclMin)) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(c->data + SLOT * c->rank, send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    Barrier(c);
    std::vector<char> acc(c->data, c->data + bytes);  // rank 0's contribution, then the others in rank order: every rank computes the same sum
    for (int r = 1; r < c->n; r++) {
        if (t == ncclFloat32) Reduce(reinterpret_cast<float *>(acc.data()), reinterpret_cast<const float *>(c->data + SLOT * r), count, op);
        else
            Reduce(reinterpret_cast<double *>(acc.data()), reinterpret_cast<const double *>(c->data + SLOT * r), count, op);
    }
    if (hipMemcpy(recv, acc.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    Barrier(c);
    return ncclSuccess;
}
}
