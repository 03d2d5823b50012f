// rccl_double.cpp — TEST DOUBLE for the six RCCL entry points librt_is.so binds (roundtrip_amd/csrc/rt_is.hip,
// rccl_load): ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllGather, ncclAllReduce, ncclGetErrorString.
//
// Why: RCCL refuses two ranks on one GPU, and the test boxes have one.  With RT_RCCL_LIB pointing here, the rt_comm_*
// entry points, roundtrip_amd/dist.py and bench.py run their N = 2 control flow unchanged — rendezvous, communicator,
// device-pointer gathers sized per * world, max-reduce, barrier — on ranks that share that GPU.  It says nothing about
// RCCL itself (the one-rank test and the driver's multi-GPU node cover the real library); it is test infrastructure and
// no product file names it.
//
// How: the ranks of a communicator meet in a POSIX shared-memory object named inside the 128-byte unique id; a
// collective is  wait for the stream -> copy my block into my slot -> barrier -> read every slot -> barrier.
// Synchronous where RCCL is stream-ordered: stricter, never weaker.  Every wait has a deadline, so a rank that dies
// makes the others fail with ncclSystemError instead of hanging a GPU box.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>

namespace {
constexpr int MAX_RANKS = 16;
constexpr size_t SLOT_BYTES = 4u << 20;      // the largest block a rank may contribute to one collective
constexpr int WAIT_SECONDS = 120;

struct Shared {
    std::atomic<uint32_t> arrived;
    std::atomic<uint32_t> generation;
    unsigned char pad[4096 - 2 * sizeof(std::atomic<uint32_t>)];
    unsigned char slots[MAX_RANKS][SLOT_BYTES];
};
}  // namespace

struct ncclComm {
    int rank, nranks;
    Shared* sh;
};

namespace {
bool barrier(ncclComm* c) {
    Shared* sh = c->sh;
    const uint32_t gen = sh->generation.load();
    if (sh->arrived.fetch_add(1) + 1 == (uint32_t)c->nranks) {
        sh->arrived.store(0);
        sh->generation.fetch_add(1);
        return true;
    }
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(WAIT_SECONDS);
    while (sh->generation.load() == gen) {
        if (std::chrono::steady_clock::now() > deadline) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    return true;
}

size_t type_bytes(ncclDataType_t t) {
    switch (t) {
        case ncclDouble: case ncclInt64: case ncclUint64: return 8;
        case ncclFloat: case ncclInt32: case ncclUint32: return 4;
        case ncclInt8: case ncclUint8: return 1;
        default: return 0;
    }
}

// my block -> my slot, once the stream's earlier work (the kernels that produce it) has finished
ncclResult_t publish(ncclComm* c, const void* send, size_t bytes, hipStream_t stream) {
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(c->sh->slots[c->rank], send, bytes, hipMemcpyDefault) != hipSuccess) return ncclUnhandledCudaError;
    return barrier(c) ? ncclSuccess : ncclSystemError;
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof *id);
    const long long now = std::chrono::steady_clock::now().time_since_epoch().count();
    snprintf(id->internal, sizeof id->internal, "/rt-comm-double-%d-%llx", (int)getpid(), now);
    const int fd = shm_open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    const int rc = ftruncate(fd, sizeof(Shared));      // zero-filled: arrived = generation = 0; pages appear on first touch
    close(fd);
    return rc == 0 ? ncclSuccess : ncclSystemError;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    id.internal[sizeof id.internal - 1] = 0;
    const int fd = shm_open(id.internal, O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    void* p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    ncclComm* c = new ncclComm{rank, nranks, (Shared*)p};
    const bool met = barrier(c);
    if (rank == 0) shm_unlink(id.internal);            // every rank has it mapped (or the wait ran out): drop the name
    if (!met) { munmap(p, sizeof(Shared)); delete c; return ncclSystemError; }
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    if (!comm) return ncclSuccess;
    munmap(comm->sh, sizeof(Shared));
    delete comm;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t type, ncclComm_t comm,
                           hipStream_t stream) {
    const size_t bytes = count * type_bytes(type);
    if (!comm || !send || !recv || bytes == 0 || bytes > SLOT_BYTES) return ncclInvalidArgument;
    ncclResult_t r = publish(comm, send, bytes, stream);
    if (r != ncclSuccess) return r;
    for (int k = 0; k < comm->nranks; ++k)
        if (hipMemcpy((char*)recv + (size_t)k * bytes, comm->sh->slots[k], bytes, hipMemcpyDefault) != hipSuccess)
            return ncclUnhandledCudaError;
    return barrier(comm) ? ncclSuccess : ncclSystemError;     // nobody reuses a slot before everybody has read it
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t type, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t stream) {
    const size_t bytes = count * sizeof(double);
    if (!comm || !send || !recv || count == 0 || bytes > SLOT_BYTES) return ncclInvalidArgument;
    if (type != ncclDouble || (op != ncclMax && op != ncclSum)) return ncclInvalidArgument;
    ncclResult_t r = publish(comm, send, bytes, stream);
    if (r != ncclSuccess) return r;
    double* acc = new double[count];
    memcpy(acc, comm->sh->slots[0], bytes);
    for (int k = 1; k < comm->nranks; ++k) {
        const double* v = (const double*)comm->sh->slots[k];
        for (size_t i = 0; i < count; ++i) acc[i] = op == ncclMax ? (v[i] > acc[i] ? v[i] : acc[i]) : acc[i] + v[i];
    }
    const hipError_t e = hipMemcpy(recv, acc, bytes, hipMemcpyDefault);
    delete[] acc;
    if (e != hipSuccess) return ncclUnhandledCudaError;
    return barrier(comm) ? ncclSuccess : ncclSystemError;
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "comm double: HIP call failed";
        case ncclSystemError: return "comm double: shared memory or a wait for the other ranks failed";
        case ncclInvalidArgument: return "comm double: invalid argument";
        default: return "comm double: error";
    }
}

}  // extern "C"
