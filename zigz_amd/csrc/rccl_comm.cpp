// RCCL transport for the sharded provers (include/zigz_hip.h: zigz_rccl_comm_*): the exchange hook for hosts whose ranks sit
// on different GPUs of an xGMI node (or on several nodes) and do not want torch in the loop -- a Zig host calls this through
// the same C ABI as everything else.
//
// What is exchanged is small (16 B .. 64 KiB) and host-resident -- it was read back for the SHA3 transcript first -- so the
// hook form (zigz_rccl_allgather, a zigz_allgather_fn) stages it through one pinned + one device buffer: H2D, ncclAllGather
// on the communicator's own stream, D2H, one stream synchronisation.  For the one exchange whose operand is ON the device --
// the partial block sums of a row-sharded sumcheck stage -- zigz_rccl_allreduce_u64_dev reduces them in place on the
// caller's stream before they are read back: the north-star's "RCCL all-reduce of the round-polynomial sums", once per
// radix stage (k <= 10 rounds) instead of once per round (src/proofs/sumcheck_prover.zig:50-77).
//
// librccl.so is 570 MB: it is loaded on first use (dlopen), never at library load, and its absence is a loud error.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <string.h>
#include <time.h>

#include <mutex>
#include <new>

#include "zigz_hip.h"

namespace {
struct Api {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
Api g_api;
std::once_flag g_once;

void load_api() {
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        g_api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_api.lib) break;
    }
    if (!g_api.lib) return;
#define ZK_SYM(field, name) *(void **)(&g_api.field) = dlsym(g_api.lib, name)
    ZK_SYM(GetUniqueId, "ncclGetUniqueId");
    ZK_SYM(CommInitRank, "ncclCommInitRank");
    ZK_SYM(CommDestroy, "ncclCommDestroy");
    ZK_SYM(CommAbort, "ncclCommAbort");
    ZK_SYM(AllGather, "ncclAllGather");
    ZK_SYM(AllReduce, "ncclAllReduce");
    ZK_SYM(GetErrorString, "ncclGetErrorString");
#undef ZK_SYM
    g_api.ok = g_api.GetUniqueId && g_api.CommInitRank && g_api.CommDestroy && g_api.AllGather && g_api.AllReduce;
}
const Api *api() {
    std::call_once(g_once, load_api);
    return g_api.ok ? &g_api : nullptr;
}
}  // namespace

struct zigz_rccl_comm {
    int device, rank, world;
    size_t max_bytes;
    ncclComm_t comm;
    hipStream_t stream;
    uint8_t *d_send, *d_recv;  // max_bytes, world * max_bytes
    uint8_t *h_pin;            // (1 + world) * max_bytes, page-locked
    double timeout_s;          // how long a wait for a collective may take before the communicator is aborted
    bool dead;                 // aborted (a wait ran out, or a collective could not be enqueued): every later call fails at once
};

// RCCL has no timeout of its own: a rank that never enters a collective leaves its peers' kernels spinning for ever.  Every
// wait behind a collective therefore polls the stream with a deadline; when it passes, the communicator is aborted (the
// device-side kernels see the abort flag and exit) and the call fails with a nonzero code instead of hanging.
static void comm_abort(zigz_rccl_comm *c) {
    if (c->dead) return;
    c->dead = true;
    const Api *a = api();
    if (a && a->CommAbort && c->comm) {
        (void)a->CommAbort(c->comm);
        c->comm = nullptr;  // (CommAbort frees it)
    }
}
static int wait_stream(zigz_rccl_comm *c, hipStream_t s) {
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    long ns = 2000;
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e == hipSuccess) return 0;
        if (e != hipErrorNotReady) {
            (void)hipGetLastError();
            comm_abort(c);
            return 6;
        }
        timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > c->timeout_s) {
            comm_abort(c);
            (void)hipStreamSynchronize(s);  // (the aborted kernels drain)
            return 7;
        }
        timespec ts{0, ns};
        nanosleep(&ts, nullptr);
        if (ns < 100000) ns += ns / 2;
    }
}

static_assert(ZIGZ_RCCL_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id crosses the ABI as plain bytes");

extern "C" zigz_status zigz_rccl_unique_id(uint8_t id[ZIGZ_RCCL_UNIQUE_ID_BYTES]) {
    const Api *a = api();
    if (!a) return ZIGZ_ERR_NO_DEVICE;
    if (!id) return ZIGZ_ERR_INVALID_ARGUMENT;
    ncclUniqueId u;
    if (a->GetUniqueId(&u) != ncclSuccess) return ZIGZ_ERR_COMM;
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return ZIGZ_OK;
}

extern "C" void zigz_rccl_comm_destroy(zigz_rccl_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm && api()) (void)api()->CommDestroy(c->comm);
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" zigz_status zigz_rccl_comm_create(int device, const uint8_t id[ZIGZ_RCCL_UNIQUE_ID_BYTES], int rank, int world,
                                             size_t max_bytes, zigz_rccl_comm **out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world || max_bytes == 0) return ZIGZ_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    const Api *a = api();
    if (!a) return ZIGZ_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return ZIGZ_ERR_NO_DEVICE;
    zigz_rccl_comm *c = new (std::nothrow) zigz_rccl_comm();
    if (!c) return ZIGZ_ERR_OUT_OF_MEMORY;
    memset(c, 0, sizeof(*c));
    c->device = device;
    c->rank = rank;
    c->world = world;
    c->max_bytes = (max_bytes + 63) & ~(size_t)63;
    c->timeout_s = 120.0;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void **)&c->d_send, c->max_bytes) != hipSuccess ||
        hipMalloc((void **)&c->d_recv, c->max_bytes * (size_t)world) != hipSuccess ||
        hipHostMalloc((void **)&c->h_pin, c->max_bytes * (size_t)(world + 1), hipHostMallocDefault) != hipSuccess) {
        zigz_rccl_comm_destroy(c);
        return ZIGZ_ERR_OUT_OF_MEMORY;
    }
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    if (a->CommInitRank(&c->comm, world, u, rank) != ncclSuccess) {
        c->comm = nullptr;
        zigz_rccl_comm_destroy(c);
        return ZIGZ_ERR_COMM;
    }
    *out = c;
    return ZIGZ_OK;
}

// zigz_allgather_fn: user = zigz_rccl_comm*.  Host buffers in and out.
extern "C" int zigz_rccl_allgather(void *user, const void *send, size_t bytes, void *recv) {
    zigz_rccl_comm *c = (zigz_rccl_comm *)user;
    const Api *a = api();
    if (!c || !a || !send || !recv || bytes == 0 || bytes > c->max_bytes) return 1;
    if (c->dead) return 8;
    if (hipSetDevice(c->device) != hipSuccess) return 2;
    memcpy(c->h_pin, send, bytes);
    uint8_t *h_recv = c->h_pin + c->max_bytes;
    if (hipMemcpyAsync(c->d_send, c->h_pin, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) return 3;
    if (a->AllGather(c->d_send, c->d_recv, bytes, ncclInt8, c->comm, c->stream) != ncclSuccess) {
        comm_abort(c);  // (the peers that did enqueue theirs run into their own deadline)
        return 4;
    }
    if (hipMemcpyAsync(h_recv, c->d_recv, bytes * (size_t)c->world, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return 5;
    if (const int w = wait_stream(c, c->stream)) return w;
    memcpy(recv, h_recv, bytes * (size_t)c->world);
    return 0;
}

// sum over the ranks of n u64 words, host buffers in and out (exact integer sums: the partial round-polynomial sums)
extern "C" int zigz_rccl_allreduce_u64(zigz_rccl_comm *c, const uint64_t *send, size_t n, uint64_t *recv) {
    const Api *a = api();
    if (!c || !a || !send || !recv || n == 0 || n * 8 > c->max_bytes) return 1;
    if (c->dead) return 8;
    if (hipSetDevice(c->device) != hipSuccess) return 2;
    memcpy(c->h_pin, send, n * 8);
    if (hipMemcpyAsync(c->d_send, c->h_pin, n * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess) return 3;
    if (a->AllReduce(c->d_send, c->d_send, n, ncclUint64, ncclSum, c->comm, c->stream) != ncclSuccess) {
        comm_abort(c);
        return 4;
    }
    if (hipMemcpyAsync(c->h_pin, c->d_send, n * 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return 5;
    if (const int w = wait_stream(c, c->stream)) return w;
    memcpy(recv, c->h_pin, n * 8);
    return 0;
}

// the same reduction on words that are already in HBM, in place, on the caller's stream (no staging, no synchronisation)
extern "C" int zigz_rccl_allreduce_u64_dev(zigz_rccl_comm *c, uint64_t *d_words, size_t n, void *hip_stream) {
    const Api *a = api();
    if (!c || !a || !d_words || n == 0) return 1;
    if (c->dead) return 8;
    if (hipSetDevice(c->device) != hipSuccess) return 2;
    if (a->AllReduce(d_words, d_words, n, ncclUint64, ncclSum, c->comm, (hipStream_t)hip_stream) != ncclSuccess) {
        comm_abort(c);
        return 4;
    }
    return 0;
}

// The wait that belongs to zigz_rccl_allreduce_u64_dev: until `hip_stream` has drained or the communicator's deadline passes
// (then the communicator is aborted and the call returns nonzero -- the caller's stream is usable again, the comm is not).
extern "C" int zigz_rccl_stream_wait(zigz_rccl_comm *c, void *hip_stream) {
    if (!c) return 1;
    if (hipSetDevice(c->device) != hipSuccess) return 2;
    return wait_stream(c, (hipStream_t)hip_stream);
}
extern "C" void zigz_rccl_comm_set_timeout(zigz_rccl_comm *c, double seconds) {
    if (c && seconds > 0) c->timeout_s = seconds;
}
extern "C" void zigz_rccl_comm_abort(zigz_rccl_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    comm_abort(c);
}

extern "C" int zigz_rccl_comm_rank(const zigz_rccl_comm *c) { return c ? c->rank : -1; }
extern "C" int zigz_rccl_comm_world(const zigz_rccl_comm *c) { return c ? c->world : 0; }
