// Gradient-bucket all-reduce over RCCL / xGMI behind the C ABI
// (dvsof_allreduce_bucket).
//
// The reference has no collective at all (single process);
// this is the data-parallel exchange of the build: every rank averages each
// gradient bucket in place as soon as the backward has produced it, on a side
// stream (parallel.GradReducer drives it).  The Python path normally uses
// torch.distributed's process group (backend 'nccl' = RCCL); these entry
// points give a binding WITHOUT torch.distributed the same exchange.
//
// librccl is opened lazily (dlopen): the library loads, and every other entry
// point works, on a machine without RCCL.
#include "common.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

namespace {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclCommCount) comm_count = nullptr;
    bool ok = false;
};

// what dvsof_comm_create / dvsof_comm_create_loopback hand out
struct Comm {
    ncclComm_t c = nullptr;
    int world = 1;
    bool loopback = false;
    int delay_us = 0;
    unsigned long long calls = 0, elements = 0;
};

// Loopback communicator: the peers are (world - 1) imaginary ranks whose
// buckets are all zeros, and the "wire" is a spin of delay_us on the exchange
// stream.  average = bucket / world, late: an exchange that is NOT the
// identity and does not finish at once, on one GPU (tests of the ordering
// between the collectives and the kernels around them).
__global__ void loopback_spin_kernel(long long ticks)
{
    // wall_clock64: the 100 MHz constant counter; time passes, so every wave gets out
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
__global__ __launch_bounds__(256) void loopback_scale_kernel(float *__restrict__ p, size_t n, float s)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] *= s;
}

Rccl &rccl()
{
    static Rccl r;
    if (r.handle || r.ok) return r;
    // An RCCL that is ALREADY in the process first (RTLD_NOLOAD matches loaded objects by
    // soname): a host that links its own copy -- PyTorch ships one -- then shares that one
    // library with these entry points instead of mapping a second 570 MB librccl whose
    // exported symbols collide with the first.  Only then the usual search.
    const char *loaded[] = {"librccl.so.1", "librccl.so"};
    for (const char *n : loaded) {
        r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (r.handle) break;
    }
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char *n : names) {
        if (r.handle) break;
        r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    }
    if (!r.handle) return r;
    r.get_unique_id = (decltype(r.get_unique_id))dlsym(r.handle, "ncclGetUniqueId");
    r.comm_init_rank = (decltype(r.comm_init_rank))dlsym(r.handle, "ncclCommInitRank");
    r.comm_destroy = (decltype(r.comm_destroy))dlsym(r.handle, "ncclCommDestroy");
    r.all_reduce = (decltype(r.all_reduce))dlsym(r.handle, "ncclAllReduce");
    r.comm_count = (decltype(r.comm_count))dlsym(r.handle, "ncclCommCount");
    r.ok = r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.all_reduce && r.comm_count;
    return r;
}

}  // namespace

extern "C" {

int dvsof_comm_unique_id(void *host_id128)
{
    if (!host_id128) return DVSOF_EINVAL;
    Rccl &r = rccl();
    if (!r.ok) return DVSOF_ECOMM;
    ncclUniqueId id;
    if (r.get_unique_id(&id) != ncclSuccess) return DVSOF_ECOMM;
    static_assert(sizeof(id) == DVSOF_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(host_id128, &id, sizeof(id));
    return DVSOF_OK;
}

int dvsof_comm_create(void **comm, int world_size, int rank, const void *host_id128)
{
    if (!comm || !host_id128 || world_size < 1 || rank < 0 || rank >= world_size) return DVSOF_EINVAL;
    Rccl &r = rccl();
    if (!r.ok) return DVSOF_ECOMM;
    ncclUniqueId id;
    memcpy(&id, host_id128, sizeof(id));
    Comm *cm = new Comm;
    cm->world = world_size;
    if (r.comm_init_rank(&cm->c, world_size, id, rank) != ncclSuccess) {
        delete cm;
        return DVSOF_ECOMM;
    }
    *comm = (void *)cm;
    return DVSOF_OK;
}

int dvsof_comm_create_loopback(void **comm, int world_size, int delay_us)
{
    if (!comm || world_size < 1 || delay_us < 0 || delay_us > 100000) return DVSOF_EINVAL;
    Comm *cm = new Comm;
    cm->world = world_size;
    cm->loopback = true;
    cm->delay_us = delay_us;
    *comm = (void *)cm;
    return DVSOF_OK;
}

int dvsof_comm_info(void *comm, int *ranks, int *loopback, unsigned long long *calls,
                    unsigned long long *elements)
{
    if (!comm) return DVSOF_EINVAL;
    Comm *cm = (Comm *)comm;
    if (ranks) {
        *ranks = cm->world;
        if (!cm->loopback) {    // what RCCL itself says the communicator spans
            Rccl &r = rccl();
            int n = 0;
            if (!r.ok || r.comm_count(cm->c, &n) != ncclSuccess) return DVSOF_ECOMM;
            *ranks = n;
        }
    }
    if (loopback) *loopback = cm->loopback ? 1 : 0;
    if (calls) *calls = cm->calls;
    if (elements) *elements = cm->elements;
    return DVSOF_OK;
}

int dvsof_comm_destroy(void *comm)
{
    if (!comm) return DVSOF_EINVAL;
    if (((Comm *)comm)->loopback) {
        delete (Comm *)comm;
        return DVSOF_OK;
    }
    Rccl &r = rccl();
    if (!r.ok) return DVSOF_ECOMM;
    Comm *cm = (Comm *)comm;
    const bool ok = r.comm_destroy(cm->c) == ncclSuccess;
    delete cm;
    return ok ? DVSOF_OK : DVSOF_ECOMM;
}

int dvsof_allreduce_bucket(void *comm, float *bucket, size_t n, void *stream)
{
    if (!comm || !bucket) return DVSOF_EINVAL;
    if (n == 0) return DVSOF_OK;
    Comm *cm = (Comm *)comm;
    ++cm->calls;
    cm->elements += n;
    if (cm->loopback) {
        hipStream_t st = as_stream(stream);
        if (cm->delay_us > 0) {
            hipLaunchKernelGGL(loopback_spin_kernel, dim3(1), dim3(1), 0, st, (long long)cm->delay_us * 100);
            DVSOF_LAUNCH_CHECK();
        }
        size_t blocks = (n + 1023) / 1024;
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(loopback_scale_kernel, dim3((unsigned)blocks), dim3(256), 0, st, bucket, n,
                           1.0f / (float)cm->world);
        DVSOF_LAUNCH_CHECK();
        return DVSOF_OK;
    }
    Rccl &r = rccl();
    if (!r.ok) return DVSOF_ECOMM;
    // in place, average over the ranks: the mean of per-rank gradients is the
    // global-batch gradient (equal per-rank batch).  RCCL
    // implements ncclAvg as a pre-multiplied sum; in a group of ONE that is a
    // scaled-copy kernel per bucket (oneRankReduce<FuncPreMulSum>, 10-50 us
    // each) for a factor of 1.0 -- the sum over one rank is the same average
    // and RCCL returns from it without launching anything.
    return r.all_reduce(bucket, bucket, n, ncclFloat32, cm->world == 1 ? ncclSum : ncclAvg, cm->c,
                        as_stream(stream)) == ncclSuccess
               ? DVSOF_OK
               : DVSOF_ECOMM;
}

}  // extern "C"
