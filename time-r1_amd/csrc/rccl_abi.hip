// RCCL behind the C ABI (SURVEY 8b last row: rccl_{init, allreduce, reduce_scatter, allgather}): the collectives of the data-parallel path for a host that
// binds libtimer1_hip.so directly instead of going through torch.distributed (whose "nccl" backend IS this library on ROCm).
//
// Reference call sites they replace: the DDP / DeepSpeed gradient exchange under `torchrun --nproc_per_node=8`
// (scripts/finetune/run_activitynet.sh:11, scripts/zero3.json:22-33 reduce_scatter / allgather buckets) and accelerate's gather_for_metrics
// (src/time_r1/rl/timer1_trainer.py:741-777).  One process per GPU; the communicator rides the caller's HIP stream, so a collective is ordered
// with the kernels enqueued before it exactly like a kernel launch (time-r1_amd/dist.py issues them per arena segment during the backward).
//
// librccl.so is opened at run time (dlopen), not linked: a process that already carries RCCL (torch ships its own copy under the same soname)
// gets THAT copy back, and a single-GPU process that never calls tr1_rccl_* loads nothing.
#include "tr1_common.h"

#include <dlfcn.h>
#include <stdio.h>

namespace {
typedef int ncclResult;                          // ncclSuccess == 0
typedef struct { char internal[128]; } UniqueId; // NCCL_UNIQUE_ID_BYTES
typedef void* Comm;
struct Rccl {
    void* h = nullptr;
    ncclResult (*GetUniqueId)(UniqueId*) = nullptr;
    ncclResult (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    ncclResult (*CommDestroy)(Comm) = nullptr;
    ncclResult (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    ncclResult (*ReduceScatter)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    ncclResult (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult) = nullptr;
    ncclResult (*GetVersion)(int*) = nullptr;
};
Rccl g_rccl;

bool rccl_load() {
    if (g_rccl.h) return true;
    const char* names[] = {getenv("TR1_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) {
        if (!n || !*n) continue;
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) { tr1_set_error_("rccl: librccl.so not found (set TR1_RCCL_LIB)"); return false; }
#define SYM(field, name) do { *(void**)(&g_rccl.field) = dlsym(h, name); if (!g_rccl.field) { tr1_set_error_("rccl: symbol " name " missing"); return false; } } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy"); SYM(AllReduce, "ncclAllReduce");
    SYM(ReduceScatter, "ncclReduceScatter"); SYM(AllGather, "ncclAllGather"); SYM(GetErrorString, "ncclGetErrorString"); SYM(GetVersion, "ncclGetVersion");
#undef SYM
    g_rccl.h = h;
    return true;
}
int rccl_fail(const char* what, ncclResult r) {
    char buf[256];
    snprintf(buf, sizeof(buf), "rccl: %s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    tr1_set_error_(buf);
    return 2000 + (int)r;
}
// dtype codes of this ABI -> ncclDataType_t (rccl.h: ncclFloat32 = 7, ncclBfloat16 = 9, ncclInt32 = 2); reduction is always a sum (ncclSum = 0)
bool rccl_dtype(int dtype, int* out) {
    switch (dtype) { case 0: *out = 9; return true; case 1: *out = 7; return true; case 2: *out = 2; return true; default: return false; }
}
}  // namespace

extern "C" int64_t tr1_rccl_version(void) {
    if (!rccl_load()) return -1;
    int v = 0;
    return g_rccl.GetVersion(&v) == 0 ? (int64_t)v : -1;
}

// rank 0 calls this and hands the 128 bytes to every rank (file, environment, the launcher's store); HOST pointer
extern "C" int tr1_rccl_unique_id(void* id_out_128) {
    TR1_CHECK_ARG(id_out_128, "rccl_unique_id: null output");
    if (!rccl_load()) return 1000;
    UniqueId id;
    const ncclResult r = g_rccl.GetUniqueId(&id);
    if (r) return rccl_fail("ncclGetUniqueId", r);
    memcpy(id_out_128, &id, sizeof(id));
    return 0;
}

// Every rank of the job calls this once, with its device already selected (hipSetDevice); *comm_out (HOST slot for one pointer) receives the communicator
extern "C" int tr1_rccl_init(const void* id_128, int64_t world, int64_t rank, void* comm_out) {
    TR1_CHECK_ARG(id_128 && comm_out && world >= 1 && rank >= 0 && rank < world, "rccl_init: bad arguments");
    if (!rccl_load()) return 1000;
    UniqueId id;
    memcpy(&id, id_128, sizeof(id));
    Comm c = nullptr;
    const ncclResult r = g_rccl.CommInitRank(&c, (int)world, id, (int)rank);
    if (r) return rccl_fail("ncclCommInitRank", r);
    *(void**)comm_out = c;
    return 0;
}

extern "C" int tr1_rccl_destroy(void* comm) {
    if (!comm) return 0;
    if (!rccl_load()) return 1000;
    const ncclResult r = g_rccl.CommDestroy((Comm)comm);
    return r ? rccl_fail("ncclCommDestroy", r) : 0;
}

// recv[i] = sum over ranks of send[i], i < count (in place when send == recv).  dtype: 0 bf16 (the gradient wire format), 1 fp32, 2 int32
extern "C" int tr1_rccl_allreduce(void* comm, const void* send, void* recv, int64_t count, int dtype, void* stream) {
    int dt;
    TR1_CHECK_ARG(comm && send && recv && count >= 0 && rccl_dtype(dtype, &dt), "rccl_allreduce: bad arguments");
    if (count == 0) return 0;
    const ncclResult r = g_rccl.AllReduce(send, recv, (size_t)count, dt, 0, (Comm)comm, (hipStream_t)stream);
    return r ? rccl_fail("ncclAllReduce", r) : 0;
}

// recv[i] = sum over ranks of send[rank * recv_count + i], i < recv_count (send holds world * recv_count elements): the sharded optimizer's gradient exchange
extern "C" int tr1_rccl_reduce_scatter(void* comm, const void* send, void* recv, int64_t recv_count, int dtype, void* stream) {
    int dt;
    TR1_CHECK_ARG(comm && send && recv && recv_count >= 0 && rccl_dtype(dtype, &dt), "rccl_reduce_scatter: bad arguments");
    if (recv_count == 0) return 0;
    const ncclResult r = g_rccl.ReduceScatter(send, recv, (size_t)recv_count, dt, 0, (Comm)comm, (hipStream_t)stream);
    return r ? rccl_fail("ncclReduceScatter", r) : 0;
}

// recv[r * send_count + i] = rank r's send[i]: the updated bf16 weight chunks of the sharded optimizer, the per-step metric vectors
extern "C" int tr1_rccl_allgather(void* comm, const void* send, void* recv, int64_t send_count, int dtype, void* stream) {
    int dt;
    TR1_CHECK_ARG(comm && send && recv && send_count >= 0 && rccl_dtype(dtype, &dt), "rccl_allgather: bad arguments");
    if (send_count == 0) return 0;
    const ncclResult r = g_rccl.AllGather(send, recv, (size_t)send_count, dt, (Comm)comm, (hipStream_t)stream);
    return r ? rccl_fail("ncclAllGather", r) : 0;
}
