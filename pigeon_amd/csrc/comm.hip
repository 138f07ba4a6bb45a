// comm.hip -- the one collective of the path behind the C ABI: an all-gather over RCCL (xGMI inside a node).
//
// Replaces `accelerator.gather` (reference preprocessing/embed.py:36-37): every rank contributes `bytes_per_rank` bytes, every
// rank receives the rank-major concatenation.  One process per GPU; the 128-byte unique id is created on rank 0
// (pg_comm_unique_id) and handed to the other ranks by the host program's own bootstrap channel (the Python side uses the
// torch.distributed store it already has from torchrun's MASTER_ADDR / MASTER_PORT).
//
// RCCL is bound at run time (dlopen of librccl.so.1, the SONAME both /opt/rocm and the PyTorch wheel ship): a process that
// already carries an RCCL -- e.g. through `import torch` -- keeps using that single copy; PIGEON_RCCL_LIB overrides the
// library path.  The communicator is tied to the HIP device current at pg_comm_init_rank.
#include "pigeon_internal.h"

#include <dlfcn.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <rccl/rccl.h>

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
};

Rccl* rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return r.lib ? &r : nullptr;
    tried = true;
    const char* env = getenv("PIGEON_RCCL_LIB");
    void* lib = nullptr;
    if (env && *env) lib = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);      // a copy already in the process (e.g. torch's)
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) { pg_set_error("comm: cannot load librccl.so.1: %s", dlerror()); return nullptr; }
#define SYM(field, name) \
    *(void**)(&r.field) = dlsym(lib, name); \
    if (!r.field) { pg_set_error("comm: %s not found in RCCL", name); return nullptr; }
    SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
    SYM(CommCount, "ncclCommCount") SYM(AllGather, "ncclAllGather") SYM(GetErrorString, "ncclGetErrorString")
    SYM(GetVersion, "ncclGetVersion") SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd")
#undef SYM
    r.lib = lib;
    return &r;
}

#define PG_NCCL(call)                                                                              \
    do {                                                                                           \
        ncclResult_t e__ = (call);                                                                 \
        if (e__ != ncclSuccess) {                                                                  \
            pg_set_error("%s failed: %s", #call, R->GetErrorString(e__));                          \
            return PG_EHIP;                                                                        \
        }                                                                                          \
    } while (0)

struct pg_comm {
    ncclComm_t comm = nullptr;
    int nranks = 0, rank = 0, device = 0;
};

}  // namespace

extern "C" int pg_comm_unique_id(void* id_out) {
    if (!id_out) { pg_set_error("comm_unique_id: null argument"); return PG_EINVAL; }
    Rccl* R = rccl();
    if (!R) return PG_EHIP;
    static_assert(sizeof(ncclUniqueId) == PG_COMM_ID_BYTES, "unique id size");
    ncclUniqueId id;
    PG_NCCL(R->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return PG_OK;
}

extern "C" int pg_comm_init_rank(void** comm, int nranks, const void* unique_id, int rank) {
    if (!comm || !unique_id || nranks < 1 || rank < 0 || rank >= nranks) {
        pg_set_error("comm_init_rank: bad argument (nranks=%d rank=%d)", nranks, rank);
        return PG_EINVAL;
    }
    Rccl* R = rccl();
    if (!R) return PG_EHIP;
    pg_comm* c = new pg_comm();
    c->nranks = nranks; c->rank = rank;
    if (hipGetDevice(&c->device) != hipSuccess) { delete c; pg_set_error("comm_init_rank: no current HIP device"); return PG_EHIP; }
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    // RCCL 2.26 prints a two-line version banner on STDOUT (through C stdio) when its first communicator comes up; a host program
    // whose stdout is a protocol (bench.py: ONE JSON line) must not carry it: stdout is pointed at stderr for the duration of the
    // call, flushed on both sides.
    fflush(stdout);
    const int saved = dup(STDOUT_FILENO);
    if (saved >= 0) (void)dup2(STDERR_FILENO, STDOUT_FILENO);
    ncclResult_t e = R->CommInitRank(&c->comm, nranks, id, rank);
    fflush(stdout);
    if (saved >= 0) { (void)dup2(saved, STDOUT_FILENO); (void)close(saved); }
    if (e != ncclSuccess) { pg_set_error("ncclCommInitRank failed: %s", R->GetErrorString(e)); delete c; return PG_EHIP; }
    *comm = c;
    return PG_OK;
}

extern "C" int pg_comm_count(void* comm, int* nranks) {
    if (!comm || !nranks) { pg_set_error("comm_count: null argument"); return PG_EINVAL; }
    Rccl* R = rccl();
    if (!R) return PG_EHIP;
    PG_NCCL(R->CommCount(((pg_comm*)comm)->comm, nranks));
    return PG_OK;
}

extern "C" int pg_allgather(void* comm, const void* send, void* recv, size_t bytes_per_rank, void* stream) {
    if (!comm || !send || !recv) { pg_set_error("allgather: null argument"); return PG_EINVAL; }
    if (bytes_per_rank == 0) return PG_OK;
    Rccl* R = rccl();
    if (!R) return PG_EHIP;
    pg_comm* c = (pg_comm*)comm;
    PG_NCCL(R->AllGather(send, recv, bytes_per_rank, ncclInt8, c->comm, (hipStream_t)stream));
    return PG_OK;
}

// Several buffers with the same rank layout in ONE RCCL group (one fused launch): the step's embeddings, candidate cells,
// candidate probabilities, initial predictions and sample indices each arrive rank-major and contiguous, without packing.
extern "C" int pg_allgather_many(void* comm, int count, const void* const* send, void* const* recv, const size_t* bytes_per_rank,
                                 void* stream) {
    if (!comm || count < 0 || (count > 0 && (!send || !recv || !bytes_per_rank))) { pg_set_error("allgather_many: bad argument"); return PG_EINVAL; }
    if (count == 0) return PG_OK;
    Rccl* R = rccl();
    if (!R) return PG_EHIP;
    pg_comm* c = (pg_comm*)comm;
    for (int i = 0; i < count; ++i)
        if (bytes_per_rank[i] && (!send[i] || !recv[i])) { pg_set_error("allgather_many: null buffer %d", i); return PG_EINVAL; }
    PG_NCCL(R->GroupStart());
    ncclResult_t first = ncclSuccess;
    for (int i = 0; i < count; ++i) {
        if (!bytes_per_rank[i]) continue;
        const ncclResult_t e = R->AllGather(send[i], recv[i], bytes_per_rank[i], ncclInt8, c->comm, (hipStream_t)stream);
        if (e != ncclSuccess && first == ncclSuccess) first = e;
    }
    const ncclResult_t e2 = R->GroupEnd();
    if (first != ncclSuccess) { pg_set_error("ncclAllGather (grouped) failed: %s", R->GetErrorString(first)); return PG_EHIP; }
    if (e2 != ncclSuccess) { pg_set_error("ncclGroupEnd failed: %s", R->GetErrorString(e2)); return PG_EHIP; }
    return PG_OK;
}

extern "C" int pg_comm_destroy(void* comm) {
    if (!comm) return PG_OK;
    Rccl* R = rccl();
    pg_comm* c = (pg_comm*)comm;
    if (R && c->comm) (void)R->CommDestroy(c->comm);
    delete c;
    return PG_OK;
}

extern "C" int pg_comm_rccl_version(void) {
    Rccl* R = rccl();
    if (!R) return PG_EHIP;
    int v = 0;
    if (R->GetVersion(&v) != ncclSuccess) return PG_EHIP;
    return v;
}
