// pg_gather.cpp — the ONE exchange of the multi-GPU path behind the C ABI: every rank's packed
// posteriors (lik f64, lik_exp i32: pg_job_packed_results) to the root over RCCL / xGMI.
//
// Chains (contig x path subset x sample) are independent — the reference runs them as independent
// thread-pool jobs and only merges results (src/commands.cpp:955-978, 163-177) — so ranks shard the
// chains with NO data-path collective; this gather is the only communication.  It is a grouped set of
// point-to-point operations (ncclSend / ncclRecv inside one ncclGroupStart/End): each peer's block is
// exactly as long as its data (no widening of the int32 exponents, no padding to the longest rank) and
// travels over that peer's own direct xGMI link to the root — nothing ring-shaped.
//
// RCCL is loaded at run time (dlopen) the first time a communicator is made: single-GPU users of
// libpangenie_hmm.so never load it.  Two ways to get communicators:
//   pg_comm_init      one process per GPU (rank 0 makes the id with pg_comm_unique_id and hands it round
//                     by whatever the host already has: a file, MPI, torch.distributed ...)
//   pg_comm_init_all  one process, several GPUs (ncclCommInitAll)
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "../../include/pangenie_hmm.h"

namespace {

void set_err(char* err, size_t errlen, const char* fmt, ...) {
    if (!err || errlen == 0) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err, errlen, fmt, ap);
    va_end(ap);
}

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

bool load_rccl(char* err, size_t errlen) {
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl.lib) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) { set_err(err, errlen, "cannot load librccl.so: %s", dlerror()); return false; }
#define SYM(field, name)                                                                    \
    do {                                                                                    \
        *(void**)(&g_rccl.field) = dlsym(h, name);                                          \
        if (!g_rccl.field) { set_err(err, errlen, "librccl.so lacks %s", name); dlclose(h); return false; } \
    } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommInitAll, "ncclCommInitAll");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl.lib = h;
    return true;
}

#define NCCL_TRY(call)                                                                               \
    do {                                                                                             \
        ncclResult_t r_ = (call);                                                                    \
        if (r_ != ncclSuccess) {                                                                     \
            set_err(err, errlen, "%s failed: %s", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?"); \
            return PG_ERR_DEVICE;                                                                    \
        }                                                                                            \
    } while (0)
#define HIP_TRY(call)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            set_err(err, errlen, "%s failed: %s", #call, hipGetErrorString(e_));          \
            return PG_ERR_DEVICE;                                                         \
        }                                                                                 \
    } while (0)

}  // namespace

struct pg_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

extern "C" int pg_comm_unique_id(uint8_t id128[128], char* err, size_t errlen) {
    if (!id128) { set_err(err, errlen, "null argument"); return PG_ERR_INVALID; }
    if (!load_rccl(err, errlen)) return PG_ERR_DEVICE;
    ncclUniqueId id;
    NCCL_TRY(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, 128);
    return PG_OK;
}

extern "C" int pg_comm_init(const uint8_t id128[128], int world, int rank, int device, pg_comm** out, char* err, size_t errlen) {
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) { set_err(err, errlen, "bad argument"); return PG_ERR_INVALID; }
    *out = nullptr;
    if (!load_rccl(err, errlen)) return PG_ERR_DEVICE;
    HIP_TRY(hipSetDevice(device));
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    pg_comm* c = new pg_comm();
    c->rank = rank; c->world = world; c->device = device;
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        set_err(err, errlen, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
        delete c;
        return PG_ERR_DEVICE;
    }
    *out = c;
    return PG_OK;
}

extern "C" int pg_comm_init_all(int n_devices, const int* devices, pg_comm** out_comms, char* err, size_t errlen) {
    if (n_devices < 1 || !devices || !out_comms) { set_err(err, errlen, "bad argument"); return PG_ERR_INVALID; }
    if (!load_rccl(err, errlen)) return PG_ERR_DEVICE;
    std::vector<ncclComm_t> comms(n_devices);
    NCCL_TRY(g_rccl.CommInitAll(comms.data(), n_devices, devices));
    for (int i = 0; i < n_devices; ++i) {
        pg_comm* c = new pg_comm();
        c->comm = comms[i]; c->rank = i; c->world = n_devices; c->device = devices[i];
        out_comms[i] = c;
    }
    return PG_OK;
}

extern "C" void pg_comm_destroy(pg_comm* c) {
    if (!c) return;
    if (c->comm && g_rccl.CommDestroy) { hipSetDevice(c->device); g_rccl.CommDestroy(c->comm); }
    delete c;
}

extern "C" int pg_comm_rank(const pg_comm* c) { return c ? c->rank : -1; }
extern "C" int pg_comm_world(const pg_comm* c) { return c ? c->world : 0; }

// The exchange for the `n_local` communicators this process holds (1 in the process-per-GPU layout, all
// of them after pg_comm_init_all).  Blocking: returns when the root's buffers hold every block.
extern "C" int pg_hmm_gather_all(int n_local, pg_comm* const* comms, pg_job* const* jobs, int root,
                                 const uint64_t* n_lik_per_rank, void* d_lik_all, void* d_exp_all,
                                 char* err, size_t errlen) {
    if (n_local < 1 || !comms || !jobs || !n_lik_per_rank) { set_err(err, errlen, "bad argument"); return PG_ERR_INVALID; }
    for (int i = 0; i < n_local; ++i)
        if (!comms[i]) { set_err(err, errlen, "communicator %d is null", i); return PG_ERR_INVALID; }
    const int world = comms[0]->world;
    if (root < 0 || root >= world) { set_err(err, errlen, "bad root"); return PG_ERR_INVALID; }
    // NOTE: the checks up to the group are LOCAL: a rank that fails one returns before the exchange while its peers
    // enter it and wait for this rank's block.  The plan (n_lik_per_rank) is known to every host up front, so a
    // mismatch is a programming error of the caller; a caller that cannot rule it out agrees on the return codes of a
    // dry call (comm = the same, jobs checked with pg_job_packed_results) before the exchange, or aborts the peers.
    std::vector<uint64_t> off((size_t)world + 1, 0);
    for (int r = 0; r < world; ++r) off[r + 1] = off[r] + n_lik_per_rank[r];
    // loop-back self test (tests on a single GPU): the root's own block also goes through ncclSend / ncclRecv
    const bool loopback = getenv("PG_GATHER_LOOPBACK") != nullptr;
    struct Local { void* d_lik; void* d_exp; uint64_t n; };
    std::vector<Local> loc(n_local);
    for (int i = 0; i < n_local; ++i) {
        const pg_comm* c = comms[i];
        if (!c || c->world != world) { set_err(err, errlen, "communicators of different worlds"); return PG_ERR_INVALID; }
        loc[i] = {nullptr, nullptr, 0};
        if (jobs[i]) pg_job_packed_results(jobs[i], &loc[i].d_lik, &loc[i].d_exp, &loc[i].n);
        if (loc[i].n != n_lik_per_rank[c->rank]) {
            set_err(err, errlen, "rank %d holds %llu genotype bins, the plan says %llu", c->rank, (unsigned long long)loc[i].n,
                    (unsigned long long)n_lik_per_rank[c->rank]);
            return PG_ERR_INVALID;
        }
        if (c->rank == root && (!d_lik_all || !d_exp_all) && off[world] > 0) { set_err(err, errlen, "root needs receive buffers"); return PG_ERR_INVALID; }
    }
    if (world > 1 || loopback) {
        if (!load_rccl(err, errlen)) return PG_ERR_DEVICE;
        NCCL_TRY(g_rccl.GroupStart());
        // inside the group nothing returns: a failure is remembered, the group is always closed (an open group would
        // swallow every later RCCL call of this thread), and the first error is reported afterwards
        int rc_in = PG_OK;
        auto note_nccl = [&](ncclResult_t r, const char* what) {
            if (r != ncclSuccess && rc_in == PG_OK) { set_err(err, errlen, "%s failed: %s", what, g_rccl.GetErrorString(r)); rc_in = PG_ERR_DEVICE; }
        };
        for (int i = 0; i < n_local && rc_in == PG_OK; ++i) {
            const pg_comm* c = comms[i];
            const hipError_t he = hipSetDevice(c->device);
            if (he != hipSuccess) { set_err(err, errlen, "hipSetDevice(%d) failed: %s", c->device, hipGetErrorString(he)); rc_in = PG_ERR_DEVICE; break; }
            if (c->rank == root) {
                for (int q = 0; q < world && rc_in == PG_OK; ++q) {
                    if ((q == root && !loopback) || n_lik_per_rank[q] == 0) continue;
                    note_nccl(g_rccl.Recv((double*)d_lik_all + off[q], n_lik_per_rank[q], ncclDouble, q, c->comm, nullptr), "ncclRecv(lik)");
                    note_nccl(g_rccl.Recv((int32_t*)d_exp_all + off[q], n_lik_per_rank[q], ncclInt32, q, c->comm, nullptr), "ncclRecv(lik_exp)");
                }
            }
            if ((c->rank != root || loopback) && loc[i].n > 0 && rc_in == PG_OK) {
                note_nccl(g_rccl.Send(loc[i].d_lik, loc[i].n, ncclDouble, root, c->comm, nullptr), "ncclSend(lik)");
                note_nccl(g_rccl.Send(loc[i].d_exp, loc[i].n, ncclInt32, root, c->comm, nullptr), "ncclSend(lik_exp)");
            }
        }
        const ncclResult_t r_end = g_rccl.GroupEnd();
        if (rc_in != PG_OK) return rc_in;
        note_nccl(r_end, "ncclGroupEnd");
        if (rc_in != PG_OK) return rc_in;
    }
    for (int i = 0; i < n_local; ++i) {
        const pg_comm* c = comms[i];
        HIP_TRY(hipSetDevice(c->device));
        if (c->rank == root && !loopback && loc[i].n > 0) {  // the root's own block: device to device
            HIP_TRY(hipMemcpyAsync((double*)d_lik_all + off[root], loc[i].d_lik, loc[i].n * sizeof(double), hipMemcpyDeviceToDevice, nullptr));
            HIP_TRY(hipMemcpyAsync((int32_t*)d_exp_all + off[root], loc[i].d_exp, loc[i].n * sizeof(int32_t), hipMemcpyDeviceToDevice, nullptr));
        }
        HIP_TRY(hipStreamSynchronize(nullptr));
    }
    return PG_OK;
}

extern "C" int pg_hmm_gather(pg_comm* comm, pg_job* job, int root, const uint64_t* n_lik_per_rank, void* d_lik_all,
                             void* d_exp_all, char* err, size_t errlen) {
    if (!comm) { set_err(err, errlen, "null communicator"); return PG_ERR_INVALID; }
    pg_comm* cs[1] = {comm};
    pg_job* js[1] = {job};
    return pg_hmm_gather_all(1, cs, js, root, n_lik_per_rank, d_lik_all, d_exp_all, err, errlen);
}

// Convenience for hosts without their own device buffers: the gather above into temporary buffers on the
// root's device, then one D2H copy of each into h_lik_all / h_exp_all (root's process only).
extern "C" int pg_hmm_gather_to_host(int n_local, pg_comm* const* comms, pg_job* const* jobs, int root,
                                     const uint64_t* n_lik_per_rank, double* h_lik_all, int32_t* h_exp_all,
                                     char* err, size_t errlen) {
    if (n_local < 1 || !comms || !n_lik_per_rank) { set_err(err, errlen, "bad argument"); return PG_ERR_INVALID; }
    for (int i = 0; i < n_local; ++i)
        if (!comms[i]) { set_err(err, errlen, "communicator %d is null", i); return PG_ERR_INVALID; }
    const int world = comms[0]->world;
    uint64_t total = 0;
    for (int r = 0; r < world; ++r) total += n_lik_per_rank[r];
    const pg_comm* rootc = nullptr;
    for (int i = 0; i < n_local; ++i)
        if (comms[i]->rank == root) rootc = comms[i];
    double* d_l = nullptr; int32_t* d_e = nullptr;
    if (rootc && total) {
        if (!h_lik_all || !h_exp_all) { set_err(err, errlen, "root needs host buffers"); return PG_ERR_INVALID; }
        HIP_TRY(hipSetDevice(rootc->device));
        if (hipMalloc((void**)&d_l, total * sizeof(double)) != hipSuccess || hipMalloc((void**)&d_e, total * sizeof(int32_t)) != hipSuccess) {
            if (d_l) hipFree(d_l);
            set_err(err, errlen, "hipMalloc of the gather buffers failed");
            return PG_ERR_NOMEM;
        }
    }
    int rc = pg_hmm_gather_all(n_local, comms, jobs, root, n_lik_per_rank, d_l, d_e, err, errlen);
    if (rc == PG_OK && rootc && total) {
        hipSetDevice(rootc->device);
        if (hipMemcpy(h_lik_all, d_l, total * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(h_exp_all, d_e, total * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) {
            set_err(err, errlen, "D2H of the gathered posteriors failed");
            rc = PG_ERR_DEVICE;
        }
    }
    if (d_l) hipFree(d_l);
    if (d_e) hipFree(d_e);
    return rc;
}
