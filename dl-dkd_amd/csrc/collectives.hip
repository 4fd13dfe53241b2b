// Collectives of the two exchanges BASELINE.json's north_star adds to the path: the all-gather of the per-rank score blocks of
// the video-sharded gallery (method/eval.py:188-212 cut by video) and the all-reduce of the flat gradient buffer of the
// data-parallel step (method/train.py:147-151 under DDP).  RCCL, driven directly: the library resolves librccl.so.1 at the first
// communicator call (the copy already in the process when the host is PyTorch-ROCm), a communicator is an explicit object the
// CALLER creates and destroys, and every collective is ONE enqueue on the caller's stream - no helper thread polls events, so a
// collective can sit between (or inside) hipGraph captures of the same process without anyone observing a captured event.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and enums only: every function is resolved with dlsym below

#include <string.h>

#include <mutex>

#include "../../include/dldkd_hip.h"
#include "common.hpp"

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl g_rccl;
std::once_flag g_once;

template <class F>
bool sym(void* h, const char* name, F& out) {
    out = reinterpret_cast<F>(dlsym(h, name));
    return out != nullptr;
}

void load_rccl() {
    // RTLD_NOLOAD first: PyTorch-ROCm has its own librccl.so.1 mapped, and two RCCL copies in one process would each bring their
    // own bootstrap state.  A host without one (tests/c/abi_client) gets the system library.
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { dldkd::set_error("collectives: librccl.so.1 not found (%s)", dlerror()); return; }
    Rccl& r = g_rccl;
    r.handle = h;
    bool all = true;
    all = sym(h, "ncclGetVersion", r.GetVersion) && all;
    all = sym(h, "ncclGetUniqueId", r.GetUniqueId) && all;
    all = sym(h, "ncclCommInitRank", r.CommInitRank) && all;
    all = sym(h, "ncclCommDestroy", r.CommDestroy) && all;
    all = sym(h, "ncclCommAbort", r.CommAbort) && all;
    all = sym(h, "ncclCommGetAsyncError", r.CommGetAsyncError) && all;
    all = sym(h, "ncclCommCount", r.CommCount) && all;
    all = sym(h, "ncclCommUserRank", r.CommUserRank) && all;
    all = sym(h, "ncclAllReduce", r.AllReduce) && all;
    all = sym(h, "ncclAllGather", r.AllGather) && all;
    all = sym(h, "ncclBroadcast", r.Broadcast) && all;
    all = sym(h, "ncclGroupStart", r.GroupStart) && all;
    all = sym(h, "ncclGroupEnd", r.GroupEnd) && all;
    all = sym(h, "ncclGetErrorString", r.GetErrorString) && all;
    if (!all) { dldkd::set_error("collectives: librccl.so.1 lacks a required symbol"); return; }
    r.ok = true;
}

const Rccl* rccl() {
    std::call_once(g_once, load_rccl);
    return g_rccl.ok ? &g_rccl : nullptr;
}

int fail(const Rccl* r, const char* what, ncclResult_t e) {
    dldkd::set_error("%s: RCCL error %d (%s)", what, (int)e, r && r->GetErrorString ? r->GetErrorString(e) : "?");
    return DLDKD_ECOMM;
}

bool dtype_of(int dtype, ncclDataType_t& t) {
    switch (dtype) {
        case DLDKD_F32: t = ncclFloat32; return true;
        case DLDKD_F64: t = ncclFloat64; return true;
        case DLDKD_I32: t = ncclInt32; return true;
        case DLDKD_I64: t = ncclInt64; return true;
        case DLDKD_U8: t = ncclUint8; return true;
    }
    return false;
}

bool op_of(int op, ncclRedOp_t& o) {
    switch (op) {
        case DLDKD_SUM: o = ncclSum; return true;
        case DLDKD_MAX: o = ncclMax; return true;
        case DLDKD_MIN: o = ncclMin; return true;
    }
    return false;
}

}   // namespace

extern "C" {

int dldkd_comm_rccl_version(void) {
    const Rccl* r = rccl();
    if (!r) return DLDKD_ECOMM;
    int v = 0;
    ncclResult_t e = r->GetVersion(&v);
    return e == ncclSuccess ? v : fail(r, "dldkd_comm_rccl_version", e);
}

int dldkd_comm_unique_id(void* host_id_out) {
    static_assert(sizeof(ncclUniqueId) == DLDKD_COMM_ID_BYTES, "DLDKD_COMM_ID_BYTES must be RCCL's unique-id size");
    if (!host_id_out) { dldkd::set_error("dldkd_comm_unique_id: null output"); return DLDKD_EINVAL; }
    const Rccl* r = rccl();
    if (!r) return DLDKD_ECOMM;
    ncclResult_t e = r->GetUniqueId(reinterpret_cast<ncclUniqueId*>(host_id_out));
    return e == ncclSuccess ? DLDKD_OK : fail(r, "dldkd_comm_unique_id", e);
}

int dldkd_comm_init(void** host_comm_out, int world, int rank, const void* host_id) {
    if (!host_comm_out || !host_id || world < 1 || rank < 0 || rank >= world) {
        dldkd::set_error("dldkd_comm_init: need an output slot, the %d-byte id and 0 <= rank < world (got rank %d of %d)",
                         DLDKD_COMM_ID_BYTES, rank, world);
        return DLDKD_EINVAL;
    }
    const Rccl* r = rccl();
    if (!r) return DLDKD_ECOMM;
    ncclUniqueId id;
    ::memcpy(&id, host_id, sizeof(id));
    ncclComm_t c = nullptr;
    ncclResult_t e = r->CommInitRank(&c, world, id, rank);
    if (e != ncclSuccess) return fail(r, "dldkd_comm_init", e);
    *host_comm_out = c;
    return DLDKD_OK;
}

int dldkd_comm_info(void* comm, int* host_world_out, int* host_rank_out) {
    const Rccl* r = rccl();
    if (!r) return DLDKD_ECOMM;
    if (!comm) { dldkd::set_error("dldkd_comm_info: null communicator"); return DLDKD_EINVAL; }
    ncclResult_t e = ncclSuccess;
    if (host_world_out && (e = r->CommCount((ncclComm_t)comm, host_world_out)) != ncclSuccess) return fail(r, "dldkd_comm_info", e);
    if (host_rank_out && (e = r->CommUserRank((ncclComm_t)comm, host_rank_out)) != ncclSuccess) return fail(r, "dldkd_comm_info", e);
    return DLDKD_OK;
}

int dldkd_comm_destroy(void* comm) {
    const Rccl* r = rccl();
    if (!r) return DLDKD_ECOMM;
    if (!comm) return DLDKD_OK;
    ncclResult_t e = r->CommDestroy((ncclComm_t)comm);
    return e == ncclSuccess ? DLDKD_OK : fail(r, "dldkd_comm_destroy", e);
}

int dldkd_comm_abort(void* comm) {
    const Rccl* r = rccl();
    if (!r) return DLDKD_ECOMM;
    if (!comm) return DLDKD_OK;
    ncclResult_t e = r->CommAbort((ncclComm_t)comm);
    return e == ncclSuccess ? DLDKD_OK : fail(r, "dldkd_comm_abort", e);
}

int dldkd_comm_async_error(void* comm) {
    const Rccl* r = rccl();
    if (!r) return DLDKD_ECOMM;
    if (!comm) { dldkd::set_error("dldkd_comm_async_error: null communicator"); return DLDKD_EINVAL; }
    ncclResult_t st = ncclSuccess;
    ncclResult_t e = r->CommGetAsyncError((ncclComm_t)comm, &st);
    if (e != ncclSuccess) return fail(r, "dldkd_comm_async_error", e);
    if (st != ncclSuccess && st != ncclInProgress) return fail(r, "asynchronous communicator error", st);
    return DLDKD_OK;
}

int dldkd_comm_all_reduce(void* comm, const void* send, void* recv, size_t count, int dtype, int op, void* stream) {
    ncclDataType_t t;
    ncclRedOp_t o;
    if (!comm || !dtype_of(dtype, t) || !op_of(op, o) || (count && (!send || !recv))) {
        dldkd::set_error("dldkd_comm_all_reduce: null communicator / buffer or unknown dtype %d / op %d", dtype, op);
        return DLDKD_EINVAL;
    }
    if (count == 0) return DLDKD_OK;
    const Rccl* r = rccl();
    if (!r) return DLDKD_ECOMM;
    ncclResult_t e = r->AllReduce(send, recv, count, t, o, (ncclComm_t)comm, (hipStream_t)stream);
    return e == ncclSuccess ? DLDKD_OK : fail(r, "dldkd_comm_all_reduce", e);
}

int dldkd_comm_all_gather(void* comm, const void* send, void* recv, size_t send_count, int dtype, void* stream) {
    ncclDataType_t t;
    if (!comm || !dtype_of(dtype, t) || (send_count && (!send || !recv))) {
        dldkd::set_error("dldkd_comm_all_gather: null communicator / buffer or unknown dtype %d", dtype);
        return DLDKD_EINVAL;
    }
    if (send_count == 0) return DLDKD_OK;
    const Rccl* r = rccl();
    if (!r) return DLDKD_ECOMM;
    ncclResult_t e = r->AllGather(send, recv, send_count, t, (ncclComm_t)comm, (hipStream_t)stream);
    return e == ncclSuccess ? DLDKD_OK : fail(r, "dldkd_comm_all_gather", e);
}

int dldkd_comm_broadcast(void* comm, void* buf, size_t count, int dtype, int root, void* stream) {
    ncclDataType_t t;
    if (!comm || !dtype_of(dtype, t) || root < 0 || (count && !buf)) {
        dldkd::set_error("dldkd_comm_broadcast: null communicator / buffer, unknown dtype %d or root %d", dtype, root);
        return DLDKD_EINVAL;
    }
    if (count == 0) return DLDKD_OK;
    const Rccl* r = rccl();
    if (!r) return DLDKD_ECOMM;
    ncclResult_t e = r->Broadcast(buf, buf, count, t, root, (ncclComm_t)comm, (hipStream_t)stream);
    return e == ncclSuccess ? DLDKD_OK : fail(r, "dldkd_comm_broadcast", e);
}

int dldkd_comm_group_begin(void) {
    const Rccl* r = rccl();
    if (!r) return DLDKD_ECOMM;
    ncclResult_t e = r->GroupStart();
    return e == ncclSuccess ? DLDKD_OK : fail(r, "dldkd_comm_group_begin", e);
}

int dldkd_comm_group_end(void) {
    const Rccl* r = rccl();
    if (!r) return DLDKD_ECOMM;
    ncclResult_t e = r->GroupEnd();
    return e == ncclSuccess ? DLDKD_OK : fail(r, "dldkd_comm_group_end", e);
}

}   // extern "C"
