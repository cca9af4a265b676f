// RCCL behind the C ABI: the one collective of the path -- an all-gather of the uint8 results -- on the engine's
// own stream, bound straight to librccl.so (dlopen: no link-time dependency, no torch in the data path).  The 128-byte unique
// id is produced by rank 0 and handed to the other ranks by the host side (diffpir_amd/dist.py ships it over a TCP socket on
// MASTER_ADDR); that rendezvous is plumbing, the collective itself is ncclAllGather over xGMI.
#include "engine.h"
#include <dlfcn.h>
#include <string.h>
#include <mutex>

int dpir_check_range(dpir_engine* e);     // api.hip: the f16 operand range guard (sticky DPIR_ERR_RANGE)

namespace {
typedef struct { char internal[128]; } UniqueId;        // ncclUniqueId (rccl.h:43, NCCL_UNIQUE_ID_BYTES 128)
typedef void* Comm;
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*GetVersion)(int*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
    std::mutex mu;
    bool load() {
        std::lock_guard<std::mutex> lk(mu);       // engines are driven from several host threads
        if (lib) return true;
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) { err = std::string("cannot load librccl.so: ") + dlerror(); return false; }
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
        AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(lib, "ncclAllGather"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(lib, "ncclAllReduce"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        GetVersion = reinterpret_cast<decltype(GetVersion)>(dlsym(lib, "ncclGetVersion"));
        if (!GetUniqueId || !CommInitRank || !AllGather || !AllReduce || !CommDestroy) { err = "librccl.so lacks an expected symbol"; lib = nullptr; return false; }
        return true;
    }
    std::string what(int rc) { return GetErrorString ? std::string(GetErrorString(rc)) : std::to_string(rc); }
};
Rccl g_rccl;
constexpr int kNcclUint8 = 1;      // ncclUint8 (rccl.h ncclDataType_t: ncclInt8 = 0, ncclUint8 = 1)
constexpr int kNcclFloat64 = 8;    // ncclFloat64 / ncclDouble
constexpr int kNcclMax = 2;        // ncclRedOp_t: ncclSum 0, ncclProd 1, ncclMax 2, ncclMin 3

int fail(dpir_engine* e, int code, const std::string& msg) {
    if (e) e->last_error = msg;
    return code;
}
}  // namespace

namespace dpir {
Status comm_allreduce_sum_f64(dpir_engine* e, double* dev, size_t n) {
    if (!e->comm) return Status{DPIR_ERR_STATE, "comm_allreduce_sum_f64: no communicator"};
    int rc = g_rccl.AllReduce(dev, dev, n, kNcclFloat64, /*ncclSum*/ 0, e->comm, e->stream);
    if (rc != 0) return Status{DPIR_ERR_HIP, "ncclAllReduce(sum, f64): " + g_rccl.what(rc)};
    return Status{};
}
}  // namespace dpir

extern "C" {

int dpir_comm_unique_id(void* id128_out) {
    if (!id128_out) return DPIR_ERR_INVALID;
    if (!g_rccl.load()) return DPIR_ERR_UNSUPPORTED;
    UniqueId id;
    if (g_rccl.GetUniqueId(&id) != 0) return DPIR_ERR_HIP;
    memcpy(id128_out, id.internal, 128);
    return DPIR_OK;
}

// ncclGetVersion of the librccl.so this process bound (e.g. 22203), for the self-diagnosing bench line; DPIR_ERR_UNSUPPORTED without librccl
int dpir_comm_version(int* version_out) {
    if (!version_out) return DPIR_ERR_INVALID;
    *version_out = 0;
    if (!g_rccl.load() || !g_rccl.GetVersion) return DPIR_ERR_UNSUPPORTED;
    return g_rccl.GetVersion(version_out) == 0 ? DPIR_OK : DPIR_ERR_HIP;
}

int dpir_comm_init(dpir_engine* e, int world, int rank, const void* id128) {
    if (!e || !id128 || world < 1 || rank < 0 || rank >= world) return fail(e, DPIR_ERR_INVALID, "dpir_comm_init: bad argument");
    if (!g_rccl.load()) return fail(e, DPIR_ERR_UNSUPPORTED, g_rccl.err);
    if (e->comm) return fail(e, DPIR_ERR_STATE, "dpir_comm_init: communicator already initialised");
    (void)hipSetDevice(e->device);
    UniqueId id;
    memcpy(id.internal, id128, 128);
    Comm c = nullptr;
    int rc = g_rccl.CommInitRank(&c, world, id, rank);
    if (rc != 0) return fail(e, DPIR_ERR_HIP, "ncclCommInitRank: " + g_rccl.what(rc));
    e->comm = c; e->comm_world = world; e->comm_rank = rank;
    return DPIR_OK;
}

int dpir_allgather_results(dpir_engine* e, const void* send_dev, void* recv_dev, size_t bytes_per_rank) {
    if (!e || !send_dev || !recv_dev) return fail(e, DPIR_ERR_INVALID, "dpir_allgather_results: null argument");
    if (!e->comm) return fail(e, DPIR_ERR_STATE, "dpir_allgather_results: dpir_comm_init has not been called");
    if (int rr = dpir_check_range(e)) return rr;       // never ship clamped (wrong) images to the other ranks
    int rc = g_rccl.AllGather(send_dev, recv_dev, bytes_per_rank, kNcclUint8, e->comm, e->stream);      // stream-ordered behind the loop
    if (rc != 0) return fail(e, DPIR_ERR_HIP, "ncclAllGather: " + g_rccl.what(rc));
    return DPIR_OK;
}

// MAX all-reduce of one host double over the communicator (the bench's elapsed time; a barrier when the value is ignored):
// staged through an engine-owned 8-byte device word on the engine stream, synchronous.
int dpir_comm_allreduce_max(dpir_engine* e, double* value_inout) {
    if (!e || !value_inout) return fail(e, DPIR_ERR_INVALID, "dpir_comm_allreduce_max: null argument");
    if (!e->comm) return fail(e, DPIR_ERR_STATE, "dpir_comm_allreduce_max: dpir_comm_init has not been called");
    (void)hipSetDevice(e->device);
    double* w = nullptr;
    dpir::Status st = e->ws.getT("comm#word", (size_t)2, &w);
    if (!st.ok()) return fail(e, st.code, st.msg);
    if (hipMemcpyAsync(w, value_inout, sizeof(double), hipMemcpyHostToDevice, e->stream) != hipSuccess) return fail(e, DPIR_ERR_HIP, "comm word upload failed");
    int rc = g_rccl.AllReduce(w, w + 1, 1, kNcclFloat64, kNcclMax, e->comm, e->stream);
    if (rc != 0) return fail(e, DPIR_ERR_HIP, "ncclAllReduce: " + g_rccl.what(rc));
    if (hipMemcpyAsync(value_inout, w + 1, sizeof(double), hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
        hipStreamSynchronize(e->stream) != hipSuccess) return fail(e, DPIR_ERR_HIP, "comm word download failed");
    return DPIR_OK;
}

int dpir_comm_barrier(dpir_engine* e) {
    double v = 0.0;
    return dpir_comm_allreduce_max(e, &v);
}

int dpir_comm_destroy(dpir_engine* e) {
    if (!e) return DPIR_ERR_INVALID;
    if (e->comm) {
        (void)hipStreamSynchronize(e->stream);
        (void)g_rccl.CommDestroy(e->comm);
        e->comm = nullptr;
    }
    return DPIR_OK;
}

}  // extern "C"
