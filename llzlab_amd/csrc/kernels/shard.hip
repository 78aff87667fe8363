// shard.hip -- the device half of the single-process sharded handles (csrc/host/llz_shard_host.c, include/llz_shard.h):
// device selection, per-shard streams and events, and the one collective of the path -- the broadcast of the coefficient
// tables from the first shard's device to the others at init ("one ncclBroadcast(root=0) per coefficient
// table at init ... single process, ncclCommInitAll over the selected devices"; steady state has no inter-GPU traffic).
// RCCL is loaded on first use (dlopen), so a caller that stays on one GPU never needs librccl.
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>
#include "common.hpp"

extern "C" int llzs_device_get(void)
{
    int d = -1;
    LLZ_HIP_CHECK(hipGetDevice(&d));
    return d;
}

extern "C" int llzs_device_set(int device)
{
    LLZ_HIP_CHECK(hipSetDevice(device));
    return LLZ_OK;
}

extern "C" int llzs_device_enter(int device)
{
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) return -1;
    if (prev == device) return -1;                      // nothing to restore
    if (hipSetDevice(device) != hipSuccess) return -1;
    return prev;
}

extern "C" void llzs_device_leave(int previous)
{
    if (previous >= 0) (void)hipSetDevice(previous);
}

extern "C" void *llzs_stream_create(void)
{
    hipStream_t s = nullptr;
    const hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) {
        llzs_set_error("hipStreamCreateWithFlags: %s", hipGetErrorString(e));
        return nullptr;
    }
    return s;
}

extern "C" void llzs_stream_destroy(void *stream)
{
    if (stream) (void)hipStreamDestroy(as_stream(stream));
}

extern "C" void *llzs_event_create(void)
{
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) {
        llzs_set_error("hipEventCreate failed");
        return nullptr;
    }
    return e;
}

extern "C" void llzs_event_destroy(void *event)
{
    if (event) (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(event));
}

extern "C" int llzs_event_record(void *event, void *stream)
{
    LLZ_HIP_CHECK(hipEventRecord(reinterpret_cast<hipEvent_t>(event), as_stream(stream)));
    return LLZ_OK;
}

extern "C" double llzs_event_elapsed_ms(void *start, void *stop)
{
    float ms = -1.f;
    if (hipEventSynchronize(reinterpret_cast<hipEvent_t>(stop)) != hipSuccess ||
        hipEventElapsedTime(&ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)) != hipSuccess) {
        llzs_set_error("event timing failed");
        return -1.0;
    }
    return (double)ms;
}

// ---- table capture ---------------------------------------------------------------------------------------------------
static thread_local int g_table_mode = 0;
static thread_local llzs_table_ref g_tables[LLZS_MAX_TABLES];
static thread_local int g_table_count = 0;

extern "C" void llzs_table_capture(int mode)
{
    g_table_mode = mode;
    g_table_count = 0;
}

extern "C" int llzs_table_captured(llzs_table_ref *dst, int capacity)
{
    const int n = g_table_count < capacity ? g_table_count : capacity;
    for (int i = 0; i < n; i++) dst[i] = g_tables[i];
    return g_table_count;
}

extern "C" int llzs_h2d_table(void *dev_dst, const void *host_src, size_t bytes)
{
    if (g_table_mode != 0) {
        if (g_table_count >= LLZS_MAX_TABLES) {
            llzs_set_error("more than %d coefficient tables in one handle", LLZS_MAX_TABLES);
            return LLZ_ERR_RANGE;
        }
        g_tables[g_table_count].dev = dev_dst;
        g_tables[g_table_count].bytes = bytes;
        g_table_count++;
        if (g_table_mode == 2) return LLZ_OK;           // filled by llzs_tables_broadcast
    }
    return llzs_h2d(dev_dst, host_src, bytes, nullptr);
}

// ---- RCCL, loaded on first use -----------------------------------------------------------------------------------------
namespace {

struct rccl_api {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

rccl_api &rccl()
{
    static rccl_api api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
        }
        if (!api.lib) return;
        api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(dlsym(api.lib, "ncclCommInitAll"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.lib, "ncclCommDestroy"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(dlsym(api.lib, "ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(dlsym(api.lib, "ncclGroupEnd"));
        api.Broadcast = reinterpret_cast<decltype(api.Broadcast)>(dlsym(api.lib, "ncclBroadcast"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.lib, "ncclGetErrorString"));
        api.ok = api.CommInitAll && api.CommDestroy && api.GroupStart && api.GroupEnd && api.Broadcast && api.GetErrorString;
    });
    return api;
}

#define LLZ_NCCL_CHECK(expr)                                                                    \
    do {                                                                                        \
        ncclResult_t r__ = (expr);                                                              \
        if (r__ != ncclSuccess) {                                                               \
            llzs_set_error("%s failed: %s", #expr, R.GetErrorString(r__));                      \
            rc = LLZ_ERR_DEVICE;                                                                \
            goto done;                                                                          \
        }                                                                                       \
    } while (0)

} // namespace

static thread_local int g_last_comm_ranks = 0;
extern "C" int llzs_tables_broadcast_ranks(void) { return g_last_comm_ranks; }

extern "C" int llzs_tables_broadcast(llzs_table_ref *const *tables, int ntables, int nshards, const int *device,
                                     void *const *stream)
{
    g_last_comm_ranks = 0;
    if (!tables || !device || !stream || nshards < 1 || ntables < 0 || nshards > 64) {
        llzs_set_error("tables_broadcast: bad arguments");
        return LLZ_ERR_ARG;
    }
    for (int s = 1; s < nshards; s++)
        for (int t = 0; t < ntables; t++)
            if (tables[s][t].bytes != tables[0][t].bytes) {
                llzs_set_error("tables_broadcast: shard %d table %d has %zu bytes, shard 0 %zu", s, t, tables[s][t].bytes,
                               tables[0][t].bytes);
                return LLZ_ERR_ARG;
            }
    // distinct devices in order of first appearance (shard 0's device = communicator rank 0 = the root); lead[u] = the
    // first shard on device udev[u]
    int udev[64], lead[64], nu = 0;
    for (int s = 0; s < nshards; s++) {
        int u = 0;
        while (u < nu && udev[u] != device[s]) u++;
        if (u == nu) { udev[nu] = device[s]; lead[nu] = s; nu++; }
    }
    int prev = -1, rc = LLZ_OK;
    (void)hipGetDevice(&prev);
    const bool use_rccl = nu > 1 || llzs_tune(LLZS_TUNE_SHARD_RCCL) == 1;
    ncclComm_t comm[64];
    int ncomm = 0;
    rccl_api &R = rccl();
    if (use_rccl && ntables > 0) {
        if (!R.ok) {
            llzs_set_error("sharding over %d devices needs librccl.so (not found, or symbols missing)", nu);
            return LLZ_ERR_DEVICE;
        }
        // shard 0's uploads ran on the default stream and were synchronous for the host: the data is there
        LLZ_NCCL_CHECK(R.CommInitAll(comm, nu, udev));
        ncomm = nu;
        g_last_comm_ranks = nu;
        for (int t = 0; t < ntables; t++) {
            LLZ_NCCL_CHECK(R.GroupStart());
            for (int u = 0; u < nu; u++) {
                if (hipSetDevice(udev[u]) != hipSuccess) { llzs_set_error("hipSetDevice(%d) failed", udev[u]); rc = LLZ_ERR_DEVICE; }
                llzs_table_ref &dst = tables[lead[u]][t];
                const ncclResult_t r = R.Broadcast(tables[0][t].dev, dst.dev, dst.bytes, ncclChar, 0, comm[u],
                                                   as_stream(stream[lead[u]]));
                if (r != ncclSuccess) { llzs_set_error("ncclBroadcast failed: %s", R.GetErrorString(r)); rc = LLZ_ERR_DEVICE; }
            }
            LLZ_NCCL_CHECK(R.GroupEnd());
            if (rc != LLZ_OK) goto done;
        }
    }
    // further shards on a device copy from that device's lead shard (device to device, same stream order as the broadcast)
    for (int s = 0; s < nshards && rc == LLZ_OK; s++) {
        int u = 0;
        while (udev[u] != device[s]) u++;
        if (lead[u] == s) continue;
        if (hipSetDevice(device[s]) != hipSuccess) { llzs_set_error("hipSetDevice(%d) failed", device[s]); rc = LLZ_ERR_DEVICE; break; }
        // order behind the lead shard's stream: the lead's tables are complete once its stream has drained
        if (hipStreamSynchronize(as_stream(stream[lead[u]])) != hipSuccess) { rc = LLZ_ERR_DEVICE; break; }
        for (int t = 0; t < ntables; t++)
            if (hipMemcpyAsync(tables[s][t].dev, tables[lead[u]][t].dev, tables[s][t].bytes, hipMemcpyDeviceToDevice,
                               as_stream(stream[s])) != hipSuccess) {
                llzs_set_error("table copy to shard %d failed", s);
                rc = LLZ_ERR_DEVICE;
                break;
            }
    }
done:
    for (int s = 0; s < nshards; s++) {
        if (hipSetDevice(device[s]) == hipSuccess) (void)hipStreamSynchronize(as_stream(stream[s]));
    }
    for (int u = 0; u < ncomm; u++) (void)R.CommDestroy(comm[u]);
    if (prev >= 0) (void)hipSetDevice(prev);
    return rc;
}
