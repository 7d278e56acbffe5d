// The one exchange step of a multi-GPU frame, behind the C ABI: EXRTexture::Accumulate under total_ob_mx
// (reference src/render_driver.cpp:177-182, src/texture.cpp:403-412) becomes one RCCL sum-reduce of the per-GPU
// accumulators to the root rank after each round.  One process per GPU; tiles are dealt round-robin by the host
// (rgk_shard_tiles) and carry their seeds, so no other data crosses the GPUs.
//
// RCCL is bound at run time (dlopen), not at link time: a single-GPU host needs no librccl at all, and a process that
// already holds a copy (PyTorch-ROCm bundles its own librccl.so.1) gets THAT copy instead of a second one.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/rgk.h"

extern "C" int rgk_internal_fail(int code, const char* msg);

namespace {

// the slice of rccl.h this file uses (RCCL keeps NCCL's ABI: /opt/rocm/include/rccl/rccl.h)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclUint32 = 3, ncclFloat32 = 7 }; // ncclDataType_t
enum { ncclSum = 0 };                     // ncclRedOp_t

struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*Reduce)(const void*, void*, size_t, int, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
};
Rccl g_rccl;
std::once_flag g_once;

void load_rccl() {
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names) // a copy the process already holds first
        if ((g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!g_rccl.lib)
        for (const char* n : names)
            if ((g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!g_rccl.lib) { g_rccl.err = std::string("cannot load librccl: ") + dlerror(); return; }
    auto sym = [](const char* s) { return dlsym(g_rccl.lib, s); };
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
    g_rccl.Reduce = reinterpret_cast<decltype(g_rccl.Reduce)>(sym("ncclReduce"));
    g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(sym("ncclGroupStart"));
    g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(sym("ncclGroupEnd"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.Reduce || !g_rccl.GroupStart || !g_rccl.GroupEnd ||
        !g_rccl.GetErrorString)
        g_rccl.err = "librccl lacks an expected symbol";
}
int need_rccl() {
    std::call_once(g_once, load_rccl);
    if (!g_rccl.err.empty()) return rgk_internal_fail(RGK_ERR_UNSUPPORTED, g_rccl.err.c_str());
    return RGK_OK;
}
int nccl_fail(const char* what, int rc) {
    char buf[256];
    snprintf(buf, sizeof(buf), "%s: %s", what, g_rccl.GetErrorString(rc));
    return rgk_internal_fail(RGK_ERR_DEVICE, buf);
}

} // namespace

struct rgk_comm {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int rank = 0, world = 1, device = 0;
};

extern "C" {

int rgk_comm_get_unique_id(uint8_t id[RGK_COMM_ID_BYTES]) {
    if (!id) return rgk_internal_fail(RGK_ERR_INVALID, "null argument");
    int rc = need_rccl();
    if (rc) return rc;
    static_assert(RGK_COMM_ID_BYTES == sizeof(ncclUniqueId), "id size");
    ncclUniqueId u;
    const int n = g_rccl.GetUniqueId(&u);
    if (n != ncclSuccess) return nccl_fail("ncclGetUniqueId", n);
    std::memcpy(id, u.internal, sizeof(u.internal));
    return RGK_OK;
}

int rgk_comm_create(const uint8_t id[RGK_COMM_ID_BYTES], int rank, int world_size, int device, rgk_comm** out) {
    if (!id || !out) return rgk_internal_fail(RGK_ERR_INVALID, "null argument");
    *out = nullptr;
    if (world_size < 1 || rank < 0 || rank >= world_size) return rgk_internal_fail(RGK_ERR_INVALID, "rank outside [0, world_size)");
    int rc = need_rccl();
    if (rc) return rc;
    if (hipSetDevice(device) != hipSuccess) return rgk_internal_fail(RGK_ERR_DEVICE, "hipSetDevice failed");
    rgk_comm* c = new rgk_comm;
    c->rank = rank; c->world = world_size; c->device = device;
    ncclUniqueId u;
    std::memcpy(u.internal, id, sizeof(u.internal));
    const int n = g_rccl.CommInitRank(&c->comm, world_size, u, rank);
    if (n != ncclSuccess) { delete c; return nccl_fail("ncclCommInitRank", n); }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        g_rccl.CommDestroy(c->comm);
        delete c;
        return rgk_internal_fail(RGK_ERR_DEVICE, "hipStreamCreate failed");
    }
    *out = c;
    return RGK_OK;
}

void rgk_comm_destroy(rgk_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (c->comm) g_rccl.CommDestroy(c->comm);
    delete c;
}

int rgk_accum_reduce(rgk_comm* c, float* d_accum_rgb, uint32_t* d_accum_count, uint32_t xres, uint32_t yres, int root) {
    if (!c || !d_accum_rgb) return rgk_internal_fail(RGK_ERR_INVALID, "null argument");
    if (root < 0 || root >= c->world) return rgk_internal_fail(RGK_ERR_INVALID, "root outside [0, world_size)");
    if (hipSetDevice(c->device) != hipSuccess) return rgk_internal_fail(RGK_ERR_DEVICE, "hipSetDevice failed");
    const size_t P = (size_t)xres * yres;
    // in place: the root's buffers end up holding the sum, the others keep their own contribution.  Both reduces form one
    // group (one launch); the caller's render round has synchronised its stream before returning.
    int n = g_rccl.GroupStart();
    if (n == ncclSuccess) n = g_rccl.Reduce(d_accum_rgb, d_accum_rgb, 3 * P, ncclFloat32, ncclSum, root, c->comm, c->stream);
    if (n == ncclSuccess && d_accum_count) n = g_rccl.Reduce(d_accum_count, d_accum_count, P, ncclUint32, ncclSum, root, c->comm, c->stream);
    const int e = g_rccl.GroupEnd();
    if (n != ncclSuccess) return nccl_fail("ncclReduce", n);
    if (e != ncclSuccess) return nccl_fail("ncclGroupEnd", e);
    if (hipStreamSynchronize(c->stream) != hipSuccess) return rgk_internal_fail(RGK_ERR_DEVICE, "reduce stream failed");
    return RGK_OK;
}

// Round-robin deal of the centre-out tile list (tile i -> rank i mod world_size): the tiles keep their seeds, so the image
// does not depend on the number of GPUs.  Call with out == NULL for the count.
int rgk_shard_tiles(const rgk_tile* tiles, uint32_t n_tiles, int rank, int world_size, rgk_tile* out, uint32_t* n_out) {
    if (!n_out || (!tiles && n_tiles) || world_size < 1 || rank < 0 || rank >= world_size) return rgk_internal_fail(RGK_ERR_INVALID, "bad argument");
    uint32_t n = 0;
    for (uint32_t i = (uint32_t)rank; i < n_tiles; i += (uint32_t)world_size) {
        if (out) {
            if (n >= *n_out) return rgk_internal_fail(RGK_ERR_INVALID, "tile buffer too small");
            out[n] = tiles[i];
        }
        n++;
    }
    *n_out = n;
    return RGK_OK;
}

} // extern "C"
