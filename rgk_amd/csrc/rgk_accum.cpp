// The frame accumulator on the device (EXRTexture total_ob of RenderFrame, reference src/render_driver.cpp:199,
// src/texture.hpp:83-118) and its raw checkpoint.  Host code; the kernels add into it (k_resolve, splats).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rgk.h"

extern "C" int rgk_internal_fail(int code, const char* msg);

struct rgk_accum {
    uint32_t xres = 0, yres = 0;
    int device = 0;
    float* rgb = nullptr;
    uint32_t* count = nullptr;
    uint64_t tag = 0; // what the frame was rendered from (rgk_accum_set_tag): saved with a checkpoint, checked when one is loaded
    size_t pixels() const { return (size_t)xres * yres; }
};

namespace {
int hip_fail(const char* what, hipError_t e) {
    char buf[256];
    snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    return rgk_internal_fail(e == hipErrorOutOfMemory ? RGK_ERR_OOM : RGK_ERR_DEVICE, buf);
}
struct CkptHeader { // little-endian, 32 bytes
    char magic[8];  // "RGKACC1\0"
    uint32_t xres, yres, rounds_done, seedcount;
    uint32_t tag_lo, tag_hi; // rgk_accum_set_tag of the saving run (0: untagged; these words were reserved and zero before)
};
// dst += src, element-wise: what the root of a reduce does with the round's sum (EXRTexture::Accumulate, src/texture.cpp:403-412)
__global__ void k_accum_add(float* __restrict__ drgb, uint32_t* __restrict__ dcnt, const float* __restrict__ srgb, const uint32_t* __restrict__ scnt, size_t P) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < 3 * P; i += (size_t)gridDim.x * blockDim.x) {
        drgb[i] = drgb[i] + srgb[i];
        if (i < P) dcnt[i] += scnt[i];
    }
}
} // namespace

extern "C" {

int rgk_accum_create(uint32_t xres, uint32_t yres, int device, rgk_accum** out) {
    if (!out) return rgk_internal_fail(RGK_ERR_INVALID, "null output pointer");
    *out = nullptr;
    if (xres == 0 || yres == 0 || xres > 65535 || yres > 65535) return rgk_internal_fail(RGK_ERR_INVALID, "resolution out of range");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return rgk_internal_fail(RGK_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return rgk_internal_fail(RGK_ERR_INVALID, "device out of range");
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail("hipSetDevice", e);
    rgk_accum* a = new rgk_accum;
    a->xres = xres; a->yres = yres; a->device = device;
    if ((e = hipMalloc((void**)&a->rgb, a->pixels() * 3 * sizeof(float))) != hipSuccess) { delete a; return hip_fail("hipMalloc", e); }
    if ((e = hipMalloc((void**)&a->count, a->pixels() * sizeof(uint32_t))) != hipSuccess) { (void)hipFree(a->rgb); delete a; return hip_fail("hipMalloc", e); }
    const int rc = rgk_accum_clear(a);
    if (rc) { rgk_accum_destroy(a); return rc; }
    *out = a;
    return RGK_OK;
}

void rgk_accum_destroy(rgk_accum* a) {
    if (!a) return;
    (void)hipSetDevice(a->device);
    if (a->rgb) (void)hipFree(a->rgb);
    if (a->count) (void)hipFree(a->count);
    delete a;
}

int rgk_accum_clear(rgk_accum* a) {
    if (!a) return rgk_internal_fail(RGK_ERR_INVALID, "null argument");
    hipError_t e = hipSetDevice(a->device);
    if (e == hipSuccess) e = hipMemset(a->rgb, 0, a->pixels() * 3 * sizeof(float));
    if (e == hipSuccess) e = hipMemset(a->count, 0, a->pixels() * sizeof(uint32_t));
    if (e == hipSuccess) e = hipDeviceSynchronize();
    return e == hipSuccess ? RGK_OK : hip_fail("clear accumulator", e);
}

int rgk_accum_add(rgk_accum* dst, const rgk_accum* src) {
    if (!dst || !src) return rgk_internal_fail(RGK_ERR_INVALID, "null argument");
    if (dst->xres != src->xres || dst->yres != src->yres || dst->device != src->device) return rgk_internal_fail(RGK_ERR_INVALID, "accumulators differ in size or device");
    hipError_t e = hipSetDevice(dst->device);
    if (e != hipSuccess) return hip_fail("hipSetDevice", e);
    const size_t P = dst->pixels();
    k_accum_add<<<(unsigned)std::min<size_t>((3 * P + 255) / 256, 8192), 256>>>(dst->rgb, dst->count, src->rgb, src->count, P);
    if ((e = hipGetLastError()) == hipSuccess) e = hipDeviceSynchronize();
    return e == hipSuccess ? RGK_OK : hip_fail("add accumulators", e);
}

int rgk_accum_set_tag(rgk_accum* a, uint64_t tag) {
    if (!a) return rgk_internal_fail(RGK_ERR_INVALID, "null argument");
    a->tag = tag;
    return RGK_OK;
}

float* rgk_accum_rgb(rgk_accum* a) { return a ? a->rgb : nullptr; }
uint32_t* rgk_accum_count(rgk_accum* a) { return a ? a->count : nullptr; }

int rgk_accum_download(const rgk_accum* a, float* rgb, uint32_t* count) {
    if (!a) return rgk_internal_fail(RGK_ERR_INVALID, "null argument");
    hipError_t e = hipSetDevice(a->device);
    if (e == hipSuccess && rgb) e = hipMemcpy(rgb, a->rgb, a->pixels() * 3 * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess && count) e = hipMemcpy(count, a->count, a->pixels() * sizeof(uint32_t), hipMemcpyDeviceToHost);
    return e == hipSuccess ? RGK_OK : hip_fail("download accumulator", e);
}

int rgk_accum_upload(rgk_accum* a, const float* rgb, const uint32_t* count) {
    if (!a) return rgk_internal_fail(RGK_ERR_INVALID, "null argument");
    hipError_t e = hipSetDevice(a->device);
    if (e == hipSuccess && rgb) e = hipMemcpy(a->rgb, rgb, a->pixels() * 3 * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && count) e = hipMemcpy(a->count, count, a->pixels() * sizeof(uint32_t), hipMemcpyHostToDevice);
    return e == hipSuccess ? RGK_OK : hip_fail("upload accumulator", e);
}

int rgk_accum_save(const rgk_accum* a, const char* path, uint32_t rounds_done, uint32_t seedcount) {
    if (!a || !path) return rgk_internal_fail(RGK_ERR_INVALID, "null argument");
    std::vector<float> rgb(a->pixels() * 3);
    std::vector<uint32_t> cnt(a->pixels());
    int rc = rgk_accum_download(a, rgb.data(), cnt.data());
    if (rc) return rc;
    // written beside the target and renamed over it: a run killed mid-write leaves the previous checkpoint intact
    const std::string tmp = std::string(path) + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return rgk_internal_fail(RGK_ERR_INVALID, "cannot open the checkpoint file for writing");
    CkptHeader h{};
    std::memcpy(h.magic, "RGKACC1", 8);
    h.xres = a->xres; h.yres = a->yres; h.rounds_done = rounds_done; h.seedcount = seedcount;
    h.tag_lo = (uint32_t)a->tag; h.tag_hi = (uint32_t)(a->tag >> 32);
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && fwrite(rgb.data(), sizeof(float), rgb.size(), f) == rgb.size() &&
              fwrite(cnt.data(), sizeof(uint32_t), cnt.size(), f) == cnt.size();
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(tmp.c_str(), path) != 0) { remove(tmp.c_str()); return rgk_internal_fail(RGK_ERR_INVALID, "short write to the checkpoint file"); }
    return RGK_OK;
}

int rgk_accum_load(rgk_accum* a, const char* path, uint32_t* rounds_done, uint32_t* seedcount) {
    if (!a || !path) return rgk_internal_fail(RGK_ERR_INVALID, "null argument");
    FILE* f = fopen(path, "rb");
    if (!f) return rgk_internal_fail(RGK_ERR_INVALID, "cannot open the checkpoint file");
    CkptHeader h{};
    std::vector<float> rgb(a->pixels() * 3);
    std::vector<uint32_t> cnt(a->pixels());
    bool ok = fread(&h, sizeof(h), 1, f) == 1 && std::memcmp(h.magic, "RGKACC1", 8) == 0;
    if (ok && (h.xres != a->xres || h.yres != a->yres)) { fclose(f); return rgk_internal_fail(RGK_ERR_INVALID, "checkpoint resolution differs from the accumulator's"); }
    if (ok) { // a checkpoint of another scene / camera / parameter set must not be continued (both sides tagged: compared)
        const uint64_t ftag = (uint64_t)h.tag_lo | ((uint64_t)h.tag_hi << 32);
        if (a->tag && ftag && a->tag != ftag) { fclose(f); return rgk_internal_fail(RGK_ERR_INVALID, "checkpoint was written for a different scene, camera or parameter set"); }
    }
    ok = ok && fread(rgb.data(), sizeof(float), rgb.size(), f) == rgb.size() && fread(cnt.data(), sizeof(uint32_t), cnt.size(), f) == cnt.size();
    fclose(f);
    if (!ok) return rgk_internal_fail(RGK_ERR_INVALID, "not a checkpoint file, or truncated");
    if (rounds_done) *rounds_done = h.rounds_done;
    if (seedcount) *seedcount = h.seedcount;
    return rgk_accum_upload(a, rgb.data(), cnt.data());
}

} // extern "C"
