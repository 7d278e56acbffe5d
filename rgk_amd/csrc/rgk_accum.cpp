// The frame accumulator on the device (EXRTexture total_ob of RenderFrame, reference src/render_driver.cpp:199,
// src/texture.hpp:83-118) and its raw checkpoint.  Host code; the kernels add into it (k_resolve, splats).
#include <hip/hip_runtime.h>

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
    uint32_t reserved[2];
};
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
    ok = ok && fread(rgb.data(), sizeof(float), rgb.size(), f) == rgb.size() && fread(cnt.data(), sizeof(uint32_t), cnt.size(), f) == cnt.size();
    fclose(f);
    if (!ok) return rgk_internal_fail(RGK_ERR_INVALID, "not a checkpoint file, or truncated");
    if (rounds_done) *rounds_done = h.rounds_done;
    if (seedcount) *seedcount = h.seedcount;
    return rgk_accum_upload(a, rgb.data(), cnt.data());
}

} // extern "C"
