// ORACLE -- test infrastructure only.
// Driver around the reference's vendored external/stb_image.h (compiled from where it lies under /root/reference; no reference
// source is copied into this repo): decodes a Radiance .hdr file exactly as FileTexture::CreateNewFromHDR does (reference
// src/texture.cpp:294-321: stbi_loadf(path, &w, &h, &n, 0)) and writes `w h n` as three int32 followed by w*h*n float32.
//   stbi_ref <in.hdr> <out.bin>
#include <cstdio>
#include <cstdlib>
#define STB_IMAGE_IMPLEMENTATION
#define STBI_ONLY_HDR // as the reference builds it, src/stbi.cpp:1-3
#include "stb_image.h"

int main(int argc, char** argv) {
    if (argc != 3) return 2;
    int w, h, n;
    float* d = stbi_loadf(argv[1], &w, &h, &n, 0);
    if (!d) { fprintf(stderr, "stbi_loadf failed: %s\n", stbi_failure_reason()); return 1; }
    FILE* f = fopen(argv[2], "wb");
    if (!f) return 1;
    int hdr[3] = {w, h, n};
    fwrite(hdr, sizeof(int), 3, f);
    fwrite(d, sizeof(float), (size_t)w * h * n, f);
    fclose(f);
    free(d);
    return 0;
}
