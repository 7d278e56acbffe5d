// A compiled C++ host of the C ABI (tests/test_cpp_caller.py builds and runs it): plays the part of the reference's
// RenderFrame -- holds a Scene, a Camera, a Config and an EXRTexture shaped like the reference's (reference_mirror.hpp),
// binds them with the code INTEGRATION.md shows (rgk_binding.inc) and renders rounds.  A second mode drives the
// multi-GPU entry points (device accumulator, tile sharding, RCCL reduce) the way one rank of an 8-GPU host would.
//   caller <scene.bin> <out.bin> [--device-accum | --emulate-world W]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

#include "reference_mirror.hpp"

static mat33 g_ggx_m[4096], g_bek_m[4096]; // the reference compiles its tables in (src/LTC/ltc_ggx.cpp); here they come with the scene file
static float g_ggx_a[4096], g_bek_a[4096];
const LTCdef LTC::GGX = {64, g_ggx_m, g_ggx_a};
const LTCdef LTC::Beckmann = {64, g_bek_m, g_bek_a};

#include "rgk_binding.inc"

struct Reader {
    std::ifstream f;
    explicit Reader(const char* p) : f(p, std::ios::binary) { if (!f) throw std::runtime_error("cannot open scene file"); }
    template <typename T> T get() { T v; f.read(reinterpret_cast<char*>(&v), sizeof(T)); if (!f) throw std::runtime_error("scene file truncated"); return v; }
    template <typename T> std::vector<T> arr(size_t n) { std::vector<T> v(n); if (n) f.read(reinterpret_cast<char*>(v.data()), n * sizeof(T)); if (!f) throw std::runtime_error("scene file truncated"); return v; }
};

static void fill_ltc(Reader& r, mat33* M, float* A) {
    std::vector<float> rec = r.arr<float>(4096 * 5);
    for (int i = 0; i < 4096; i++) {
        M[i] = mat33{};
        M[i].m[0] = rec[5 * i]; M[i].m[2] = rec[5 * i + 1]; M[i].m[4] = rec[5 * i + 2]; M[i].m[6] = rec[5 * i + 3]; M[i].m[8] = 1.0;
        A[i] = rec[5 * i + 4];
    }
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: caller <scene.bin> <out.bin> [--device-accum | --emulate-world W]\n"); return 2; }
    const int emulate_world = (argc > 4 && !strcmp(argv[3], "--emulate-world")) ? atoi(argv[4]) : 1;
    if (emulate_world < 1 || emulate_world > 64) { fprintf(stderr, "--emulate-world: 1..64\n"); return 2; }
    const bool device_accum = (argc > 3 && !strcmp(argv[3], "--device-accum")) || emulate_world > 1;
    try {
        Reader r(argv[1]);
        if (r.get<uint32_t>() != 0x524b4753u) throw std::runtime_error("bad magic");
        // ---- a Scene as the reference's loaders leave it after Commit()
        Scene scene;
        const uint32_t nv = r.get<uint32_t>();
        std::vector<float> V = r.arr<float>(3 * nv), N = r.arr<float>(3 * nv), T = r.arr<float>(3 * nv);
        const uint32_t has_uv = r.get<uint32_t>();
        std::vector<float> UV = r.arr<float>(has_uv ? 2 * nv : 0);
        std::vector<glm::vec3> verts(nv), norms(nv), tangs(nv);
        std::vector<glm::vec2> uvs(has_uv ? nv : 0);
        for (uint32_t i = 0; i < nv; i++) {
            verts[i] = glm::vec3(V[3 * i], V[3 * i + 1], V[3 * i + 2]); norms[i] = glm::vec3(N[3 * i], N[3 * i + 1], N[3 * i + 2]);
            tangs[i] = glm::vec3(T[3 * i], T[3 * i + 1], T[3 * i + 2]);
            if (has_uv) uvs[i] = glm::vec2(UV[2 * i], UV[2 * i + 1]);
        }
        scene.vertices = verts.data(); scene.normals = norms.data(); scene.tangents = tangs.data(); scene.texcoords = has_uv ? uvs.data() : nullptr;
        scene.n_vertices = scene.n_normals = scene.n_tangents = nv; scene.n_texcoords = has_uv ? nv : 0;
        const uint32_t nt = r.get<uint32_t>();
        std::vector<uint32_t> F = r.arr<uint32_t>(3 * nt), FM = r.arr<uint32_t>(nt);
        const uint32_t nm = r.get<uint32_t>();
        struct MatRec { uint32_t kind, flags; float emission[3], roughness, ior, amount; int32_t tex_diffuse, tex_color, tex_bump, mix_m1, mix_m2; };
        std::vector<MatRec> mrec = r.arr<MatRec>(nm);
        const uint32_t ntex = r.get<uint32_t>();
        std::vector<std::shared_ptr<ReadableTexture>> textures;
        for (uint32_t i = 0; i < ntex; i++) {
            const uint32_t kind = r.get<uint32_t>(), w = r.get<uint32_t>(), h = r.get<uint32_t>();
            const float cr = r.get<float>(), cg = r.get<float>(), cb = r.get<float>();
            if (kind == 0) textures.push_back(std::make_shared<SolidTexture>(Color(cr, cg, cb)));
            else {
                auto ft = std::make_shared<FileTexture>((int)w, (int)h);
                std::vector<float> px = r.arr<float>((size_t)3 * w * h);
                for (uint32_t y = 0; y < h; y++) for (uint32_t x = 0; x < w; x++) { const size_t k = 3 * ((size_t)y * w + x); ft->SetPixel((int)x, (int)y, Color(px[k], px[k + 1], px[k + 2])); }
                textures.push_back(ft);
            }
        }
        auto tex = [&](int32_t id) -> std::shared_ptr<ReadableTexture> { return id < 0 ? std::make_shared<EmptyTexture>() : textures[id]; };
        std::vector<std::shared_ptr<Material>> materials(nm);
        for (uint32_t i = 0; i < nm; i++) materials[i] = std::make_shared<Material>();
        for (uint32_t i = 0; i < nm; i++) {
            const MatRec& m = mrec[i];
            Material& o = *materials[i];
            o.name = "m" + std::to_string(i);
            o.emission = Radiance(m.emission[0], m.emission[1], m.emission[2]);
            o.no_russian = (m.flags & 1u) != 0;
            o.bumpmap = tex(m.tex_bump);
            switch (m.kind) {
            case 0: { auto b = new BxDFDiffuse; b->diffuse = tex(m.tex_diffuse); o.bxdf.reset(b); break; }
            case 1: { auto b = new BxDFMirror; b->color = tex(m.tex_color); o.bxdf.reset(b); break; }
            case 2: { auto b = new BxDFDielectric; b->ior = m.ior; b->color = tex(m.tex_color); o.bxdf.reset(b); break; }
            case 3: o.bxdf.reset(new BxDFTransparent); break;
            case 4: { auto b = new BxDFMix; b->m1 = materials[m.mix_m1]; b->m2 = materials[m.mix_m2]; b->amt1 = m.amount; o.bxdf.reset(b); break; }
            case 5: { auto b = new BxDFLTC<LTC::Beckmann>; b->roughness = m.roughness; b->color = tex(m.tex_color); o.bxdf.reset(b); break; }
            case 6: { auto b = new BxDFLTC<LTC::GGX>; b->roughness = m.roughness; b->color = tex(m.tex_color); o.bxdf.reset(b); break; }
            case 7: { auto b = new BxDFLTCDiffuse<LTC::Beckmann>; b->roughness = m.roughness; b->color = tex(m.tex_color); b->diffuse = tex(m.tex_diffuse); o.bxdf.reset(b); break; }
            case 8: { auto b = new BxDFLTCDiffuse<LTC::GGX>; b->roughness = m.roughness; b->color = tex(m.tex_color); b->diffuse = tex(m.tex_diffuse); o.bxdf.reset(b); break; }
            default: throw std::runtime_error("bad material kind in the scene file");
            }
        }
        std::vector<Triangle> tris;
        for (uint32_t i = 0; i < nt; i++) tris.emplace_back(&scene, F[3 * i], F[3 * i + 1], F[3 * i + 2], materials[FM[i]].get());
        scene.triangles = tris.data(); scene.n_triangles = nt;
        const uint32_t npl = r.get<uint32_t>();
        for (uint32_t i = 0; i < npl; i++) {
            Light l(Light::FULL_SPHERE);
            const float px = r.get<float>(), py = r.get<float>(), pz = r.get<float>(); // (not as call arguments: their evaluation order is unspecified)
            l.pos = glm::vec3(px, py, pz);
            const float cr = r.get<float>(), cg = r.get<float>(), cb = r.get<float>();
            l.color = Radiance(cr, cg, cb);
            l.intensity = r.get<float>(); l.size = r.get<float>();
            scene.pointlights.push_back(l);
        }
        const uint32_t nal = r.get<uint32_t>();
        std::vector<uint32_t> offs = r.arr<uint32_t>(nal + 1);
        std::vector<uint32_t> atris = r.arr<uint32_t>(offs[nal]);
        for (uint32_t i = 0; i < nal; i++) {
            Scene::ArealLight al;
            for (uint32_t j = offs[i]; j < offs[i + 1]; j++) al.triangles_with_areas.push_back({0.0f, atris[j]});
            scene.areal_lights.push_back({0.0f, al});
        }
        const uint32_t sky_mode = r.get<uint32_t>();
        const float sr = r.get<float>(), sg = r.get<float>(), sb = r.get<float>(), sky_i = r.get<float>(), sky_rot = r.get<float>();
        const int32_t sky_tex = r.get<int32_t>();
        if (sky_mode == 0) scene.SetSkyboxColor(Color(sr, sg, sb), sky_i); else scene.SetSkyboxTexture(tex(sky_tex), sky_i, sky_rot);
        fill_ltc(r, g_ggx_m, g_ggx_a);
        fill_ltc(r, g_bek_m, g_bek_a);
        // ---- Camera (ConfigJSON::GetCamera's constructor call, src/config.cpp:369) and Config
        float pos[3], la[3], up[3];
        for (float& v : pos) v = r.get<float>();
        for (float& v : la) v = r.get<float>();
        for (float& v : up) v = r.get<float>();
        const float yview = r.get<float>(), xview = r.get<float>();
        const int32_t xsize = r.get<int32_t>(), ysize = r.get<int32_t>();
        const float focus_plane = r.get<float>(), lens_size = r.get<float>();
        rgk_camera rc;
        if (rgk_camera_init(&rc, pos, la, up, yview, xview, xsize, ysize, focus_plane, lens_size) != RGK_OK) throw std::runtime_error(rgk_last_error());
        Camera camera;
        camera.origin = glm::vec3(rc.origin[0], rc.origin[1], rc.origin[2]); camera.lookat = glm::vec3(la[0], la[1], la[2]);
        camera.direction = glm::vec3(rc.direction[0], rc.direction[1], rc.direction[2]);
        camera.cameraup = glm::vec3(rc.cameraup[0], rc.cameraup[1], rc.cameraup[2]); camera.cameraleft = glm::vec3(rc.cameraleft[0], rc.cameraleft[1], rc.cameraleft[2]);
        camera.viewscreen = glm::vec3(rc.viewscreen[0], rc.viewscreen[1], rc.viewscreen[2]);
        camera.viewscreen_x = glm::vec3(rc.viewscreen_x[0], rc.viewscreen_x[1], rc.viewscreen_x[2]);
        camera.viewscreen_y = glm::vec3(rc.viewscreen_y[0], rc.viewscreen_y[1], rc.viewscreen_y[2]);
        camera.lens_size = rc.lens_size; camera.xsize = rc.xsize; camera.ysize = rc.ysize;
        auto cfg = std::make_shared<Config>();
        cfg->xres = r.get<uint32_t>(); cfg->yres = r.get<uint32_t>(); cfg->multisample = r.get<uint32_t>(); cfg->recursion_level = r.get<uint32_t>();
        cfg->clamp = r.get<float>(); cfg->russian = r.get<float>(); cfg->bumpmap_scale = r.get<float>(); cfg->reverse = r.get<uint32_t>();
        cfg->render_rounds = r.get<uint32_t>();

        // ---- RenderFrame (src/render_driver.cpp:192-253)
        RgkBinding hip;
        hip.MakeDeviceScene(scene, 0);
        uint32_t n_tiles = 0;
        if (rgk_generate_task_list(32, cfg->xres, cfg->yres, cfg->xres / 2.0f, cfg->yres / 2.0f, 0, 0, nullptr, &n_tiles) != RGK_OK) throw std::runtime_error(rgk_last_error());
        std::vector<rgk_tile> order(n_tiles);
        if (rgk_generate_task_list(32, cfg->xres, cfg->yres, cfg->xres / 2.0f, cfg->yres / 2.0f, 0, 0, order.data(), &n_tiles) != RGK_OK) throw std::runtime_error(rgk_last_error());
        std::vector<RenderTask> tasks; // GenerateTaskList(TILE_SIZE, xres, yres, midpoint) with the core's tie order
        for (const rgk_tile& t : order) tasks.emplace_back(cfg->xres, cfg->yres, t.x0, t.x1, t.y0, t.y1);
        EXRTexture total_ob(cfg->xres, cfg->yres);
        unsigned int seedcount = 0, seedstart = 42;
        unsigned long long rays_done = 0;
        if (!device_accum) {
            for (unsigned int roundno = 0; roundno < cfg->render_rounds; roundno++) hip.RenderRound(cfg, camera, tasks, seedcount, seedstart, total_ob, rays_done);
        } else {
            // one rank of a multi-GPU host: a per-round device accumulator, the rank's share of the
            // tiles, ONE RCCL reduce per round, the root adds the round's sum to its frame total.
            // --emulate-world W: the W ranks of such a host one after the other on this one GPU, each with its own round
            // accumulator, the in-place reduce replaced by what it computes (root += every other rank's buffer, the others keep
            // theirs) -- the arithmetic of the pattern over several rounds, without needing W GPUs.
            std::vector<rgk_accum*> round_acc((size_t)emulate_world, nullptr);
            rgk_accum* total = nullptr;
            rgk_comm* comm = nullptr;
            uint8_t id[RGK_COMM_ID_BYTES];
            const int world = emulate_world;
            for (auto& a : round_acc) if (rgk_accum_create(cfg->xres, cfg->yres, 0, &a) != RGK_OK) throw std::runtime_error(rgk_last_error());
            if (rgk_accum_create(cfg->xres, cfg->yres, 0, &total) != RGK_OK) throw std::runtime_error(rgk_last_error());
            if (world == 1 && (rgk_comm_get_unique_id(id) != RGK_OK || rgk_comm_create(id, 0, 1, 0, &comm) != RGK_OK)) throw std::runtime_error(rgk_last_error());
            rgk_params p = {cfg->xres, cfg->yres, cfg->multisample, cfg->recursion_level, cfg->clamp, cfg->russian, cfg->bumpmap_scale, 0u, cfg->reverse, RGK_SAMPLER_HALTON, 0u};
            for (unsigned int roundno = 0; roundno < cfg->render_rounds; roundno++) {
                std::vector<rgk_tile> tiles(n_tiles), mine(n_tiles);
                uint32_t n = n_tiles;
                if (rgk_generate_task_list(32, cfg->xres, cfg->yres, cfg->xres / 2.0f, cfg->yres / 2.0f, seedstart, seedcount, tiles.data(), &n) != RGK_OK) throw std::runtime_error(rgk_last_error());
                seedcount += n;
                for (int rank = 0; rank < world; rank++) {
                    rgk_accum* acc = round_acc[(size_t)rank];
                    uint32_t n_mine = n_tiles;
                    if (rgk_shard_tiles(tiles.data(), n, rank, world, mine.data(), &n_mine) != RGK_OK) throw std::runtime_error(rgk_last_error());
                    if (rgk_accum_clear(acc) != RGK_OK) throw std::runtime_error(rgk_last_error());
                    rgk_counters cnt;
                    if (rgk_render_round_device(hip.scene, &rc, &p, mine.data(), n_mine, rgk_accum_rgb(acc), rgk_accum_count(acc), &cnt) != RGK_OK) throw std::runtime_error(rgk_last_error());
                    rays_done += cnt.path_rays;
                }
                if (comm) { // the real collective (one rank here: the sum of one buffer)
                    if (rgk_accum_reduce(comm, rgk_accum_rgb(round_acc[0]), rgk_accum_count(round_acc[0]), cfg->xres, cfg->yres, 0) != RGK_OK) throw std::runtime_error(rgk_last_error());
                } else      // what rgk_accum_reduce leaves behind: the root holds the sum, every other rank its own contribution
                    for (int rank = 1; rank < world; rank++) if (rgk_accum_add(round_acc[0], round_acc[(size_t)rank]) != RGK_OK) throw std::runtime_error(rgk_last_error());
                if (rgk_accum_add(total, round_acc[0]) != RGK_OK) throw std::runtime_error(rgk_last_error()); // root: total_ob.Accumulate(output_buffer)
            }
            if (rgk_accum_download(total, &total_ob.data[0].r, total_ob.count.data()) != RGK_OK) throw std::runtime_error(rgk_last_error());
            rgk_comm_destroy(comm);
            for (auto a : round_acc) rgk_accum_destroy(a);
            rgk_accum_destroy(total);
        }
        std::ofstream o(argv[2], std::ios::binary);
        const uint64_t rd = rays_done;
        o.write(reinterpret_cast<const char*>(&rd), 8);
        o.write(reinterpret_cast<const char*>(total_ob.data.data()), total_ob.data.size() * sizeof(Radiance));
        o.write(reinterpret_cast<const char*>(total_ob.count.data()), total_ob.count.size() * sizeof(unsigned int));
        printf("caller ok: %u x %u x %u spp, %u rounds, %llu path rays\n", cfg->xres, cfg->yres, cfg->multisample, cfg->render_rounds, rays_done);
        return 0;
    } catch (const std::exception& e) {
        fprintf(stderr, "caller failed: %s\n", e.what());
        return 1;
    }
}
