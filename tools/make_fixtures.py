#!/usr/bin/env python3
"""Generate the committed fixtures under tests/golden/ (run in the build container only;
needs /root/reference).  Fixtures are DATA: inputs and expected outputs.

  halton_faure.npz   HS::Halton_sampler::sample(dim, index) after init_faure() from the
                     reference's own external/halton_sampler.h, compiled where it lies
                     (oracle/_ref/halton_ref): 256 dims x (256 leading + 64 large indices).
  rgk_amd/data/cornell_scene.npz  the flat arrays ConfigJSON::Install* would hand to Scene for
                     scenes/cornell-box.json (built by rgk_amd.config from the reference's
                     config file), plus camera / render parameters.
  cornell_config0_half.npz  the oracle's accumulator for BASELINE configs[0] at half resolution (128x128x16 spp).
  scene_<name>.npz   the same for the other scenes the reference ships complete (mesh + textures):
                     rubiks-bump (PNG texture + bump map, point light), cube3 (8966 faces, global
                     ltc_beckmann override, sphere light), box6 (17 k triangles with uv, emissive
                     triangles, reverse = 3), cornell-box-spheres (LTC Beckmann + dielectric spheres).
  rgk_amd/data/sponza_textures_u8.npz  the 17 JPGs shipped under scenes/sponza-fixed/ as decoded bytes (the Sponza proxy's
                     textures; the mesh itself is absent from the reference checkout).
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
REF = os.environ.get("RGK_REFERENCE", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")


def halton():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    rng = np.random.default_rng(20261004)
    idx = np.concatenate([np.arange(256, dtype=np.uint64), rng.integers(0, 2 ** 32, 64, dtype=np.uint64)]).astype(np.uint32)
    out = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "halton_ref"), "256"] + [str(int(i)) for i in idx],
                         capture_output=True, check=True).stdout
    vals = np.frombuffer(out, dtype=np.float32).reshape(256, len(idx))
    np.savez_compressed(os.path.join(GOLD, "halton_faure.npz"), index=idx, values=vals)
    print("halton_faure.npz", vals.shape)


def cornell():
    from rgk_amd.config import Config
    cfg = Config(os.path.join(REF, "scenes", "cornell-box.json"))
    sb = cfg.build_scene()
    cam = cfg.get_camera()
    extra = dict(camera=dict(pos=list(cam.ctor["pos"]), lookat=list(cam.ctor["lookat"]), up=list(cam.ctor["up"]), fov=19.5),
                 xres=cfg.xres, yres=cfg.yres, multisample=cfg.multisample, depth=cfg.recursion_level,
                 clamp=float(cfg.clamp), russian=float(cfg.russian), bumpscale=float(cfg.bumpmap_scale),
                 reverse=cfg.reverse, source="scenes/cornell-box.json")
    sb.save_npz(os.path.join(ROOT, "rgk_amd", "data", "cornell_scene.npz"), extra)
    print("cornell_scene.npz", len(sb.V), "vertices", len(sb.F), "triangles")


# scenes the reference ships COMPLETE (config + mesh + textures), converted to the flat arrays its loader
# would hand to Scene; the .npz holds data only (geometry, decoded texel bytes, scalars)
REFERENCE_SCENES = ["rubiks-bump", "cube3", "box6", "cornell-box-spheres"]


def reference_scene(name):
    from rgk_amd.config import Config
    cfg = Config(os.path.join(REF, "scenes", name + ".json"))
    sb = cfg.build_scene()
    cfg.get_camera()
    cam = cfg.root.d["camera"]
    extra = dict(camera=dict(pos=cam["position"], lookat=cam["lookat"], up=cam.get("upvector", [0.0, 1.0, 0.0]),
                             fov=cam.get("fov"), focal=cam.get("focal"), focus_plane=cam.get("focus-plane", 1.0),
                             lens_size=cam.get("lens-size", 0.0)),
                 xres=cfg.xres, yres=cfg.yres, multisample=cfg.multisample, depth=cfg.recursion_level,
                 clamp=float(cfg.clamp), russian=float(cfg.russian), bumpscale=float(cfg.bumpmap_scale),
                 reverse=cfg.reverse, source="scenes/%s.json" % name)
    out = os.path.join(GOLD, "scene_%s.npz" % name)
    sb.save_npz(out, extra)
    print(os.path.basename(out), len(sb.V), "vertices", len(sb.F), "triangles", os.path.getsize(out) // 1024, "KiB")


def cornell_image():
    """BASELINE configs[0] (cornell 256x256x16, scenes/cornell-box.json) at half resolution, one round, rendered by
    the oracle: the frozen expected accumulator both the oracle (drift guard) and the GPU are compared with."""
    from oracle import rgk_oracle as O
    from rgk_amd.workloads import Workload
    wl = Workload("cornell-256", scale=0.5)
    o = O.OracleScene(wl.builder.to_desc())
    acc, cnt, k = o.render_round(wl.camera, wl.params(), O.generate_task_list(wl.xres, wl.yres))
    np.savez_compressed(os.path.join(GOLD, "cornell_config0_half.npz"), accum=acc, count=cnt,
                        counters=np.array([k.paths, k.path_rays, k.shadow_rays], dtype=np.uint64))
    print("cornell_config0_half.npz", acc.shape, int(k.paths), "paths")


def sponza_textures():
    """The 17 JPGs the reference ships for Sponza (scenes/sponza-fixed/*.JPG; Dabrovic Sponza, (c) 2002 Marko Dabrovic, bump
    maps by Morgan McGuire -- scenes/sponza-fixed/copyright.txt), decoded ONCE here with PIL into the bytes an 8-bit loader
    holds (h, w, 3 uint8, top row first): both the oracle and the HIP path then see the same texels whatever JPEG decoder
    a machine has.  The proxy geometry is uv-mapped onto them through the 20 sponza.mtl materials."""
    from PIL import Image
    d = os.path.join(REF, "scenes", "sponza-fixed")
    arrs = {}
    for f in sorted(os.listdir(d)):
        if f.lower().endswith(".jpg"):
            arrs[f] = np.asarray(Image.open(os.path.join(d, f)).convert("RGB"), dtype=np.uint8)
    out = os.path.join(ROOT, "rgk_amd", "data", "sponza_textures_u8.npz")
    np.savez_compressed(out, **arrs)
    print("sponza_textures_u8.npz", len(arrs), "images", sum(a.nbytes for a in arrs.values()), "bytes decoded,", os.path.getsize(out), "on disk")


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "sponza_textures":
        sponza_textures()
        sys.exit(0)
    halton()
    cornell()
    cornell_image()
    for n in REFERENCE_SCENES:
        reference_scene(n)
    sponza_textures()
