"""CPU suite for the host side: config front-end, C-ABI surface, tile sharding + reduce (gloo)."""
import ctypes as C
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from rgk_amd import capi

REF_DIR = os.environ.get("RGK_REFERENCE", "/root/reference")
from rgk_amd.config import Config, ConfigFileException, parse_json, strip_json_comments
from rgk_amd.scene import SceneBuilder, primitive_data

from conftest import ROOT


# ----------------------------------------------------------------------- C ABI
def test_library_exports_every_symbol_the_header_declares(product_lib):
    hdr = open(os.path.join(ROOT, "include", "rgk.h")).read()
    declared = sorted(set(re.findall(r"\b(rgk_[a-z_]+)\s*\(", hdr)))
    assert declared == sorted(capi.EXPORTS)
    for name in declared:
        assert getattr(product_lib, name) is not None


def test_struct_layouts_match_the_header(tmp_path):
    """sizeof() of every struct, C compiler vs ctypes mirror."""
    src = tmp_path / "sz.c"
    names = ["rgk_material", "rgk_texture", "rgk_pointlight", "rgk_scene_desc", "rgk_camera", "rgk_params", "rgk_tile",
             "rgk_counters", "rgk_scene_info", "rgk_hit", "rgk_progress"]
    src.write_text('#include <stdio.h>\n#include "rgk.h"\nint main(){' +
                   "".join(f'printf("%zu\\n", sizeof({n}));' for n in names) + "return 0;}")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    mirrors = [capi.Material, capi.Texture, capi.PointLight, capi.SceneDesc, capi.Camera, capi.Params, capi.Tile,
               capi.Counters, capi.SceneInfo, capi.Hit, capi.Progress]
    assert sizes == [C.sizeof(m) for m in mirrors]


def test_product_task_list_equals_oracle(product_lib, oracle):
    from rgk_amd import render_driver as rd
    for (w, h) in ((1920, 1080), (100, 70), (32, 32), (1, 1), (33, 65)):
        a = rd.generate_task_list(w, h, 42, 7)
        b = oracle.generate_task_list(w, h, 42, 7)
        assert len(a) == len(b) == ((w + 31) // 32) * ((h + 31) // 32)
        assert [(t.x0, t.x1, t.y0, t.y1, t.seed) for t in a] == [(t.x0, t.x1, t.y0, t.y1, t.seed) for t in b]


def test_host_argument_errors_need_no_gpu(product_lib):
    n = C.c_uint32(1)
    tiles = (capi.Tile * 1)()
    assert product_lib.rgk_generate_task_list(32, 100, 100, 50.0, 50.0, 42, 0, tiles, C.byref(n)) == -1   # buffer too small
    assert b"too small" in product_lib.rgk_last_error()
    assert product_lib.rgk_generate_task_list(0, 100, 100, 50.0, 50.0, 42, 0, None, C.byref(n)) == -1
    h = C.c_void_p()
    assert product_lib.rgk_scene_create(None, 0, C.byref(h)) == -1                                        # null descriptor
    d = capi.SceneDesc()
    assert product_lib.rgk_scene_create(C.byref(d), 0, C.byref(h)) == -1 and not h.value                  # empty scene


def test_camera_constructor_equals_the_oracles(product_lib, oracle):
    """rgk_camera_init (host code of the product) against the oracle's own restatement of Camera::Camera
    (src/camera.cpp:7-24): the derived members RenderRound's `const Camera&` carries, bit for bit."""
    rng = np.random.default_rng(11)
    L = oracle.lib()
    for k in range(200):
        pos, la = rng.normal(size=3).astype(np.float32) * 10, rng.normal(size=3).astype(np.float32) * 10
        up = (0.0, 1.0, 0.0) if k % 2 else tuple(rng.normal(size=3).astype(np.float32))
        yview, xview = np.float32(rng.uniform(0.2, 2.0)), np.float32(rng.uniform(0.2, 2.0))
        fp, ls = np.float32(rng.uniform(0.5, 5.0)), np.float32(0.0 if k % 3 else 0.1)
        a, b = capi.Camera(), capi.Camera()
        args = (capi.f3(*map(float, pos)), capi.f3(*map(float, la)), capi.f3(*map(float, up)), float(yview), float(xview), 640, 480, float(fp), float(ls))
        assert product_lib.rgk_camera_init(C.byref(a), *args) == 0
        assert L.orc_camera_init(C.byref(b), *args) == 0
        assert bytes(a) == bytes(b)
    assert product_lib.rgk_camera_init(None, *args) == -1


def test_mix_nesting_is_validated_without_a_gpu(product_lib):
    """BxDFMix recurses (bxdf.cpp:235-249); the kernels evaluate two levels.  Deeper nesting and cycles are refused by
    rgk_scene_create's descriptor check (before any device is touched) instead of being rendered wrong."""
    from rgk_amd.scene import SceneBuilder
    def scene_with(mats):
        sb = SceneBuilder.load_npz(os.path.join(ROOT, "rgk_amd", "data", "cornell_scene.npz"))
        base = dict(sb.materials[0])
        sb.materials = [dict(base, name=f"x{i}", **m) for i, m in enumerate(mats)] + sb.materials
        sb.tri_mat = [sb.FM + len(mats)]
        sb.FM = None
        return sb
    leaf = dict(kind=capi.BXDF_DIFFUSE)
    mix = lambda a, b: dict(kind=capi.BXDF_MIX, mix_m1=a, mix_m2=b, amount=0.5)
    h = C.c_void_p()
    ok2 = scene_with([leaf, leaf, mix(0, 1), mix(2, 0)])            # mix of (mix of leaves): two levels
    rc = product_lib.rgk_scene_create(C.byref(ok2.to_desc()), 0, C.byref(h))
    assert rc in (0, -4), product_lib.rgk_last_error()               # accepted: created, or no device in this container
    if rc == 0:
        product_lib.rgk_scene_destroy(h)
    deep = scene_with([leaf, leaf, mix(0, 1), mix(2, 0), mix(3, 1)])  # three levels
    assert product_lib.rgk_scene_create(C.byref(deep.to_desc()), 0, C.byref(h)) == -5
    assert b"nested 3 levels" in product_lib.rgk_last_error()
    cyc = scene_with([leaf, mix(2, 0), mix(1, 0)])                    # 1 -> 2 -> 1
    assert product_lib.rgk_scene_create(C.byref(cyc.to_desc()), 0, C.byref(h)) == -1
    assert b"cycle" in product_lib.rgk_last_error()


def test_shard_tiles_is_a_round_robin_deal(product_lib):
    from rgk_amd import render_driver as rd
    tiles = rd.generate_task_list(1920, 1080, 42, 5)
    seen = []
    for rank in range(8):
        mine = rd.shard_tiles(tiles, rank, 8)
        assert [(t.x0, t.y0, t.seed) for t in mine] == [(tiles[i].x0, tiles[i].y0, tiles[i].seed) for i in range(rank, len(tiles), 8)]
        seen += [(t.x0, t.y0) for t in mine]
    assert sorted(seen) == sorted((t.x0, t.y0) for t in tiles)
    n = C.c_uint32(0)
    assert product_lib.rgk_shard_tiles(tiles, len(tiles), 8, 8, None, C.byref(n)) == -1


def test_product_fails_loudly_without_the_extension(monkeypatch):
    monkeypatch.setattr(capi, "LIB_PATH", "/nonexistent/librgk_hip.so")
    monkeypatch.setattr(capi, "_product", None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        capi.load_product()


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rgk_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                # comments may name the oracle; code must not import, include, link or call it
                for needle in ("import oracle", "from oracle", "rgk_oracle", "librgk_oracle", "orc_", "oracle/", "rgk_cpu"):
                    assert needle not in text, (f, needle)


# ----------------------------------------------------------------------- config front-end (f1, minimal)
def test_json_comments_and_defaults(tmp_path):
    p = tmp_path / "s.json"
    p.write_text('''{ // line comment
        "output-file": "a.exr", /* block */ "output-width": 64, "output-height": 48,
        "camera": {"position": [0,1,5], "lookat": [0,1,0], "fov": 30},
        "materials": [{"name": "m", "brdf": "diffuse", "diffuse255": [255, 128, 0], "emission": [1,2,3]},
                      {"name": "g", "brdf": "ltc_ggx_diffuse", "exponent": 200, "specular": [0.3,0.3,0.3], "url": "http://x//y"}],
        "scene": [{"primitive": "cube", "material": "m", "translate": [0,0.5,0]},
                  {"primitive": "plane", "material": "g", "scale": [4,1,4]}],
        "lights": [{"position": [0,5,0], "intensity": 10, "color255": [255,255,255]}],
        "unused-key": 1
    }''')
    cfg = Config(str(p))
    assert (cfg.xres, cfg.yres, cfg.multisample, cfg.recursion_level, cfg.reverse) == (64, 48, 1, 40, 0)
    assert abs(cfg.russian - 0.74) < 1e-7 and cfg.clamp == 1e7 and cfg.bumpmap_scale == 1.0 and cfg.output_scale == -1.0
    sb = cfg.build_scene().finalize()
    assert len(sb.F) == 12 + 2 and len(sb.areal) == 1 and len(sb.areal[0]) == 12
    m, g = sb.materials
    assert sb.textures[m["tex_diffuse"]]["color"] == pytest.approx((1.0, 128 / 255, 0.0))
    assert g["kind"] == capi.BXDF_LTC_GGX_DIFFUSE and g["roughness"] == pytest.approx((2 / 202) ** 0.5)
    assert sb.textures[g["tex_diffuse"]]["color"] == (0.0, 0.0, 0.0)          # LTC diffuse fallback is black
    cam = cfg.get_camera()
    assert cam.ctor["yview"] == pytest.approx(cam.ctor["xview"] * 48 / 64)
    assert "unused-key" in cfg.perform_post_check() and "url" not in cfg.perform_post_check()
    assert sb.pointlights[0]["color"] == (1.0, 1.0, 1.0)


def _canonical(v):
    """The canonical text form oracle/ref_jsoncpp_main.cpp prints for a parsed JSON tree."""
    if v is None:
        return "null"
    if isinstance(v, bool):
        return "true" if v else "false"
    if isinstance(v, int):
        return ("i%d" % v) if v <= 2 ** 63 - 1 else ("u%d" % v)
    if isinstance(v, float):
        return "r%s" % ("%.17g" % v)
    if isinstance(v, str):
        return '"' + "".join("\\" + c if c in '"\\' else ("\\u%04x" % ord(c) if ord(c) < 0x20 else c) for c in v) + '"'
    if isinstance(v, list):
        return "[" + ",".join(_canonical(x) for x in v) + "]"
    return "{" + ",".join('"%s":%s' % (k, _canonical(v[k])) for k in sorted(v)) + "}"


def test_json_front_end_parses_every_shipped_scene_like_the_references_jsoncpp():
    """f1 pinned: the reference's own JSON reader (external/jsoncpp.cpp, compiled where it lies into oracle/_ref/jsoncpp_ref,
    called as src/config.cpp:266-272 calls it) against rgk_amd.config's comment stripping + json parse, on all the scene
    files the reference ships: identical trees -- keys, nesting, strings, integer-vs-real typing, every number to 17 digits."""
    exe = os.path.join(ROOT, "oracle", "_ref", "jsoncpp_ref")
    scenes = os.path.join(REF_DIR, "scenes")
    if not (os.path.exists(exe) and os.path.isdir(scenes)):
        pytest.skip("oracle/_ref/jsoncpp_ref or the reference's scenes are absent on this box")
    files = sorted(os.path.join(d, f) for d, _, fs in os.walk(scenes) for f in fs if f.endswith(".json") or f.endswith(".rtc"))
    assert len(files) >= 39
    out = subprocess.run([exe] + files, capture_output=True, text=True, check=True).stdout.strip().split("\n")
    assert len(out) == len(files)
    n_comments = 0
    for line, path in zip(out, files):
        p, tree = line.split("\t", 1)
        assert p == path and tree != "ERROR", path
        text = open(path).read()
        n_comments += ("//" in text) or ("/*" in text)
        mine = _canonical(parse_json(text))
        assert mine == tree, path
    assert n_comments >= 5   # the comment syntax is exercised (scenes/cornell-box.json and others carry comments)


def _write_hdr(path, img_rgbe, rle, header_extra=b"", first=b"#?RADIANCE"):
    """A Radiance file from (h, w, 4) RGBE bytes; rle: new-style per-channel run-length scan lines (runs and dumps)."""
    h, w = img_rgbe.shape[:2]
    out = [first + b"\n", header_extra, b"FORMAT=32-bit_rle_rgbe\n", b"\n", b"-Y %d +X %d\n" % (h, w)]
    if not rle:
        out.append(img_rgbe.tobytes())
    else:
        for j in range(h):
            out.append(bytes([2, 2, w >> 8, w & 255]))
            for k in range(4):
                row, i = img_rgbe[j, :, k], 0
                while i < w:
                    run = 1
                    while i + run < w and run < 127 and row[i + run] == row[i]:
                        run += 1
                    if run >= 3:
                        out.append(bytes([128 + run, int(row[i])])); i += run
                    else:
                        n = 1
                        while i + n < w and n < 128 and not (i + n + 2 < w and row[i + n] == row[i + n + 1] == row[i + n + 2]):
                            n += 1
                        out.append(bytes([n]) + row[i:i + n].tobytes()); i += n
    open(path, "wb").write(b"".join(out))


def test_hdr_decoder_equals_the_references_stb_image(tmp_path):
    """f1 pinned: rgk_amd.scene.load_hdr against the decoder the reference vendors and calls (external/stb_image.h through
    stbi_loadf, src/texture.cpp:294-321; compiled where it lies into oracle/_ref/stbi_ref) on run-length encoded, flat and
    narrow Radiance files: the same floats bit for bit."""
    from rgk_amd.scene import load_hdr
    exe = os.path.join(ROOT, "oracle", "_ref", "stbi_ref")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/stbi_ref not built (reference absent on this box)")
    rng = np.random.default_rng(9)
    cases = []
    for (h, w, rle) in ((17, 64, True), (9, 300, True), (5, 40, False), (6, 5, False), (3, 8, True)):
        img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        img[..., 3] = rng.integers(100, 150, (h, w))                # exponents around 128
        img[1:3, : w // 2] = img[1, 0]                               # long runs
        img[0, ::7, 3] = 0                                           # zero exponent -> black
        if not rle:
            img[0, 0, :3] = (200, 100, 50)                           # a first pixel that cannot be mistaken for a scan-line header
        cases.append((img, rle))
    for n, (img, rle) in enumerate(cases):
        path = str(tmp_path / f"t{n}.hdr")
        _write_hdr(path, img, rle, header_extra=b"# made by the test\nEXPOSURE=1.0\n")
        out = str(tmp_path / f"t{n}.bin")
        subprocess.run([exe, path, out], check=True)
        b = open(out, "rb").read()
        w, h, c = np.frombuffer(b, dtype=np.int32, count=3)
        ref = np.frombuffer(b, dtype=np.float32, offset=12).reshape(h, w, c)
        mine = load_hdr(path)
        assert mine.shape == ref.shape == img.shape[:2] + (3,)
        assert np.array_equal(mine.view(np.uint32), ref.view(np.uint32)), n
    bad = str(tmp_path / "bad.hdr")
    _write_hdr(bad, cases[0][0], True, first=b"#?RGBE")              # this stb version knows only #?RADIANCE
    assert subprocess.run([exe, bad, str(tmp_path / "bad.bin")], capture_output=True).returncode != 0
    with pytest.raises(ValueError):
        load_hdr(bad)


def test_rtc_config_format(tmp_path):
    """ConfigRTC (src/config.cpp:27-258): the reference's line-based format.  Its only shipped .rtc file holds JSON
    (scenes/sponza.rtc) and is refused like the reference refuses it; a well-formed file parses to the same
    fields, with the base-class defaults where it is silent."""
    from rgk_amd.config import ConfigRTC, load_config
    p = tmp_path / "scene.rtc"
    p.write_text("a comment\ncubes/cube3.obj\nout.exr\n7\n640 480\n1.5 2.85 -4.0\n1.0 1.0 1.0\n0 1 0\n1.2\n# c\n"
                 "L 3.0 6.0 -2.0 255 255 200 400.0 0.4\nms 16\nsky 145 200 235 0.3\nclamp 10\nroulette 0.6\nbrdf diffuse\nreverse 2\nbogus 1\n")
    c = load_config(str(p))
    assert isinstance(c, ConfigRTC)
    assert (c.comment, c.model_file, c.output_file, c.recursion_level, c.xres, c.yres) == ("a comment", "cubes/cube3.obj", "out.exr", 7, 640, 480)
    assert c.multisample == 16 and c.reverse == 2 and c.brdf == "diffusecosine" and float(c.clamp) == 10.0 and abs(float(c.russian) - 0.6) < 1e-7
    assert float(c.bumpmap_scale) == 10.0 and c.render_rounds == 1 and float(c.sky_brightness) == pytest.approx(0.3)      # bumpscale: base-class default
    assert c.lights == [dict(pos=(3.0, 6.0, -2.0), color=(1.0, 1.0, float(np.float32(200) / np.float32(255))), intensity=400.0, size=float(np.float32(0.4)))]
    assert c.perform_post_check() == ["WARNING: Unrecognized option `bogus` in the config file."]
    cam = c.get_camera()
    assert cam.ctor["yview"] == pytest.approx(1.2) and cam.ctor["xview"] == pytest.approx(1.2 * 640 / 480)
    prm = c.get_params()
    assert (prm.xres, prm.yres, prm.multisample, prm.depth, prm.reverse) == (640, 480, 16, 7, 2)
    short = tmp_path / "short.rtc"
    short.write_text("c\nm.obj\no.exr\n3\n64 48")
    with pytest.raises(ConfigFileException, match="prematurely"):
        load_config(str(short))
    ref = os.path.join(REF_DIR, "scenes", "sponza.rtc")
    if os.path.exists(ref):
        with pytest.raises(ConfigFileException):
            load_config(ref)                                          # JSON inside a .rtc: std::stoi of `"output-width": 1200,` (F7)
    with pytest.raises(ConfigFileException, match="not recognized"):
        load_config(str(tmp_path / "x.txt"))


def test_config_errors(tmp_path):
    def cfg(d):
        p = tmp_path / "e.json"
        p.write_text(json.dumps(d))
        return Config(str(p))
    base = {"output-file": "a", "output-width": 8, "output-height": 8}
    with pytest.raises(ConfigFileException):
        Config(str(tmp_path / "missing.json"))
    with pytest.raises(ConfigFileException):
        cfg({"output-width": 8})
    with pytest.raises(ConfigFileException):
        cfg(dict(base, rounds=1, **{"render-time": 2}))
    c = cfg(dict(base, materials=[{"name": "m", "brdf": "cooktorr"}], scene=[]))
    with pytest.raises(ConfigFileException, match="Unsupported BRDF"):
        c.build_scene()
    c = cfg(dict(base, scene=[{"primitive": "plane", "material": "nope"}]))
    with pytest.raises(RuntimeError, match="was not defined"):
        c.build_scene()
    c = cfg(dict(base))
    with pytest.raises(ConfigFileException, match="neither"):
        c.build_scene()
    assert strip_json_comments('{"a": "x//y" /* c */ } // t') == '{"a": "x//y"  } '
    assert parse_json('{"a": "x//y" /* c */, "b": [000.50, -07, 1e2, "\\u00e9\\n"] } // t') == {"a": "x//y", "b": [0.5, -7, 100.0, "\u00e9\n"]}
    for bad in ('{"a": [1, 2,]}', '{"a" 1}', '{1: 2}', '[1 2]', '"abc'):
        with pytest.raises(ConfigFileException):
            parse_json(bad)


def test_builtin_primitives_match_reference_shapes():
    for kind, ntri in (("plane", 2), ("tri", 1), ("cube", 12)):
        pos, nrm, uv, tan = primitive_data(kind)
        assert len(pos) == 3 * ntri
        tri = pos.reshape(-1, 3, 3)
        gn = np.cross(tri[:, 2] - tri[:, 0], tri[:, 1] - tri[:, 0])     # CalculatePlane's cross(d1, d0)
        gn /= np.linalg.norm(gn, axis=1, keepdims=True)
        assert np.allclose(np.abs(np.sum(gn * nrm.reshape(-1, 3, 3)[:, 0], axis=1)), 1)   # faces are planar, normals axis-aligned
        assert np.allclose(np.sum(nrm * tan, axis=1), 0)
    pos, _, uv, _ = primitive_data("plane")
    assert pos[0].tolist() == [1, 0, 1] and uv[0].tolist() == [1, 1] and pos[3].tolist() == [-1, 0, -1]


def test_cornell_fixture_matches_reference_config():
    ref = "/root/reference/scenes/cornell-box.json"
    if not os.path.exists(ref):
        pytest.skip("reference not present on this box")
    a = Config(ref).build_scene().finalize()
    b = SceneBuilder.load_npz(os.path.join(ROOT, "rgk_amd", "data", "cornell_scene.npz"))
    assert np.array_equal(a.V, b.V) and np.array_equal(a.F, b.F) and np.array_equal(a.N, b.N) and np.array_equal(a.FM, b.FM)
    assert len(a.F) == 36 and len(a.V) == 108 and a.areal == b.areal == [[34], [35]]   # 36 triangles, 2 lights


def test_obj_loader_on_reference_meshes():
    ref = "/root/reference/scenes/cubes/cube3.obj"
    if not os.path.exists(ref):
        pytest.skip("reference not present on this box")
    sb = SceneBuilder()
    sb.load_obj(ref, np.eye(4, dtype=np.float32))
    sb.finalize()
    assert len(sb.F) > 8000 and np.isfinite(sb.N).all() and sb.F.max() < len(sb.V)
    assert all(m["kind"] == capi.BXDF_LTC_GGX_DIFFUSE for m in sb.materials)


# ----------------------------------------------------------------------- multi-GPU path on CPU (gloo, world_size 2 and 4)
def _case_workload(case):
    from rgk_amd.workloads import SceneFixture, Workload
    if case == "reverse":  # emissive triangles + BDPT light sub-paths: splats make the reduce a true sum
        return SceneFixture(os.path.join(ROOT, "tests", "golden", "scene_box6.npz"), scale=0.04, spp=2, depth=3)
    return Workload("cornell-256", scale=0.25, spp=4)


def _rank_main(rank, world, port, q, case):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from oracle import rgk_oracle as O
    from rgk_amd import render_driver as rd
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    wl = _case_workload(case)
    osc = O.OracleScene(wl.builder.to_desc())

    class HostScene:  # stands in for the GPU scene: same entry point, host pointers, oracle underneath
        device = 0

        def render_round_device(self, camera, params, tiles, d_accum, d_count):
            cnt = capi.Counters()
            O.lib().orc_render_round(osc.h, C.byref(camera), C.byref(params), tiles, len(tiles), d_accum, d_count, C.byref(cnt), 1)
            return cnt

    class Cfg:
        xres, yres, render_rounds, render_minutes = wl.xres, wl.yres, 2, None
        get_params = staticmethod(lambda sampler=0, flags=0: wl.params(sampler, flags))
    real_tl = rd.generate_task_list
    rd.generate_task_list = lambda *a, **k: O.generate_task_list(*a, **k)   # host-only twin (needs no HIP runtime)
    torch.cuda.current_stream = lambda dev=None: type("S", (), {"synchronize": lambda self: None})()
    drv = rd.RenderDriver(HostScene(), Cfg, wl.camera, rank=rank, world_size=world, device=torch.device("cpu"))
    if case == "timed":
        # skewed clocks: rank 0's clock allows exactly two rounds, every other rank's says the time was up before the first.
        # The decision is rank 0's (broadcast), so all ranks run two rounds and every reduce is matched.
        calls = [0]

        def clock():
            calls[0] += 1
            return 0.0 if (rank == 0 and calls[0] <= 3) else 1e9  # call 1 is render_frame's own t0
        if rank != 0:
            drv.clock = lambda: 1e9
            drv.render_frame(minutes=1.0)
        else:
            drv.clock = clock
            drv.render_frame(minutes=1.0)
    else:
        drv.render_frame()
    rd.generate_task_list = real_tl
    q.put((rank, drv.rounds_done, drv.total_ob.data.numpy().copy() if rank == 0 else None, drv.total_ob.count.numpy().copy() if rank == 0 else None))
    dist.barrier()
    dist.destroy_process_group()


def _run_ranks(world, case):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 7 + world * 13 + len(case)) % 2000
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, q, case)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rounds = {r: n for r, n, _, _ in got}
    data, count = next((d, c) for r, _, d, c in got if r == 0)
    return rounds, data, count


def _single_process(oracle, case, rounds=2):
    wl = _case_workload(case)
    osc = oracle.OracleScene(wl.builder.to_desc())
    acc = np.zeros((wl.yres, wl.xres, 3), np.float32); cnt = np.zeros((wl.yres, wl.xres), np.uint32)
    n = len(oracle.generate_task_list(wl.xres, wl.yres))
    for r in range(rounds):
        osc.render_round(wl.camera, wl.params(), oracle.generate_task_list(wl.xres, wl.yres, seedcount_base=r * n), acc, cnt)
    return acc, cnt


@pytest.mark.parametrize("world", [2, 4])
def test_tile_sharding_and_reduce(oracle, world):
    rounds, data, count = _run_ranks(world, "rounds")
    assert all(n == 2 for n in rounds.values())
    acc, cnt = _single_process(oracle, "rounds")  # two rounds over the whole tile list
    assert np.array_equal(count.view(np.uint32), cnt)
    assert np.array_equal(data, acc)      # disjoint tiles: the reduce is exact, the image does not depend on G


def test_timed_mode_is_a_collective_decision(oracle):
    """in timed mode every rank used to read its own clock; a rank that ran one more round left a reduce
    unmatched.  Rank 0 decides and broadcasts: with badly skewed clocks all ranks still run the same two rounds."""
    rounds, data, count = _run_ranks(2, "timed")
    assert rounds == {0: 2, 1: 2}
    acc, cnt = _single_process(oracle, "timed")
    assert np.array_equal(count.view(np.uint32), cnt) and np.array_equal(data, acc)


def test_reverse_splats_reduce_is_a_true_sum_world_size_4(oracle):
    """reverse > 0: light-tracing splats land on other ranks' pixels, so the per-round reduce is a true sum -- equal to the
    single-process image up to float re-association, and bit-identical run to run (fixed reduction order)."""
    rounds, data, count = _run_ranks(4, "reverse")
    _, data2, count2 = _run_ranks(4, "reverse")
    assert all(n == 2 for n in rounds.values())
    acc, cnt = _single_process(oracle, "reverse")
    assert np.array_equal(count.view(np.uint32), cnt)         # splats add radiance with count 0 (tracer.cpp:25)
    rel = np.linalg.norm(data - acc) / np.linalg.norm(acc)
    print(f"[gloo ws4 reverse] rel-L2 vs single process {rel:.3e}, bit-identical pixels {np.mean(data == acc):.4f}")
    assert rel < 1e-6
    assert np.array_equal(data, data2) and np.array_equal(count, count2)


def test_bench_launches_its_own_ranks_when_started_plainly():
    """`python bench.py --gpus N` is how the driver starts the scaling run: with no launcher around it (WORLD_SIZE unset)
    bench.py itself starts one child per GPU before touching the GPU, gives each what torch.distributed.run would have
    (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_*), passes the arguments on unchanged, prints rank 0's line alone on stdout and fails if
    any rank does.  RGK_BENCH_ECHO_ENV makes a rank report its environment and stop before it needs a GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["RGK_BENCH_ECHO_ENV"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(out) == 1 and out[0]["RANK"] == "0"          # ONE JSON line on stdout: rank 0's
    others = [json.loads(ln) for ln in r.stderr.splitlines() if ln.startswith("{")]
    ranks = sorted(int(o["RANK"]) for o in out + others)
    assert ranks == [0, 1, 2, 3]
    for o in out + others:
        assert o["WORLD_SIZE"] == "4" and o["LOCAL_RANK"] == o["RANK"] and o["MASTER_ADDR"] == "127.0.0.1"
        assert o["MASTER_PORT"] == out[0]["MASTER_PORT"] and int(o["MASTER_PORT"]) > 0
        assert o["argv"] == ["--gpus", "4", "--steps", "2", "--warmup", "1"]
    env["RGK_BENCH_ECHO_FAIL_RANK"] = "2"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "rank(s) failed" in r.stderr
    # under a launcher (WORLD_SIZE set) bench.py is a rank, not a launcher
    env.pop("RGK_BENCH_ECHO_FAIL_RANK")
    env.update({"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1"})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and json.loads(r.stdout)["RANK"] == "1"


# ----------------------------------------------------------------------- output path
def test_float_to_half_is_round_to_nearest_even(product_lib):
    rng = np.random.default_rng(5)
    bits = np.concatenate([rng.integers(0, 2 ** 32, 200000, dtype=np.uint64).astype(np.uint32),
                           np.array([0, 0x80000000, 0x33000000, 0x33000001, 0x387fc000, 0x38800000, 0x477fefff, 0x477ff000,
                                     0x7f800000, 0xff800000, 0x3f800000, 0x3f801000, 0x3f803000], dtype=np.uint32)])
    vals = bits.view(np.float32)
    vals = vals[np.isfinite(vals)]
    with np.errstate(over="ignore"):
        want = vals.astype(np.float16).view(np.uint16)   # IEEE round-to-nearest-even, what OpenEXR's half(float) yields
    got = np.array([product_lib.rgk_float_to_half(float(v)) for v in vals[:20000]] , dtype=np.uint16)
    assert np.array_equal(got, want[:20000])
    tail = np.array([product_lib.rgk_float_to_half(float(v)) for v in vals[-13:]], dtype=np.uint16)
    assert np.array_equal(tail, want[-13:])


def test_output_normalize_and_exr_round_trip(product_lib, tmp_path):
    """EXRTexture::Normalize + Write semantics (texture.cpp:349-400): auto scale = 1 / brightest channel of data/count,
    pixel = (data * scale) / count, zero where count is 0; the file holds those values as halves with A = 1."""
    import ctypes as C
    from rgk_amd.render_driver import read_exr
    rng = np.random.default_rng(6)
    W, H = 37, 23
    cnt = rng.integers(0, 5, (H, W)).astype(np.uint32) * 16
    acc = np.ascontiguousarray((rng.random((H, W, 3)) * 40.0 * cnt[..., None]).astype(np.float32))
    out = np.empty_like(acc)
    val = C.c_float()
    assert product_lib.rgk_output_normalize(acc.ctypes.data, cnt.ctypes.data, W, H, -1.0, out.ctypes.data, C.byref(val)) == 0
    with np.errstate(invalid="ignore", divide="ignore"):
        px = np.where(cnt[..., None] > 0, acc / cnt[..., None].astype(np.float32), 0).astype(np.float32)
    assert val.value == np.float32(1.0) / px.max()
    want = np.where(cnt[..., None] > 0, (acc * np.float32(val.value)) / np.maximum(cnt, 1)[..., None].astype(np.float32), 0).astype(np.float32)
    assert np.array_equal(out, want) and out.max() <= 1.0 + 1e-6
    assert product_lib.rgk_output_normalize(acc.ctypes.data, cnt.ctypes.data, W, H, 0.25, out.ctypes.data, None) == 0
    assert np.array_equal(out, np.where(cnt[..., None] > 0, (acc * np.float32(0.25)) / np.maximum(cnt, 1)[..., None].astype(np.float32), 0).astype(np.float32))
    path = str(tmp_path / "img.exr")
    assert product_lib.rgk_output_write_exr(path.encode(), W, H, out.ctypes.data) == 0
    img = read_exr(path)
    assert img.shape == (H, W, 4) and (img[..., 3] == 1.0).all()
    assert np.array_equal(img[..., :3], out.astype(np.float16).astype(np.float32))
    assert product_lib.rgk_output_write_exr(b"/nonexistent-dir/x.exr", W, H, out.ctypes.data) != 0
