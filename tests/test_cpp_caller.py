"""A compiled C++ host of the C ABI: tests/cpp/caller_main.cpp holds a Scene / Camera / Config /
EXRTexture shaped like the reference's (tests/cpp/reference_mirror.hpp: member declarations, cited), binds them with
the code INTEGRATION.md shows (tests/cpp/rgk_binding.inc) and renders.  Its accumulator must equal the ctypes path's
bit for bit -- same library, same inputs, another language on top."""
import os
import struct
import subprocess

import numpy as np
import pytest

from rgk_amd import capi
from rgk_amd.workloads import SceneFixture, Workload

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
CSRC = os.path.join(ROOT, "rgk_amd", "csrc")


def build_caller(tmp_path):
    capi.load_product()  # fails loudly if the product library has not been built
    exe = str(tmp_path / "rgk_caller")
    cmd = ["g++", "-std=c++17", "-Wall", "-Werror", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(CPP, "caller_main.cpp"),
           "-o", exe, "-L" + CSRC, "-lrgk_hip", "-Wl,-rpath," + CSRC, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


def dump_scene(path, sb, cam, prm, rounds):
    """The scene as flat arrays + the Camera constructor arguments + the Config scalars, for caller_main.cpp."""
    sb.finalize()
    f32, u32 = np.float32, np.uint32
    out = [struct.pack("<I", 0x524B4753)]
    nv = len(sb.V)
    out += [struct.pack("<I", nv), sb.V.astype(f32).tobytes(), sb.N.astype(f32).tobytes(), sb.T.astype(f32).tobytes(),
            struct.pack("<I", 1), sb.UV.astype(f32).tobytes()]
    out += [struct.pack("<I", len(sb.F)), sb.F.astype(u32).tobytes(), sb.FM.astype(u32).tobytes()]
    out.append(struct.pack("<I", len(sb.materials)))
    for m in sb.materials:
        out.append(struct.pack("<II3ffff5i", m["kind"], m["flags"], *m["emission"], m["roughness"], m["ior"], m["amount"],
                               m["tex_diffuse"], m["tex_color"], m["tex_bump"], m["mix_m1"], m["mix_m2"]))
    out.append(struct.pack("<I", len(sb.textures)))
    for t in sb.textures:
        if t["kind"] == capi.TEX_SOLID:
            out.append(struct.pack("<III3f", 0, 0, 0, *t["color"]))
        else:  # the floats the reference's FileTexture would hold: 8-bit sources through their byte -> float table
            data = t["data"] if t["kind"] == capi.TEX_RGB32F else t["lut"][t["data"]]
            h, w = data.shape[:2]
            out += [struct.pack("<III3f", 1, w, h, 0.0, 0.0, 0.0), np.ascontiguousarray(data, dtype=f32).tobytes()]
    out.append(struct.pack("<I", len(sb.pointlights)))
    for l in sb.pointlights:
        out.append(struct.pack("<8f", *l["pos"], *l["color"], l["intensity"], l["size"]))
    offs = np.cumsum([0] + [len(a) for a in sb.areal]).astype(u32)
    out += [struct.pack("<I", len(sb.areal)), offs.tobytes(), np.array([t for a in sb.areal for t in a], dtype=u32).tobytes()]
    s = sb.sky
    out.append(struct.pack("<I5fi", s["mode"], *s["color"], s["intensity"], s["rotate"], s["tex"]))
    for name in ("ggx", "beckmann"):
        out.append(np.fromfile(os.path.join(ROOT, "rgk_amd", "data", f"ltc_{name}.f32"), dtype=f32).tobytes())
    c = cam.ctor
    out.append(struct.pack("<11f2i2f", *c["pos"], *c["lookat"], *c["up"], c["yview"], c["xview"], c["xsize"], c["ysize"], c["focus_plane"], c["lens_size"]))
    out.append(struct.pack("<4I3f2I", prm.xres, prm.yres, prm.multisample, prm.depth, prm.clamp, prm.russian, prm.bumpmap_scale, prm.reverse, rounds))
    open(path, "wb").write(b"".join(out))


def read_result(path, xres, yres):
    b = open(path, "rb").read()
    (rays,) = struct.unpack_from("<Q", b, 0)
    acc = np.frombuffer(b, dtype=np.float32, count=xres * yres * 3, offset=8).reshape(yres, xres, 3)
    cnt = np.frombuffer(b, dtype=np.uint32, count=xres * yres, offset=8 + xres * yres * 12).reshape(yres, xres)
    return acc, cnt, rays


def test_binding_compiles_against_the_reference_shaped_types(tmp_path):
    """The INTEGRATION.md code (rgk_binding.inc) compiles -Wall -Werror against include/rgk.h and the mirrored member
    declarations, and links against the product library: every identifier it uses exists."""
    exe = build_caller(tmp_path)
    assert os.path.exists(exe)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


def _workloads():
    yield "cornell", Workload("cornell-256", scale=0.25, spp=8)
    yield "rubiks-bump", SceneFixture(os.path.join(ROOT, "tests", "golden", "scene_rubiks-bump.npz"), scale=0.1, spp=4, depth=4)
    yield "spheres", SceneFixture(os.path.join(ROOT, "tests", "golden", "scene_cornell-box-spheres.npz"), scale=0.1, spp=4, depth=6)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["host-accum", "device-accum"])
def test_cpp_caller_equals_ctypes_path(tmp_path, mode):
    from rgk_amd import render_driver as rd
    exe = build_caller(tmp_path)
    for name, wl in _workloads():
        prm, rounds = wl.params(), 2
        scene_bin, out_bin = str(tmp_path / f"{name}.bin"), str(tmp_path / f"{name}.out")
        dump_scene(scene_bin, wl.builder, wl.camera, prm, rounds)
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        r = subprocess.run([exe, scene_bin, out_bin] + (["--device-accum"] if mode == "device-accum" else []), capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr + r.stdout
        acc, cnt, rays = read_result(out_bin, prm.xres, prm.yres)
        sc = rd.Scene(wl.builder.to_desc(), device=0)
        ref_acc = np.zeros((prm.yres, prm.xres, 3), np.float32)
        ref_cnt = np.zeros((prm.yres, prm.xres), np.uint32)
        ref_rays, seedcount = 0, 0
        for _ in range(rounds):
            tiles = rd.generate_task_list(prm.xres, prm.yres, rd.SEEDSTART, seedcount)
            seedcount += len(tiles)
            ref_acc, ref_cnt, c = sc.render_round(wl.camera, prm, tiles, ref_acc, ref_cnt)
            ref_rays += c.path_rays
        sc.close()
        assert np.array_equal(cnt, ref_cnt), name
        assert rays == ref_rays, name
        assert np.array_equal(acc.view(np.uint32), ref_acc.view(np.uint32)), f"{name}: C++ caller differs from the ctypes path"
        print(f"[cpp-caller {mode}] {name}: {prm.xres}x{prm.yres}x{prm.multisample} x {rounds} rounds, {rays} path rays, bit-identical")


@pytest.mark.gpu
def test_cpp_caller_multi_rank_rounds_sum_once(tmp_path):
    """the multi-GPU pattern INTEGRATION.md section 3 documents (per-round accumulator -> in-place reduce -> the root adds
    the round to its total) must count every rank's every round exactly once.  The earlier form reduced an accumulator that
    kept growing, which adds the other ranks' round-1 data again in round 2 -- invisible with one rank.  Here the caller plays
    2 and 3 ranks on one GPU over 3 rounds, with the reduce replaced by what it computes (root += the others, the others keep
    theirs): bit-identical to the single-process frame, because tiles are disjoint and the per-pixel additions are the same."""
    from rgk_amd import render_driver as rd
    exe = build_caller(tmp_path)
    wl = Workload("cornell-256", scale=0.25, spp=4)
    prm, rounds = wl.params(), 3
    scene_bin = str(tmp_path / "c.bin")
    dump_scene(scene_bin, wl.builder, wl.camera, prm, rounds)
    sc = rd.Scene(wl.builder.to_desc(), device=0)
    ref_acc = np.zeros((prm.yres, prm.xres, 3), np.float32)
    ref_cnt = np.zeros((prm.yres, prm.xres), np.uint32)
    ref_rays, seedcount = 0, 0
    for _ in range(rounds):
        tiles = rd.generate_task_list(prm.xres, prm.yres, rd.SEEDSTART, seedcount)
        seedcount += len(tiles)
        ref_acc, ref_cnt, c = sc.render_round(wl.camera, prm, tiles, ref_acc, ref_cnt)
        ref_rays += c.path_rays
    sc.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for world in (2, 3):
        out_bin = str(tmp_path / f"c{world}.out")
        r = subprocess.run([exe, scene_bin, out_bin, "--emulate-world", str(world)], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr + r.stdout
        acc, cnt, rays = read_result(out_bin, prm.xres, prm.yres)
        assert np.array_equal(cnt, ref_cnt), f"world {world}: sample counts differ (a round counted twice?)"
        assert rays == ref_rays
        assert np.array_equal(acc.view(np.uint32), ref_acc.view(np.uint32)), f"world {world}: frame differs from the single-process one"
