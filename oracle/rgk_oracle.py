"""ORACLE -- test infrastructure only (ctypes wrapper over oracle/librgk_oracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
The product (rgk_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from rgk_amd import capi

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "librgk_oracle.so")
_p = C.POINTER
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [_p(capi.SceneDesc)]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_scene_get_info.argtypes = [C.c_void_p, _p(capi.SceneInfo)]
        L.orc_generate_task_list.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float,
                                             C.c_uint32, C.c_uint32, _p(capi.Tile), _p(C.c_uint32)]
        L.orc_render_round.argtypes = [C.c_void_p, _p(capi.Camera), _p(capi.Params), _p(capi.Tile),
                                       C.c_uint32, C.c_void_p, C.c_void_p, _p(capi.Counters), C.c_int]
        L.orc_trace_closest.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                        _p(capi.Counters)]
        L.orc_trace_visibility.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                           _p(capi.Counters)]
        L.orc_halton_raw.restype = C.c_float
        L.orc_halton_raw.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_sampler_eval.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_bxdf_value.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_bxdf_sample.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, _p(C.c_int)]
        L.orc_texture_sample.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, _p(C.c_float), _p(C.c_float)]
        L.orc_libm.argtypes = [C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_camera_init.argtypes = [_p(capi.Camera), capi.f3, capi.f3, capi.f3, C.c_float, C.c_float, C.c_int, C.c_int, C.c_float, C.c_float]
        L.orc_camera_ray.argtypes = [_p(capi.Camera), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_void_p]
        L.orc_test_intersection.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_stratified_sample.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def generate_task_list(xres, yres, seedstart=42, seedcount_base=0, tile_size=32, mid=None):
    L = lib()
    mid = mid or (xres / 2.0, yres / 2.0)
    n = C.c_uint32(0)
    L.orc_generate_task_list(tile_size, xres, yres, mid[0], mid[1], seedstart, seedcount_base, None, C.byref(n))
    tiles = (capi.Tile * n.value)()
    L.orc_generate_task_list(tile_size, xres, yres, mid[0], mid[1], seedstart, seedcount_base, tiles, C.byref(n))
    return tiles


class OracleScene:
    def __init__(self, desc):
        self.L = lib()
        self.h = self.L.orc_scene_create(C.byref(desc))

    def close(self):
        if self.h:
            self.L.orc_scene_destroy(self.h)
            self.h = None

    __del__ = close

    def info(self):
        i = capi.SceneInfo()
        self.L.orc_scene_get_info(self.h, C.byref(i))
        return i

    def render_round(self, camera, params, tiles, accum=None, count=None, n_threads=0):
        n = len(tiles)
        if accum is None:
            accum = np.zeros((params.yres, params.xres, 3), dtype=np.float32)
            count = np.zeros((params.yres, params.xres), dtype=np.uint32)
        cnt = capi.Counters()
        self.L.orc_render_round(self.h, C.byref(camera), C.byref(params), tiles, n,
                                accum.ctypes.data, count.ctypes.data, C.byref(cnt), n_threads)
        return accum, count, cnt

    def trace_closest(self, rays, ignore=None):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = len(rays)
        ig = None if ignore is None else np.ascontiguousarray(ignore, dtype=np.int32)
        hits = np.zeros(n, dtype=[("t", "f4"), ("tri", "i4"), ("a", "f4"), ("b", "f4"), ("c", "f4")])
        cnt = capi.Counters()
        self.L.orc_trace_closest(self.h, n, rays.ctypes.data, None if ig is None else ig.ctypes.data,
                                 hits.ctypes.data, C.byref(cnt))
        return hits, cnt

    def visibility(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 3)
        b = np.ascontiguousarray(b, dtype=np.float32).reshape(-1, 3)
        vis = np.zeros(len(a), dtype=np.uint8)
        cnt = capi.Counters()
        self.L.orc_trace_visibility(self.h, len(a), a.ctypes.data, b.ctypes.data, vis.ctypes.data, C.byref(cnt))
        return vis, cnt


def sampler_eval(seed, index, dim, is2d):
    seed = np.ascontiguousarray(seed, dtype=np.uint32)
    index = np.ascontiguousarray(index, dtype=np.uint32)
    dim = np.ascontiguousarray(dim, dtype=np.uint32)
    out = np.zeros((len(seed), 2), dtype=np.float32)
    lib().orc_sampler_eval(len(seed), seed.ctypes.data, index.ctypes.data, dim.ctypes.data, int(is2d), out.ctypes.data)
    return out
