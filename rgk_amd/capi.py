"""ctypes mirror of include/rgk.h plus loaders for the product library.

The product library is rgk_amd/csrc/librgk_hip.so (hand-written HIP for gfx950).
There is NO CPU fallback: if the library is missing `load_product()` raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RGK_LIB") or os.path.join(HERE, "csrc", "librgk_hip.so")  # RGK_LIB: a tuning variant

# rgk_bxdf_kind
BXDF_DIFFUSE, BXDF_MIRROR, BXDF_DIELECTRIC, BXDF_TRANSPARENT, BXDF_MIX = 0, 1, 2, 3, 4
BXDF_LTC_BECKMANN, BXDF_LTC_GGX, BXDF_LTC_BECKMANN_DIFFUSE, BXDF_LTC_GGX_DIFFUSE = 5, 6, 7, 8
MAT_NO_RUSSIAN = 1
TEX_SOLID, TEX_RGB32F, TEX_RGB8 = 0, 1, 2
SKY_COLOR, SKY_ENVMAP = 0, 1
SAMPLER_HALTON, SAMPLER_STRATIFIED = 0, 1
FLAG_COUNT_TRAVERSAL, FLAG_TIME_KERNELS = 1, 2
BUILD_AUTO, BUILD_DEVICE, BUILD_KEEP_FLOAT_TEXTURES, BUILD_HOST_SAH = 0, 1, 2, 4

BRDF_IDS = {  # Material::LoadFromJson, reference src/bxdf/bxdf.cpp:63-84
    "diffusecosine": BXDF_DIFFUSE, "diffuse": BXDF_DIFFUSE, "mix": BXDF_MIX,
    "dielectric": BXDF_DIELECTRIC, "mirror": BXDF_MIRROR, "transparent": BXDF_TRANSPARENT,
    "ltc_beckmann": BXDF_LTC_BECKMANN, "ltc_ggx": BXDF_LTC_GGX,
    "ltc_beckmann_diffuse": BXDF_LTC_BECKMANN_DIFFUSE, "ltc_ggx_diffuse": BXDF_LTC_GGX_DIFFUSE,
}

f3 = C.c_float * 3


class Material(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("flags", C.c_uint32), ("emission", f3),
                ("roughness", C.c_float), ("ior", C.c_float), ("amount", C.c_float),
                ("tex_diffuse", C.c_int32), ("tex_color", C.c_int32), ("tex_bump", C.c_int32),
                ("mix_m1", C.c_int32), ("mix_m2", C.c_int32)]


class Texture(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32),
                ("color", f3), ("texels", C.POINTER(C.c_float)), ("texels8", C.POINTER(C.c_uint8)),
                ("lut", C.POINTER(C.c_float))]


class PointLight(C.Structure):
    _fields_ = [("pos", f3), ("color", f3), ("intensity", C.c_float), ("size", C.c_float)]


class SceneDesc(C.Structure):
    _fields_ = [("n_vertices", C.c_uint32),
                ("vertices", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)),
                ("tangents", C.POINTER(C.c_float)), ("texcoords", C.POINTER(C.c_float)),
                ("n_triangles", C.c_uint32),
                ("tri_indices", C.POINTER(C.c_uint32)), ("tri_material", C.POINTER(C.c_uint32)),
                ("n_materials", C.c_uint32), ("materials", C.POINTER(Material)),
                ("n_textures", C.c_uint32), ("textures", C.POINTER(Texture)),
                ("n_pointlights", C.c_uint32), ("pointlights", C.POINTER(PointLight)),
                ("n_areal_lights", C.c_uint32),
                ("areal_offsets", C.POINTER(C.c_uint32)), ("areal_tris", C.POINTER(C.c_uint32)),
                ("sky_mode", C.c_uint32), ("sky_color", f3), ("sky_intensity", C.c_float),
                ("sky_rotate", C.c_float), ("sky_texture", C.c_int32),
                ("ltc_ggx", C.POINTER(C.c_float)), ("ltc_beckmann", C.POINTER(C.c_float)), ("build_flags", C.c_uint32)]


class Camera(C.Structure):
    """rgk_camera: the public members of the reference's Camera (src/camera.hpp:27-41)."""
    _fields_ = [("origin", f3), ("direction", f3), ("cameraup", f3), ("cameraleft", f3),
                ("viewscreen", f3), ("viewscreen_x", f3), ("viewscreen_y", f3),
                ("lens_size", C.c_float), ("xsize", C.c_int32), ("ysize", C.c_int32)]


class Params(C.Structure):
    _fields_ = [("xres", C.c_uint32), ("yres", C.c_uint32), ("multisample", C.c_uint32),
                ("depth", C.c_uint32), ("clamp", C.c_float), ("russian", C.c_float),
                ("bumpmap_scale", C.c_float), ("force_fresnell", C.c_uint32),
                ("reverse", C.c_uint32), ("sampler", C.c_uint32), ("flags", C.c_uint32)]


class Tile(C.Structure):
    _fields_ = [("x0", C.c_uint32), ("x1", C.c_uint32), ("y0", C.c_uint32), ("y1", C.c_uint32),
                ("seed", C.c_uint32)]


KERNEL_IDS = ["trace_camera", "trace_closest", "shade_first", "shade", "shadow_first", "shadow", "shadow_jobs", "connect",
              "light_trace", "light_shade", "light_splat", "resolve", "other"]  # rgk_kernel_id
KERNEL_NAMES = {"trace_camera": "k_trace_camera", "trace_closest": "k_trace_closest", "shade_first": "k_shade<false, true", "shade": "k_shade<false, false",
                "shadow_first": "k_trace_shadow_first", "shadow": "k_trace_shadow<", "shadow_jobs": "k_trace_shadow_jobs", "connect": "k_connect",
                "resolve": "k_resolve"}  # prefixes of the kernel names rocprofv3 reports


class KernelStat(C.Structure):
    _fields_ = [("ms", C.c_double), ("launches", C.c_uint32), ("reserved", C.c_uint32), ("units", C.c_uint64),
                ("node_visits", C.c_uint64), ("tri_tests", C.c_uint64)]


class Counters(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("path_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("node_visits", C.c_uint64), ("tri_tests", C.c_uint64),
                ("shadow_node_visits", C.c_uint64), ("shadow_tri_tests", C.c_uint64),
                ("ms_trace", C.c_double), ("ms_shadow", C.c_double), ("ms_shade", C.c_double),
                ("ms_other", C.c_double), ("n_trace_launches", C.c_uint32),
                ("n_shadow_launches", C.c_uint32), ("n_shade_launches", C.c_uint32),
                ("reserved", C.c_uint32), ("kernel", KernelStat * len(KERNEL_IDS))]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k not in ("reserved", "kernel")}


class SceneInfo(C.Structure):
    _fields_ = [("epsilon", C.c_float), ("bbox_min", f3), ("bbox_max", f3),
                ("total_areal_power", C.c_float), ("total_point_power", C.c_float),
                ("n_nodes", C.c_uint32), ("node_bytes", C.c_uint32), ("tri_bytes", C.c_uint32),
                ("max_depth", C.c_uint32), ("n_leaf_refs", C.c_uint32),
                ("n_float_textures", C.c_uint32), ("n_palettized_textures", C.c_uint32)]


class Progress(C.Structure):
    _fields_ = [("stage", C.c_uint32), ("stages", C.c_uint32), ("rounds", C.c_uint32), ("busy", C.c_uint32),
                ("round_pixels", C.c_uint64), ("round_paths", C.c_uint64)]


class Hit(C.Structure):
    _fields_ = [("t", C.c_float), ("tri", C.c_int32), ("a", C.c_float), ("b", C.c_float),
                ("c", C.c_float)]


# every symbol include/rgk.h declares (tests check the .so exports all of them)
EXPORTS = ["rgk_last_error", "rgk_device_count", "rgk_scene_create", "rgk_scene_destroy",
           "rgk_scene_get_info", "rgk_scene_get_progress", "rgk_scene_set_tuning", "rgk_scene_refit", "rgk_generate_task_list", "rgk_camera_init", "rgk_render_round",
           "rgk_render_round_device", "rgk_trace_closest", "rgk_trace_visibility",
           "rgk_bxdf_value", "rgk_bxdf_sample", "rgk_texture_sample",
           "rgk_libm_eval", "rgk_sampler_eval", "rgk_output_normalize", "rgk_output_write_exr", "rgk_float_to_half",
           "rgk_accum_create", "rgk_accum_destroy", "rgk_accum_clear", "rgk_accum_rgb", "rgk_accum_count",
           "rgk_accum_download", "rgk_accum_upload", "rgk_accum_add", "rgk_accum_set_tag", "rgk_accum_save", "rgk_accum_load",
           "rgk_shard_tiles", "rgk_comm_get_unique_id", "rgk_comm_create", "rgk_comm_destroy", "rgk_accum_reduce"]

_p = C.POINTER


def _bind(lib):
    lib.rgk_last_error.restype = C.c_char_p
    lib.rgk_device_count.restype = C.c_int
    lib.rgk_scene_create.argtypes = [_p(SceneDesc), C.c_int, _p(C.c_void_p)]
    lib.rgk_scene_destroy.argtypes = [C.c_void_p]
    lib.rgk_scene_destroy.restype = None
    lib.rgk_scene_get_info.argtypes = [C.c_void_p, _p(SceneInfo)]
    lib.rgk_scene_get_progress.argtypes = [C.c_void_p, _p(Progress)]
    lib.rgk_scene_set_tuning.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
    lib.rgk_scene_refit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rgk_generate_task_list.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float,
                                           C.c_uint32, C.c_uint32, _p(Tile), _p(C.c_uint32)]
    lib.rgk_camera_init.argtypes = [_p(Camera), f3, f3, f3, C.c_float, C.c_float, C.c_int32, C.c_int32, C.c_float, C.c_float]
    lib.rgk_render_round.argtypes = [C.c_void_p, _p(Camera), _p(Params), _p(Tile), C.c_uint32,
                                     C.c_void_p, C.c_void_p, _p(Counters)]
    lib.rgk_render_round_device.argtypes = lib.rgk_render_round.argtypes
    lib.rgk_trace_closest.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                      _p(Counters)]
    lib.rgk_trace_visibility.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                         C.c_void_p, _p(Counters)]
    lib.rgk_bxdf_value.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rgk_bxdf_sample.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rgk_texture_sample.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rgk_libm_eval.argtypes = [C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rgk_sampler_eval.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                     C.c_void_p]
    lib.rgk_output_normalize.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_void_p, _p(C.c_float)]
    lib.rgk_output_write_exr.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.rgk_accum_create.argtypes = [C.c_uint32, C.c_uint32, C.c_int, _p(C.c_void_p)]
    lib.rgk_accum_destroy.argtypes = [C.c_void_p]
    lib.rgk_accum_destroy.restype = None
    lib.rgk_accum_clear.argtypes = [C.c_void_p]
    lib.rgk_accum_rgb.argtypes = [C.c_void_p]
    lib.rgk_accum_rgb.restype = C.c_void_p
    lib.rgk_accum_count.argtypes = [C.c_void_p]
    lib.rgk_accum_count.restype = C.c_void_p
    lib.rgk_accum_download.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rgk_accum_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rgk_accum_add.argtypes = [C.c_void_p, C.c_void_p]
    lib.rgk_accum_set_tag.argtypes = [C.c_void_p, C.c_uint64]
    lib.rgk_accum_save.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint32]
    lib.rgk_accum_load.argtypes = [C.c_void_p, C.c_char_p, _p(C.c_uint32), _p(C.c_uint32)]
    lib.rgk_shard_tiles.argtypes = [_p(Tile), C.c_uint32, C.c_int, C.c_int, _p(Tile), _p(C.c_uint32)]
    lib.rgk_comm_get_unique_id.argtypes = [C.c_void_p]
    lib.rgk_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _p(C.c_void_p)]
    lib.rgk_comm_destroy.argtypes = [C.c_void_p]
    lib.rgk_comm_destroy.restype = None
    lib.rgk_accum_reduce.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]
    lib.rgk_float_to_half.argtypes = [C.c_float]
    lib.rgk_float_to_half.restype = C.c_uint16
    return lib


_product = None


def load_product():
    """Load librgk_hip.so.  Fails loudly when the HIP extension has not been built."""
    global _product
    if _product is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        # One HIP runtime per process: when PyTorch-ROCm is going to share the device (accumulator
        # tensors, RCCL) its bundled libamdhip64 must be the one the loader resolves for us too,
        # so import it first; without torch the system ROCm runtime is used.
        if os.environ.get("RGK_NO_TORCH") != "1":
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        _product = _bind(C.CDLL(LIB_PATH))
    return _product


class RgkError(RuntimeError):
    pass


def check(lib, rc):
    if rc != 0:
        msg = lib.rgk_last_error()
        raise RgkError(f"rgk error {rc}: {msg.decode() if msg else ''}")
