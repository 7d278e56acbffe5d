"""The BASELINE.json configurations as data.

Each entry is the JSON the reference ships for that scene (scenes/<name>.json, values
transcribed, file:line cited) with BASELINE.json's overrides applied on top (SURVEY F6:
the BASELINE configs are overrides of the shipped files).  Geometry for the Sponza
family is the labelled procedural proxy unless RGK_ASSET_DIR supplies the real mesh.
"""
import os

from . import capi
from .config import Config, make_camera, make_params
from .proxy import proxy_texture, sponza_mesh_provider
from .scene import SceneBuilder

HERE = os.path.dirname(os.path.abspath(__file__))

# scenes/sponza.json:1-27
SPONZA_JSON = {
    "model-file": "sponza-fixed/sponza.obj", "output-file": "sponza.exr",
    "output-width": 1200, "output-height": 900, "recursion-max": 2,
    "camera": {"position": [-9.5, 1.5, -1.5], "lookat": [3.0, 3.0, -0.5], "focal": 1.6},
    "lights": [{"position": [-16.0, 100.0, -10.0], "color255": [255, 240, 200], "intensity": 20000.0}],
    "multisample": 40, "rounds": 20,
    "sky": {"color255": [145, 200, 235], "intensity": 0.3}, "bumpscale": 10,
}
# scenes/sponza4.json:1-31
SPONZA4_JSON = {
    "model-file": "sponza-fixed/sponza.obj", "output-file": "sponza4.exr",
    "output-width": 1200, "output-height": 900, "recursion-max": 4,
    "camera": {"position": [-13, 1.8, -1.2], "lookat": [3.0, 3.5, 3.2], "focal": 1.2},
    "lights": [{"position": [-5.0, 50.0, -20.0], "color255": [255, 240, 200], "intensity": 20000.0, "size": 1.0}],
    "multisample": 40, "rounds": 40,
    "sky": {"color255": [185, 220, 240], "intensity": 0.9},
    "bumpscale": 20, "clamp": 5.0, "russian": 0.6, "brdf": "ltc_ggx",
}

# scenes/dragon-sponza.json:1-38, with `"brdf": "ltc_ggx_diffuse"` injected into its material (the current
# loader requires the key) and a constant sky standing in for the absent cloudy1.hdr
DRAGON_SPONZA_JSON = {
    "materials": [{"name": "sp_00_pod", "brdf": "ltc_ggx_diffuse", "specular255": [209, 197, 181], "exponent": 800.0,
                   "diffuse-texture": "sponza-fixed/KAMEN.JPG", "bump-map": "sponza-fixed/KAMEN-bump.jpg"}],
    "scene": [{"file": "dragon-sponza/dragon.obj", "import-materials": True}],
    "output-file": "dragon-sponza.exr", "output-width": 1280, "output-height": 720,
    "camera": {"position": [-1.3, 2.2, -0.5], "lookat": [0.0, 1.9, 0.0], "focal": 1.0},
    "lights": [{"position": [-5.0, 35.0, -15.0], "color": [1.0, 0.85, 0.6], "intensity": 2000.0, "size": 0.7}],
    "multisample": 800, "rounds": 4,
    "sky": {"color255": [185, 220, 240], "intensity": 0.9},
    "bumpscale": 12, "clamp": 5.0, "russian": 0.7, "brdf": "ltc_ggx",
}

WORKLOADS = {
    # BASELINE.json configs[0]: plumbing on the CPU path
    "cornell-256": dict(kind="cornell", xres=256, yres=256, multisample=16, russian=0.74),
    # configs[1]
    "cornell-1024": dict(kind="cornell", xres=1024, yres=1024, multisample=256, russian=0.75),
    # configs[2]: the configuration BASELINE.json's metric is quoted on
    "sponza-1080p": dict(kind="json", root=SPONZA_JSON,
                         overrides={"output-width": 1920, "output-height": 1080, "multisample": 256, "rounds": 1}),
    # configs[3]: BDPT light sub-paths
    "dragon-sponza-1080p": dict(kind="json", root=DRAGON_SPONZA_JSON,
                                overrides={"output-width": 1920, "output-height": 1080, "multisample": 512, "reverse": 3, "rounds": 1}),
    # configs[4] (8 GPUs)
    "sponza4-2160p": dict(kind="json", root=SPONZA4_JSON,
                          overrides={"output-width": 3840, "output-height": 2160, "multisample": 1024, "rounds": 2}),
}


class Workload:
    def __init__(self, name, scale=1.0, spp=None, detail=1.0, dragon_level=7):
        """scale shrinks the resolution, spp overrides multisample (tests use small sizes)."""
        w = WORKLOADS[name]
        self.name = name
        if w["kind"] == "cornell":
            sb = SceneBuilder.load_npz(os.path.join(HERE, "data", "cornell_scene.npz"))
            ex = sb.extra
            self.xres, self.yres = max(1, int(w["xres"] * scale)), max(1, int(w["yres"] * scale))
            self.multisample = spp or w["multisample"]
            self.camera = make_camera(ex["camera"]["pos"], ex["camera"]["lookat"], ex["camera"]["up"],
                                      fov=ex["camera"]["fov"], xres=self.xres, yres=self.yres)
            self.builder = sb
            self.depth, self.clamp, self.russian, self.bumpscale = ex["depth"], ex["clamp"], w["russian"], ex["bumpscale"]
            self.reverse = 0
            self.rounds = 1
            self.geometry = "real"
        else:
            ov = dict(w["overrides"])
            ov["output-width"] = max(1, int(ov["output-width"] * scale))
            ov["output-height"] = max(1, int(ov["output-height"] * scale))
            if spp:
                ov["multisample"] = spp
            cfg = Config("<%s>" % name, ov, root=w["root"])
            self.cfg = cfg
            sb = SceneBuilder()
            sb.texture_fallback = proxy_texture
            self.builder = cfg.build_scene(builder=sb, asset_dir=os.environ.get("RGK_ASSET_DIR"),
                                           mesh_provider=sponza_mesh_provider(detail, dragon_level))
            self.xres, self.yres, self.multisample = cfg.xres, cfg.yres, cfg.multisample
            self.camera = cfg.get_camera()
            self.depth, self.clamp, self.russian = cfg.recursion_level, float(cfg.clamp), float(cfg.russian)
            self.bumpscale, self.reverse, self.rounds = float(cfg.bumpmap_scale), cfg.reverse, cfg.render_rounds
            self.geometry = self.builder.geometry_label

    def params(self, sampler=capi.SAMPLER_HALTON, flags=0):
        return make_params(self.xres, self.yres, self.multisample, self.depth, self.clamp, self.russian, self.bumpscale,
                           self.reverse, sampler, flags)


class SceneFixture:
    """A committed scene fixture (tests/golden/scene_<name>.npz, made by tools/make_fixtures.py from a scene
    the reference ships complete) with the render parameters of its config file."""

    def __init__(self, path, scale=1.0, spp=None, depth=None):
        sb = SceneBuilder.load_npz(path)
        ex = sb.extra
        self.name = ex.get("source", os.path.basename(path))
        self.builder = sb
        self.xres, self.yres = max(1, int(ex["xres"] * scale)), max(1, int(ex["yres"] * scale))
        self.multisample = spp or ex["multisample"]
        c = ex["camera"]
        self.camera = make_camera(c["pos"], c["lookat"], c["up"], fov=c.get("fov"), focal=c.get("focal"), xres=self.xres,
                                  yres=self.yres, focus_plane=c.get("focus_plane", 1.0), lens_size=c.get("lens_size", 0.0))
        self.depth = depth or ex["depth"]
        self.clamp, self.russian, self.bumpscale, self.reverse = ex["clamp"], ex["russian"], ex["bumpscale"], ex["reverse"]
        self.rounds = 1
        self.geometry = "real"

    params = Workload.params
