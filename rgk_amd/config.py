"""JSON config front-end: the host-side mirror of the reference's ConfigJSON.

Follows reference src/config.cpp:260-326 (CreateFromFile: scalar fields and their
defaults), :328-370 (GetCamera), :372-387 (InstallLights), :389-413 (InstallSky),
:415-543 (InstallScene), :545-558 (InstallMaterials) and the typed getters of
src/jsonutils.cpp:21-120 (incl. the `key255` variants that divide by 255).
JSON is parsed with `//` and `/* */` comments allowed (jsoncpp Reader behaviour).
"""
import ctypes as C
import json
import math
import os
import re

import numpy as np

from . import capi
from .scene import SceneBuilder

f32 = np.float32


class ConfigFileException(RuntimeError):
    """src/config.hpp:16-18"""


def strip_json_comments(text):
    out = []
    i, n = 0, len(text)
    in_str = False
    while i < n:
        c = text[i]
        if in_str:
            out.append(c)
            if c == "\\" and i + 1 < n:
                out.append(text[i + 1])
                i += 1
            elif c == '"':
                in_str = False
        elif c == '"':
            in_str = True
            out.append(c)
        elif c == "/" and i + 1 < n and text[i + 1] == "/":
            while i < n and text[i] != "\n":
                i += 1
            continue
        elif c == "/" and i + 1 < n and text[i + 1] == "*":
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            continue
        else:
            out.append(c)
        i += 1
    return "".join(out)


class _JsonReader:
    """The grammar of the reader the reference vendors and calls (external/jsoncpp.cpp `Json::Reader`, src/config.cpp:266-272:
    comments allowed, root not strict): standard JSON plus // and /* */ comments between tokens and jsoncpp's number token
    `-?digits[.digits][(e|E)[+-]digits]` -- which takes leading zeros (`000.0` in scenes/conference.json) that Python's json
    module rejects.  Integers without '.', 'e', 'E' stay integers; everything else is a double.  Trailing content after the
    root value is ignored, as jsoncpp's old Reader ignores it."""

    def __init__(self, text):
        self.t, self.i, self.n = text, 0, len(text)

    def fail(self, msg):
        line = self.t.count("\n", 0, self.i) + 1
        raise ConfigFileException(f"Failed to parse JSON contents: line {line}: {msg}")

    def skip(self):
        t, n = self.t, self.n
        while self.i < n:
            c = t[self.i]
            if c in " \t\r\n":
                self.i += 1
            elif c == "/" and self.i + 1 < n and t[self.i + 1] == "/":
                j = t.find("\n", self.i)
                self.i = n if j < 0 else j + 1
            elif c == "/" and self.i + 1 < n and t[self.i + 1] == "*":
                j = t.find("*/", self.i + 2)
                if j < 0:
                    self.fail("unterminated comment")
                self.i = j + 2
            else:
                break

    def value(self):
        self.skip()
        if self.i >= self.n:
            self.fail("unexpected end of input")
        c = self.t[self.i]
        if c == "{":
            return self.obj()
        if c == "[":
            return self.arr()
        if c == '"':
            return self.string()
        if c == "-" or c.isdigit():
            return self.number()
        for word, val in (("true", True), ("false", False), ("null", None)):
            if self.t.startswith(word, self.i):
                self.i += len(word)
                return val
        self.fail("Syntax error: value, object or array expected.")

    def number(self):
        t, n, j = self.t, self.n, self.i
        if t[j] == "-":
            j += 1
        while j < n and t[j].isdigit():
            j += 1
        if j < n and t[j] == ".":
            j += 1
            while j < n and t[j].isdigit():
                j += 1
        if j < n and t[j] in "eE":
            j += 1
            if j < n and t[j] in "+-":
                j += 1
            while j < n and t[j].isdigit():
                j += 1
        tok = t[self.i:j]
        self.i = j
        try:
            if any(ch in tok for ch in ".eE"):
                return float(tok)
            v = int(tok)
            return v if -2 ** 63 <= v <= 2 ** 64 - 1 else float(tok)
        except ValueError:
            self.fail(f"'{tok}' is not a number.")

    def string(self):
        t, n = self.t, self.n
        self.i += 1
        out = []
        esc = {'"': '"', "\\": "\\", "/": "/", "b": "\b", "f": "\f", "n": "\n", "r": "\r", "t": "\t"}
        while True:
            if self.i >= n:
                self.fail("unterminated string")
            c = t[self.i]
            self.i += 1
            if c == '"':
                return "".join(out)
            if c != "\\":
                out.append(c)
                continue
            if self.i >= n:
                self.fail("Empty escape sequence in string")
            e = t[self.i]
            self.i += 1
            if e in esc:
                out.append(esc[e])
            elif e == "u":
                cp = self.hex4()
                if 0xD800 <= cp <= 0xDBFF and t.startswith("\\u", self.i):
                    self.i += 2
                    lo = self.hex4()
                    cp = 0x10000 + ((cp & 0x3FF) << 10) + (lo & 0x3FF)
                out.append(chr(cp))
            else:
                self.fail("Bad escape sequence in string")

    def hex4(self):
        h = self.t[self.i:self.i + 4]
        if len(h) != 4 or any(ch not in "0123456789abcdefABCDEF" for ch in h):
            self.fail("Bad unicode escape sequence in string")
        self.i += 4
        return int(h, 16)

    def arr(self):
        self.i += 1
        out = []
        self.skip()
        if self.i < self.n and self.t[self.i] == "]":
            self.i += 1
            return out
        while True:
            out.append(self.value())
            self.skip()
            if self.i >= self.n:
                self.fail("Missing ',' or ']' in array declaration")
            c = self.t[self.i]
            self.i += 1
            if c == "]":
                return out
            if c != ",":
                self.fail("Missing ',' or ']' in array declaration")

    def obj(self):
        self.i += 1
        out = {}
        while True:
            self.skip()
            if self.i >= self.n:
                self.fail("Missing '}' or object member name")
            c = self.t[self.i]
            if c == "}" and not out:
                self.i += 1
                return out
            if c != '"':
                self.fail("Missing '}' or object member name")
            k = self.string()
            self.skip()
            if self.i >= self.n or self.t[self.i] != ":":
                self.fail("Missing ':' after object member name")
            self.i += 1
            out[k] = self.value()
            self.skip()
            if self.i >= self.n:
                self.fail("Missing ',' or '}' in object declaration")
            c = self.t[self.i]
            self.i += 1
            if c == "}":
                return out
            if c != ",":
                self.fail("Missing ',' or '}' in object declaration")


def parse_json(text):
    """Parse a config file's text the way the reference's vendored jsoncpp reads it (see _JsonReader)."""
    return _JsonReader(text).value()


def _vec3(v, what):
    if isinstance(v, (list, tuple)):
        if len(v) != 3 or not all(isinstance(x, (int, float)) and not isinstance(x, bool) for x in v):
            raise ConfigFileException(f'value "{what}" must be an array of 3 numbers or a single number.')
        return np.array(v, dtype=f32)
    if isinstance(v, (int, float)) and not isinstance(v, bool):
        return np.array([v, v, v], dtype=f32)
    raise ConfigFileException(f'value "{what}" must be an array of 3 numbers or a single number.')


class Node:
    """A JSON object with the reference's typed getters and used-key tracking."""

    def __init__(self, d, name):
        self.d, self.name, self.used = d, name, set()

    def has(self, k):
        return k in self.d

    def _num(self, k, kind):
        v = self.d[k]
        if isinstance(v, bool) or not isinstance(v, (int, float)):
            raise ConfigFileException(f'{kind} value "{k}" in {self.name} must be a number.')
        self.used.add(k)
        return v

    def req_str(self, k):
        if k not in self.d:
            raise ConfigFileException(f'Required value "{k}" is missing from {self.name}.')
        if not isinstance(self.d[k], str):
            raise ConfigFileException(f'Required value "{k}" in {self.name} must be a string.')
        self.used.add(k)
        return self.d[k]

    def req_int(self, k):
        if k not in self.d:
            raise ConfigFileException(f'Required value "{k}" is missing from {self.name}.')
        return int(self._num(k, "Required"))

    def req_float(self, k):
        if k not in self.d:
            raise ConfigFileException(f'Required value "{k}" is missing from {self.name}.')
        return f32(self._num(k, "Required"))

    def req_vec3(self, k):
        if k not in self.d:
            raise ConfigFileException(f'Required value "{k}" is missing from {self.name}.')
        self.used.add(k)
        return _vec3(self.d[k], k)

    def req_vec3_255(self, k):
        if k in self.d:
            return self.req_vec3(k)
        if k + "255" in self.d:
            self.used.add(k + "255")
            return _vec3(self.d[k + "255"], k + "255") / f32(255.0)
        raise ConfigFileException(f'Required value "{k}" is missing from {self.name}.')

    def opt_str(self, k, d):
        return self.req_str(k) if k in self.d else d

    def opt_int(self, k, d):
        return self.req_int(k) if k in self.d else d

    def opt_float(self, k, d):
        return self.req_float(k) if k in self.d else f32(d)

    def opt_bool(self, k, d):
        if k not in self.d:
            return d
        if not isinstance(self.d[k], bool):
            raise ConfigFileException(f'Optional value "{k}" in {self.name} must be a bool.')
        self.used.add(k)
        return self.d[k]

    def opt_vec3(self, k, d):
        return self.req_vec3(k) if k in self.d else np.array(d, dtype=f32)

    def opt_vec3_255(self, k, d):
        if k in self.d or (k + "255") in self.d:
            return self.req_vec3_255(k)
        return np.array(d, dtype=f32)

    def unused(self):
        return [k for k in self.d if k not in self.used]


def fov2xview(fov):
    """src/config.cpp:328-330"""
    return f32(2.0) * f32(math.tan(float(f32(fov) * f32(0.0174533) / f32(2.0))))


class Config:
    """Scalar fields of reference `Config` (src/config.hpp:27-44) + the parsed root."""

    def __init__(self, path, overrides=None, root=None):
        self.config_file_path = path
        if root is None:
            try:
                text = open(path).read()
            except OSError:
                raise ConfigFileException("Failed to open file: " + path)
            root = parse_json(text)
        else:
            root = json.loads(json.dumps(root))
        if overrides:
            root.update(overrides)  # BASELINE configs are overrides of the shipped files
        self.root = Node(root, "the config file")
        r = self.root
        self.output_file = r.req_str("output-file")
        self.xres = r.req_int("output-width")
        self.yres = r.req_int("output-height")
        if r.has("rounds") and r.has("render-time"):
            raise ConfigFileException('The config file may not contain both "rounds" and "render-time" keys simultaneously.')
        self.render_minutes = None
        self.render_rounds = 1
        if r.has("rounds"):
            self.render_rounds = r.req_int("rounds")
        elif r.has("render-time"):
            self.render_minutes = r.req_int("render-time")
        self.recursion_level = r.opt_int("recursion-max", 40)
        self.multisample = r.opt_int("multisample", 1)
        self.clamp = r.opt_float("clamp", 10000000.0)
        self.bumpmap_scale = r.opt_float("bumpscale", 1.0)
        self.russian = r.opt_float("russian", 0.74)
        self.reverse = r.opt_int("reverse", 0)
        self.force_fresnell = r.opt_bool("force-fresnell", False)
        self.output_scale = -1.0
        if r.has("output-scale"):
            v = r.d["output-scale"]
            r.used.add("output-scale")
            if isinstance(v, str):
                if v != "auto":
                    raise ConfigFileException('The value of "output-scale" must either be a number, or "auto".')
            elif isinstance(v, (int, float)) and not isinstance(v, bool):
                self.output_scale = float(v)
            else:
                raise ConfigFileException('The value of "output-scale" must either be a number, or "auto".')
        self.thinglass = []
        if r.has("thinglass"):
            r.used.add("thinglass")
            t = r.d["thinglass"]
            if not isinstance(t, list) or not all(isinstance(x, str) for x in t):
                raise ConfigFileException('Value "thinglass" must be an array of strings')
            self.thinglass = list(t)

    # ---- src/config.cpp:332-370
    def get_camera(self, rotation=0.0):
        r = self.root
        if not r.has("camera"):
            raise ConfigFileException('Value "camera" is missing.')
        r.used.add("camera")
        if not isinstance(r.d["camera"], dict):
            raise ConfigFileException('Value "camera" is not a dictionary.')
        cam = Node(r.d["camera"], "camera configuration")
        pos = cam.req_vec3("position")
        lookat = cam.req_vec3("lookat")
        up = cam.opt_vec3("upvector", (0.0, 1.0, 0.0))
        if cam.has("focal"):
            yview = cam.req_float("focal")
            xview = f32(yview * f32(self.xres)) / f32(self.yres)
        elif cam.has("fov"):
            xview = fov2xview(cam.req_float("fov"))
            yview = f32(xview * f32(self.yres)) / f32(self.xres)
        else:
            raise ConfigFileException('Camera must either have a "fov" or "focal" key defined')
        focus_plane = cam.opt_float("focus-plane", 1.0)
        lens_size = cam.opt_float("lens-size", 0.0)
        if rotation != 0.0:
            from .scene import glm_rotate
            p = lookat - pos
            R = glm_rotate(f32(rotation) * f32(2.0) * f32(math.pi), up)[:3, :3]
            p = (R @ p).astype(f32)
            pos = lookat - p
        return camera_from_args(pos, lookat, up, yview, xview, self.xres, self.yres, focus_plane, lens_size)

    def get_params(self, sampler=capi.SAMPLER_HALTON, flags=0):
        """The PathTracer constructor arguments RenderRound passes (render_driver.cpp:164-173)."""
        p = capi.Params()
        p.xres, p.yres = self.xres, self.yres
        p.multisample = self.multisample
        p.depth = self.recursion_level
        p.clamp = float(self.clamp)
        p.russian = float(self.russian)
        p.bumpmap_scale = float(self.bumpmap_scale)
        p.force_fresnell = int(self.force_fresnell)
        p.reverse = self.reverse
        p.sampler = sampler
        p.flags = flags
        return p

    # ---- main.cpp:208-217 order: materials, scene, lights, sky, (thinglass), Commit
    def build_scene(self, asset_dir=None, mesh_provider=None, builder=None):
        sb = builder or SceneBuilder()
        self.install_materials(sb)
        self.install_scene(sb, asset_dir, mesh_provider)
        self.install_lights(sb)
        self.install_sky(sb)
        return sb

    def install_materials(self, sb):
        r = self.root
        if not r.has("materials"):
            return
        r.used.add("materials")
        mats = r.d["materials"]
        if not isinstance(mats, list):
            raise ConfigFileException('The value of "materials" key must be an array of material data')
        configdir = os.path.dirname(os.path.abspath(self.config_file_path))
        for i, m in enumerate(mats):
            sb.load_material_from_json(Node(m, f"material {i} configuration"), configdir, override=True)

    def install_lights(self, sb):
        r = self.root
        if not r.has("lights"):
            return
        r.used.add("lights")
        if not isinstance(r.d["lights"], list):
            raise ConfigFileException('Value "lights" must be an array.')
        for i, l in enumerate(r.d["lights"]):
            n = Node(l, f"light {i} configuration")
            sb.add_point_light(n.req_vec3("position"), n.opt_vec3_255("color", (1.0, 1.0, 1.0)),
                               n.req_float("intensity"), n.opt_float("size", 0.0))

    def install_sky(self, sb):
        r = self.root
        if not r.has("sky"):
            sb.set_skybox_color((0.0, 0.0, 0.0), 1.0)
            return
        r.used.add("sky")
        if not isinstance(r.d["sky"], dict):
            raise ConfigFileException('Value "sky" must be a dictionary.')
        sky = Node(r.d["sky"], "sky configuration")
        if sky.has("envmap"):
            configdir = os.path.dirname(os.path.abspath(self.config_file_path))
            path = sky.req_str("envmap")
            sb.set_skybox_envmap(os.path.join(configdir, path), sky.opt_float("intensity", 1.0),
                                 sky.opt_float("rotate", 0.0))
        elif sky.has("color") or sky.has("color255"):
            sb.set_skybox_color(sky.req_vec3_255("color"), sky.opt_float("intensity", 1.0))
        else:
            raise ConfigFileException('Sky configuration must either contain an "envmap" key or a "color" key')

    def install_scene(self, sb, asset_dir=None, mesh_provider=None):
        r = self.root
        configdir = os.path.dirname(os.path.abspath(self.config_file_path))
        if r.has("model-file") and r.has("scene"):
            raise ConfigFileException('The input file may not contain both "model-file" key and "scene" key, maximum one of these is allowed.')
        if r.has("model-file"):
            rel = r.req_str("model-file")
            brdf = r.opt_str("brdf", "ltc_ggx")  # read, then ignored by LoadAiSceneMaterials
            modelfile = _resolve(configdir, rel, asset_dir)
            if modelfile is None:
                if mesh_provider is None:
                    raise ConfigFileException(f'Unable to open model file "{os.path.join(configdir, rel)}"')
                mesh_provider(sb, rel)  # labelled proxy geometry
            else:
                sb.load_obj(modelfile, np.eye(4, dtype=f32), import_materials=True, override_materials=False)
            del brdf
        elif r.has("scene"):
            r.used.add("scene")
            if not isinstance(r.d["scene"], list):
                raise ConfigFileException('The value of "scene" key must be an array of objects')
            for i, o in enumerate(r.d["scene"]):
                obj = Node(o, f"scene object {i} configuration")
                if obj.has("file") and obj.has("primitive"):
                    raise ConfigFileException(f'Both "file" and "primitive" keys found in {obj.name}, only one can be present at a time.')
                if obj.has("file"):
                    rel = obj.req_str("file")
                    import_materials = obj.opt_bool("import-materials", False)
                    override_materials = obj.opt_bool("override-materials", False)
                    forced = obj.opt_str("material", "")
                    smooth = obj.opt_bool("smooth-normals", False)
                    obj.opt_str("brdf", r.opt_str("brdf", "ltc_ggx"))
                    T = sb.object_transform(np.eye(4, dtype=f32), obj)
                    modelfile = _resolve(configdir, rel, asset_dir)
                    if modelfile is None:
                        if mesh_provider is None:
                            raise ConfigFileException(f'Unable to find model file "{os.path.join(configdir, rel)}"')
                        mesh_provider(sb, rel, T, forced)
                    else:
                        sb.load_obj(modelfile, T, import_materials=import_materials,
                                    override_materials=override_materials, force_mat=forced,
                                    smooth_normals=smooth)
                elif obj.has("primitive"):
                    sb.add_primitive_from_json(obj)
                else:
                    raise ConfigFileException(f'Missing mesh data in {obj.name}, it must either contain a "file" key, or "primitive" key.')
        else:
            raise ConfigFileException('The input file contains neither "scene" nor "model-file" key.')

    def perform_post_check(self):
        """src/config.cpp:560-570: keys present but never read."""
        return self.root.unused()


def _stoi(tok, what):
    """std::stoi: optional sign + the leading decimal digits of the token (src/config.cpp uses it unguarded; a token without
    digits makes the reference die with an uncaught std::invalid_argument -- here a ConfigFileException)."""
    m = re.match(r"\s*[+-]?\d+", tok)
    if not m:
        raise ConfigFileException(f"Invalid {what}: '{tok}' is not an integer.")
    return int(m.group(0))


def _stof(tok, what):
    m = re.match(r"\s*[+-]?(\d+\.?\d*([eE][+-]?\d+)?|\.\d+([eE][+-]?\d+)?|inf(inity)?|nan)", tok, re.I)
    if not m:
        raise ConfigFileException(f"Invalid {what}: '{tok}' is not a number.")
    return f32(float(m.group(0)))


class ConfigRTC:
    """The reference's older line-based config format, ConfigRTC (src/config.cpp:27-258): nine fixed lines -- comment, model
    file, output file, recursion level, `xres yres`, camera position, look-at, up vector, yview -- then option lines
    (`L x y z r g b intensity [size]`, `multisample n`, `sky r g b [brightness]`, `lens`, `focus`, `bumpscale`, `clamp`,
    `russian`, `rounds`, `reverse`, `brdf`, `thinglass`, `force_fresnell`; `#` starts a comment; unknown options warn).
    Same interface as Config (get_camera / get_params / build_scene).  Fields keep the base class defaults of
    src/config.hpp:31-42 where the file sets nothing: bumpscale 10, clamp 100000, russian -1 (no roulette), sky brightness 2."""

    BRDFS = {"cooktorr": "cooktorr", "phong": "phong", "phong2": "phong2", "phongenergy": "phongenergy", "diffuse": "diffusecosine",
             "diffuseuniform": "diffuseuniform", "ltc_beckmann": "ltc_beckmann", "ltc_ggx": "ltc_ggx"}

    def __init__(self, path, text=None):
        self.config_file_path = path
        if text is None:
            try:
                text = open(path).read()
            except OSError:
                raise ConfigFileException("Failed to open config file ` " + path + " `.")
        lines = text.split("\n")
        if text.endswith("\n"):
            lines = lines[:-1]           # std::getline: the final newline does not start another line
        it = iter(lines)

        def next_line():
            # NEXT_LINE(): getline + trim, then "ends prematurely" unless the stream is still good -- it is not once a getline has
            # run into the end of the file, i.e. when there was no line left or the line just read had no newline behind it
            nonlocal pos
            try:
                ln = next(it)
            except StopIteration:
                raise ConfigFileException("Config file ends prematurely.")
            pos += 1
            if pos >= len(lines) and not text.endswith("\n"):
                raise ConfigFileException("Config file ends prematurely.")
            return ln.strip()
        pos = 0
        split = lambda ln: [t for t in ln.split(" ") if t != ""]
        try:
            first = next(it)
            pos += 1
        except StopIteration:
            first = ""
        self.comment = first.strip()
        self.model_file = next_line()
        self.output_file = next_line()
        self.recursion_level = _stoi(next_line(), "recursion level") & 0xFFFFFFFF
        vs = split(next_line())
        if len(vs) != 2:
            raise ConfigFileException("Invalid resolution format.")
        self.xres, self.yres = _stoi(vs[0], "resolution"), _stoi(vs[1], "resolution")
        if self.xres == 0 or self.yres == 0:
            raise ConfigFileException("Invalid output image resolution.")

        def vec(what):
            v = split(next_line())
            if len(v) != 3:
                raise ConfigFileException(f"Invalid {what} format.")
            return np.array([_stof(x, what) for x in v], dtype=f32)
        self.camera_position, self.camera_lookat, self.camera_upvector = vec("VP"), vec("LA"), vec("UP")
        self.yview = _stof(next_line(), "yview")
        if self.yview <= 0.0 or self.yview >= 100.0:
            raise ConfigFileException("Invalid yview value.")
        # Config base-class defaults, src/config.hpp:31-42 and :77-85
        self.multisample, self.bumpmap_scale, self.clamp, self.russian = 1, f32(10.0), f32(100000.0), f32(-1.0)
        self.output_scale, self.render_rounds, self.render_minutes, self.force_fresnell, self.reverse = -1.0, 1, None, False, 0
        self.lens_size, self.focus_plane, self.sky_color, self.sky_brightness = f32(0.0), f32(1.0), (0.0, 0.0, 0.0), f32(2.0)
        self.lights, self.thinglass, self.brdf, self.warnings = [], [], "", []
        for ln in it:
            vs = split(ln.strip())
            if not vs or vs[0][0] == "#":
                continue
            k, n = vs[0], len(vs)

            def one(what):
                if n != 2:
                    raise ConfigFileException(f"Invalid {what} line.")
                return vs[1]
            if k == "L":
                if n < 8 or n > 9:
                    raise ConfigFileException("Invalid light line.")
                pos3 = tuple(float(_stof(x, "light")) for x in vs[1:4])
                col = tuple(float(f32(_stof(x, "light") / f32(255))) for x in vs[4:7])
                self.lights.append(dict(pos=pos3, color=col, intensity=float(_stof(vs[7], "light")), size=float(_stof(vs[8], "light")) if n == 9 else 0.0))
            elif k in ("multisample", "ms"):
                self.multisample = _stoi(one("multisample"), "multisample")
                if self.multisample == 0:
                    raise ConfigFileException("Invalid multisample value.")
            elif k in ("sky", "skycolor"):
                if n < 4 or n > 5:
                    raise ConfigFileException("Invalid sky color line.")
                self.sky_color = tuple(float(f32(_stoi(x, "sky") / f32(255.0))) for x in vs[1:4])
                if n == 5:
                    self.sky_brightness = _stof(vs[4], "sky")
            elif k in ("lens", "lenssize", "lens_size"):
                self.lens_size = _stof(one("lens size"), "lens size")
                if self.lens_size < 0:
                    raise ConfigFileException("Lens size must be a poositive value.")
            elif k in ("focus", "focus_plane", "focus_dist"):
                self.focus_plane = _stof(one("focus plane"), "focus plane")
                if self.focus_plane < 0:
                    raise ConfigFileException("Focus plane must be a poositive value.")
            elif k in ("bump_scale", "bumpmap_scale", "bump", "bumpscale"):
                self.bumpmap_scale = _stof(one("bump scale config"), "bump scale")
            elif k == "clamp":
                self.clamp = _stof(one("clamp config"), "clamp")
            elif k in ("russian", "roulette"):
                self.russian = _stof(one("russian roulette config"), "russian")
            elif k == "rounds":
                self.render_rounds = _stoi(one("rounds config"), "rounds")
            elif k == "reverse":
                self.reverse = _stoi(one("reverse config"), "reverse")
            elif k == "brdf":
                b = one("brdf config")
                if b not in self.BRDFS:
                    raise ConfigFileException("Unknown BRDF type: " + b)
                self.brdf = self.BRDFS[b]
            elif k == "thinglass":
                self.thinglass.append(one("thinglass config"))
            elif k == "force_fresnell":
                self.force_fresnell = _stoi(one("force_fresnell config"), "force_fresnell") == 1
            else:
                self.warnings.append(f"WARNING: Unrecognized option `{k}` in the config file.")

    def get_camera(self, rotation=0.0):
        """ConfigRTC::GetCamera (src/config.cpp:175-189): xview = yview * xres / yres."""
        pos, lookat, up = self.camera_position, self.camera_lookat, self.camera_upvector
        if rotation != 0.0:
            from .scene import glm_rotate
            R = glm_rotate(f32(rotation) * f32(2.0) * f32(math.pi), up)[:3, :3]
            pos = lookat - (R @ (lookat - pos)).astype(f32)
        xview = f32(f32(self.yview * f32(self.xres)) / f32(self.yres))
        return camera_from_args(pos, lookat, up, self.yview, xview, self.xres, self.yres, self.focus_plane, self.lens_size)

    get_params = Config.get_params

    def build_scene(self, asset_dir=None, mesh_provider=None, builder=None):
        """InstallMaterials (nothing), InstallScene (the model file with its own materials), InstallLights, InstallSky
        (src/config.cpp:191-258)."""
        sb = builder or SceneBuilder()
        configdir = os.path.dirname(os.path.abspath(self.config_file_path))
        modelfile = _resolve(configdir, self.model_file, asset_dir)
        if modelfile is None:
            if mesh_provider is None:
                raise ConfigFileException(f'Unable to find model file "{os.path.join(configdir, self.model_file)}"')
            mesh_provider(sb, self.model_file)
        else:
            sb.load_obj(modelfile, np.eye(4, dtype=f32), import_materials=True, override_materials=False)
        for l in self.lights:
            sb.add_point_light(l["pos"], l["color"], l["intensity"], l["size"])
        sb.set_skybox_color(self.sky_color, float(self.sky_brightness))
        return sb

    def perform_post_check(self):
        return list(self.warnings)


def load_config(path, overrides=None):
    """main.cpp:166-179: the config class by file extension (.rtc -> ConfigRTC, .json -> ConfigJSON)."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".rtc":
        return ConfigRTC(path)
    if ext == ".json":
        return Config(path, overrides)
    raise ConfigFileException(f'Config file format "{ext[1:]}" not recognized')


def _resolve(configdir, rel, asset_dir):
    for base in ([asset_dir] if asset_dir else []) + [configdir]:
        p = os.path.join(base, rel)
        if os.path.exists(p):
            return p
    return None


def make_camera(pos, lookat, up=(0.0, 1.0, 0.0), fov=None, focal=None, xres=1, yres=1, focus_plane=1.0, lens_size=0.0):
    """A Camera as ConfigJSON::GetCamera builds it (src/config.cpp:332-370) from explicit values."""
    if focal is not None:
        yview = f32(focal)
        xview = f32(yview * f32(xres)) / f32(yres)
    else:
        xview = fov2xview(f32(fov))
        yview = f32(xview * f32(yres)) / f32(xres)
    return camera_from_args(pos, lookat, up, yview, xview, xres, yres, focus_plane, lens_size)


def camera_from_args(pos, lookat, up, yview, xview, xres, yres, focus_plane=1.0, lens_size=0.0):
    """Camera::Camera (src/camera.cpp:7-24) through the C ABI (rgk_camera_init): the derived members that
    RenderRound's `const Camera&` carries.  `.ctor` keeps the constructor arguments for fixtures and tests."""
    c = capi.Camera()
    lib = capi.load_product()
    args = dict(pos=[float(f32(x)) for x in pos], lookat=[float(f32(x)) for x in lookat], up=[float(f32(x)) for x in up],
                yview=float(f32(yview)), xview=float(f32(xview)), xsize=int(xres), ysize=int(yres),
                focus_plane=float(f32(focus_plane)), lens_size=float(f32(lens_size)))
    capi.check(lib, lib.rgk_camera_init(C.byref(c), capi.f3(*args["pos"]), capi.f3(*args["lookat"]), capi.f3(*args["up"]), args["yview"],
                                        args["xview"], args["xsize"], args["ysize"], args["focus_plane"], args["lens_size"]))
    c.ctor = args
    return c


def make_params(xres, yres, multisample, depth, clamp=1e7, russian=0.74, bumpscale=1.0, reverse=0,
                sampler=capi.SAMPLER_HALTON, flags=0):
    p = capi.Params()
    p.xres, p.yres, p.multisample, p.depth = xres, yres, multisample, depth
    p.clamp, p.russian, p.bumpmap_scale = float(f32(clamp)), float(f32(russian)), float(f32(bumpscale))
    p.force_fresnell, p.reverse, p.sampler, p.flags = 0, reverse, sampler, flags
    return p
