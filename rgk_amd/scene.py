"""Scene front-end: builds the flat arrays that cross the C ABI (include/rgk.h).

Host-side mirror of the reference's Scene *mutators* (the half of src/scene.cpp that
runs before Commit): RegisterMaterial (:77-96), LoadAiMesh (:130-208), AddPrimitive
(:210-248), AddPointLight (:683), SetSkybox* (src/scene.hpp:122-133), and the
material loaders Material::LoadFromJson / LoadFromAiMaterial (src/bxdf/bxdf.cpp:52-184)
with each BxDF's LoadFromJson (:207-330).  Commit() itself (epsilon, bbox, light
tables, accelerator) happens behind the ABI in rgk_scene_create.

assimp is not available: `load_obj` is a from-scratch OBJ/MTL reader that
follows the post-process steps the reference requests (src/config.cpp:196-228:
triangulate, generate flat or smooth normals, join identical vertices, tangent
space) -- results at this boundary are UNPINNED against assimp.
All arithmetic is float32, evaluated in the order GLM evaluates it.
"""
import ctypes as C
import math
import os
import re

import numpy as np

from . import capi

f32 = np.float32
HERE = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------- GLM-style float32 helpers
def glm_mat4_mul(a, b):
    """glm mat4*mat4: column j = a0*b[0][j] + a1*b[1][j] + a2*b[2][j] + a3*b[3][j] (row-major numpy)."""
    out = np.zeros((4, 4), dtype=f32)
    for j in range(4):
        col = a[:, 0] * b[0, j]
        col = (col + a[:, 1] * b[1, j]).astype(f32)
        col = (col + a[:, 2] * b[2, j]).astype(f32)
        col = (col + a[:, 3] * b[3, j]).astype(f32)
        out[:, j] = col
    return out


def glm_scale(v):
    m = np.eye(4, dtype=f32)
    m[0, 0], m[1, 1], m[2, 2] = v[0], v[1], v[2]
    return m


def glm_translate(v):
    m = np.eye(4, dtype=f32)
    m[0, 3], m[1, 3], m[2, 3] = v[0], v[1], v[2]
    return m


def glm_rotate(angle, axis):
    """glm::rotate(angle, axis) (gtc/matrix_transform), returned as a row-major 4x4."""
    a = f32(angle)
    c, s = f32(math.cos(float(a))), f32(math.sin(float(a)))
    ax = np.asarray(axis, dtype=f32)
    ax = (ax * (f32(1.0) / f32(math.sqrt(float(f32(ax[0] * ax[0]) + f32(ax[1] * ax[1]) + f32(ax[2] * ax[2])))))).astype(f32)
    t = ((f32(1.0) - c) * ax).astype(f32)
    R = np.eye(4, dtype=f32)
    # glm's Rotate[col][row]; numpy index is [row, col]
    R[0, 0] = c + t[0] * ax[0]
    R[1, 0] = t[0] * ax[1] + s * ax[2]
    R[2, 0] = t[0] * ax[2] - s * ax[1]
    R[0, 1] = t[1] * ax[0] - s * ax[2]
    R[1, 1] = c + t[1] * ax[1]
    R[2, 1] = t[1] * ax[2] + s * ax[0]
    R[0, 2] = t[2] * ax[0] + s * ax[1]
    R[1, 2] = t[2] * ax[1] - s * ax[0]
    R[2, 2] = c + t[2] * ax[2]
    return R.astype(f32)


def xform_point(T, v):
    """(T * vec4(v,1)).xyz with glm's (m0*x + m1*y) + (m2*z + m3*w) association."""
    v = np.asarray(v, dtype=f32).reshape(-1, 3)
    a = (T[:3, 0][None, :] * v[:, 0:1] + T[:3, 1][None, :] * v[:, 1:2]).astype(f32)
    b = (T[:3, 2][None, :] * v[:, 2:3] + T[:3, 3][None, :]).astype(f32)
    return (a + b).astype(f32)


def xform_dir(T, v):
    v = np.asarray(v, dtype=f32).reshape(-1, 3)
    a = (T[:3, 0][None, :] * v[:, 0:1] + T[:3, 1][None, :] * v[:, 1:2]).astype(f32)
    b = (T[:3, 2][None, :] * v[:, 2:3]).astype(f32)
    return (a + b).astype(f32)


def normalize_rows(v):
    d = (v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]).astype(f32)
    d = (d + v[:, 2] * v[:, 2]).astype(f32)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = (f32(1.0) / np.sqrt(d)).astype(f32)
    return (v * inv[:, None]).astype(f32)


# ----------------------------------------------------------------------- built-in primitives
def _quad_pattern():
    return [(1, 1), (1, -1), (-1, 1), (-1, -1), (-1, 1), (1, -1)]


def primitive_data(kind):
    """Vertex tuples of Primitives::planeY / trigY / cube (reference src/primitives.cpp:168-228),
    generated from the face pattern instead of a table.  Returns (pos, normal, uv, tangent)."""
    pos, nrm, uv, tan = [], [], [], []
    pat = _quad_pattern()

    def emit(face, n, t, count=6):
        for (u, v) in pat[:count]:
            pos.append(face(u, v))
            nrm.append(n)
            uv.append(((u + 1) / 2.0, (v + 1) / 2.0))
            tan.append(t)

    if kind in ("plane", "tri"):
        emit(lambda u, v: (u, 0.0, v), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0), 6 if kind == "plane" else 3)
    elif kind == "cube":
        emit(lambda u, v: (1.0, u, v), (1.0, 0.0, 0.0), (0.0, 0.0, 1.0))
        emit(lambda u, v: (-1.0, u, v), (-1.0, 0.0, 0.0), (0.0, 0.0, 1.0))
        emit(lambda u, v: (u, 1.0, v), (0.0, 1.0, 0.0), (1.0, 0.0, 0.0))
        emit(lambda u, v: (-u, -1.0, v), (0.0, -1.0, 0.0), (1.0, 0.0, 0.0))
        emit(lambda u, v: (v, u, 1.0), (0.0, 0.0, 1.0), (0.0, 1.0, 0.0))
        emit(lambda u, v: (v, u, -1.0), (0.0, 0.0, -1.0), (0.0, 1.0, 0.0))
    else:
        raise ValueError(kind)
    return (np.array(pos, dtype=f32), np.array(nrm, dtype=f32), np.array(uv, dtype=f32),
            np.array(tan, dtype=f32))


# ----------------------------------------------------------------------- textures
def gamma_lut():
    """byte -> Color(byte/255).gammaDecode(2.2) (src/texture.cpp:203,252-254), the 256 floats the
    reference's 8-bit loaders can produce."""
    return np.power((np.arange(256, dtype=f32) / f32(255.0)).astype(f32), f32(2.2)).astype(f32)


def gamma_decode_u8(img_u8):
    return gamma_lut()[img_u8]


def load_texture_file(path):
    """FileTexture::CreateNewFrom{PNG,JPEG} (src/texture.cpp:189-292): RGB float, gamma-2.2
    decoded; JPEGs are stored flipped vertically (Q12), PNGs are not.  Decoder = PIL
    (not the author's libjpeg: +-1 LSB possible)."""
    from PIL import Image
    ext = os.path.splitext(path)[1].lower()
    im = Image.open(path).convert("RGB")
    a = np.asarray(im, dtype=np.uint8)
    if ext in (".jpg", ".jpeg"):
        a = a[::-1]
    elif ext == ".png":
        pass
    else:
        raise ValueError(f"Texture format '{ext}' is not supported!")
    return np.ascontiguousarray(a)  # uint8; decoded through gamma_lut() (RGK_TEX_RGB8)


def load_hdr(path):
    """FileTexture::CreateNewFromHDR (src/texture.cpp:294-321): a Radiance RGBE file through the decoder the reference vendors
    (external/stb_image.h, stbi__hdr_load + stbi__hdr_convert, built STBI_ONLY_HDR) -> (h, w, 3) float32, top row first, no
    gamma, no flip.  Header: the first line must be `#?RADIANCE`, some line `FORMAT=32-bit_rle_rgbe`, a blank line, then
    `-Y <h> +X <w>`.  Scan lines are run-length encoded per channel when they start with 2, 2, <len hi < 128>, <len lo> (and
    8 <= w < 32768); otherwise the file is flat RGBE from there on.  Pixel = byte * 2^(e - 136), or 0 when e == 0."""
    b = open(path, "rb").read()
    pos = 0

    def token():
        nonlocal pos
        e = b.find(b"\n", pos)
        if e < 0:
            e = len(b)
        t = b[pos:e]
        pos = min(e + 1, len(b))
        return t[:1023]   # STBI__HDR_BUFLEN
    if token() != b"#?RADIANCE":
        raise ValueError(f"Failed to load texture '{path}': not HDR")
    valid = False
    while True:
        t = token()
        if len(t) == 0:
            break
        if t == b"FORMAT=32-bit_rle_rgbe":
            valid = True
        if pos >= len(b):
            break
    if not valid:
        raise ValueError(f"Failed to load texture '{path}': unsupported HDR format")
    t = token()
    m = re.match(rb"-Y ([+-]?\d+) *\+X ([+-]?\d+)", t)
    if not m:
        raise ValueError(f"Failed to load texture '{path}': unsupported HDR data layout")
    h, w = int(m.group(1)), int(m.group(2))
    data = np.frombuffer(b, dtype=np.uint8, offset=pos)
    rgbe = np.zeros((h, w, 4), dtype=np.uint8)

    def flat(start):
        n = h * w * 4
        if len(data) - start < n:
            raise ValueError(f"Failed to load texture '{path}': truncated HDR data")
        return data[start:start + n].reshape(h, w, 4)
    if w < 8 or w >= 32768:
        rgbe = flat(0)
    else:
        p = 0
        for j in range(h):
            if p + 4 > len(data):
                raise ValueError(f"Failed to load texture '{path}': truncated HDR data")
            c1, c2, ln = int(data[p]), int(data[p + 1]), int(data[p + 2])
            if c1 != 2 or c2 != 2 or (ln & 0x80):
                rgbe = flat(p)       # "not run-length encoded": stb restarts from pixel 0 with these very bytes, flat to the end
                break
            if ((ln << 8) | int(data[p + 3])) != w:
                raise ValueError(f"Failed to load texture '{path}': corrupt HDR (invalid decoded scanline length)")
            p += 4
            for k in range(4):
                i = 0
                while i < w:
                    count = int(data[p]); p += 1
                    if count > 128:
                        count -= 128
                        rgbe[j, i:i + count, k] = data[p]; p += 1
                    else:
                        rgbe[j, i:i + count, k] = data[p:p + count]; p += count
                    i += count
    e = rgbe[..., 3].astype(np.int32)
    f1 = np.where(e != 0, np.ldexp(np.float32(1.0), e - 136), np.float32(0.0)).astype(f32)
    return np.ascontiguousarray(rgbe[..., :3].astype(f32) * f1[..., None])


class SceneBuilder:
    def __init__(self):
        self.vertices, self.normals, self.tangents, self.texcoords = [], [], [], []
        self.n_verts = 0
        self.tri_idx, self.tri_mat = [], []
        self.n_tris = 0
        self.materials = []      # list of dicts
        self.mat_by_name = {}
        self.textures = []       # dicts {kind, color, data}
        self.tex_by_path = {}
        self.pointlights = []
        self.areal = []          # list of triangle-id lists
        self.sky = dict(mode=capi.SKY_COLOR, color=(0.0, 0.0, 0.0), intensity=1.0, rotate=0.0, tex=-1)
        self.geometry_label = "real"
        self.texture_fallback = None
        self._keep = []

    # ---- textures -------------------------------------------------------------
    def create_solid_texture(self, color):
        self.textures.append(dict(kind=capi.TEX_SOLID, color=tuple(float(f32(c)) for c in color), data=None))
        return len(self.textures) - 1

    def add_image_texture(self, key, data):
        """data: (h, w, 3) float32, already decoded/flipped the way the reference stores it."""
        if key in self.tex_by_path:
            return self.tex_by_path[key]
        data = np.ascontiguousarray(data, dtype=f32)
        self.textures.append(dict(kind=capi.TEX_RGB32F, color=(0.0, 0.0, 0.0), data=data))
        self.tex_by_path[key] = len(self.textures) - 1
        return self.tex_by_path[key]

    def add_image_texture8(self, key, data_u8, lut=None):
        """8-bit image (h, w, 3) uint8 as the reference's PNG/JPEG loaders store it + its byte->float table."""
        if key in self.tex_by_path:
            return self.tex_by_path[key]
        data_u8 = np.ascontiguousarray(data_u8, dtype=np.uint8)
        lut = gamma_lut() if lut is None else np.ascontiguousarray(lut, dtype=f32)
        self.textures.append(dict(kind=capi.TEX_RGB8, color=(0.0, 0.0, 0.0), data=data_u8, lut=lut))
        self.tex_by_path[key] = len(self.textures) - 1
        return self.tex_by_path[key]

    def get_texture(self, path):
        """Scene::GetTexture (src/scene.cpp:250-278): cached by path; failure -> no texture."""
        if path == "":
            return -1
        key = os.path.normpath(path)
        if key in self.tex_by_path:
            return self.tex_by_path[key]
        try:
            if os.path.splitext(path)[1].lower() == ".hdr":   # Scene::GetTexture dispatches on the extension, src/scene.cpp:258-270
                return self.add_image_texture(key, load_hdr(path))
            return self.add_image_texture8(key, load_texture_file(path))
        except Exception as e:  # "Failed to load texture ..., ignoring it."
            if self.texture_fallback is not None:  # labelled proxy assets (rgk_amd.proxy)
                t = self.texture_fallback(self, path)
                if t >= 0:
                    return t
            print(f"Failed to load texture '{path}' ({e}), ignoring it.")
            return -1

    # ---- materials ------------------------------------------------------------
    def register_material(self, m, override=False):
        """Scene::RegisterMaterial (src/scene.cpp:77-96)."""
        if m["name"] in self.mat_by_name and not override:
            return self.mat_by_name[m["name"]]
        self.materials.append(m)
        self.mat_by_name[m["name"]] = len(self.materials) - 1
        return len(self.materials) - 1

    def material_index(self, name):
        if name not in self.mat_by_name:
            raise RuntimeError(f'Error: Material named "{name}" was not defined')
        return self.mat_by_name[name]

    @staticmethod
    def new_material(name, kind):
        return dict(name=name, kind=kind, flags=0, emission=(0.0, 0.0, 0.0), roughness=0.0, ior=1.0,
                    amount=0.0, tex_diffuse=-1, tex_color=-1, tex_bump=-1, mix_m1=-1, mix_m2=-1)

    def _tex_or_solid(self, node, texkeys, colorkeys, texturedir, default):
        texfile = ""
        for k in texkeys:  # later keys are the inner default of the nested getOptionalString
            texfile = node.opt_str(k, texfile)
        if texfile != "":
            return self.get_texture(os.path.join(texturedir, texfile))
        for k in colorkeys:
            if node.has(k) or node.has(k + "255"):
                return self.create_solid_texture(node.req_vec3_255(k))
        return self.create_solid_texture(default)

    def load_material_from_json(self, node, texturedir, override=True):
        """Material::LoadFromJson + the BxDF's LoadFromJson (src/bxdf/bxdf.cpp:52-86,207-330)."""
        from .config import ConfigFileException
        name = node.req_str("name")
        emission = node.opt_vec3_255("emission", (0.0, 0.0, 0.0))
        bump = node.opt_str("bump-map", "")
        no_russian = node.opt_bool("no-russian", False)
        brdf = node.req_str("brdf")
        if brdf not in capi.BRDF_IDS:
            raise ConfigFileException("Unsupported BRDF id in config!")
        kind = capi.BRDF_IDS[brdf]
        m = self.new_material(name, kind)
        m["emission"] = tuple(float(x) for x in emission)
        m["flags"] = capi.MAT_NO_RUSSIAN if no_russian else 0
        if bump != "":
            m["tex_bump"] = self.get_texture(os.path.join(texturedir, bump))
        if kind == capi.BXDF_DIFFUSE:
            m["tex_diffuse"] = self._tex_or_solid(node, ["diffuse-texture"], ["diffuse"], texturedir, (0.5, 0.5, 0.5))
        elif kind == capi.BXDF_MIX:
            for key, slot in (("material1", "mix_m1"), ("material2", "mix_m2")):
                mn = node.req_str(key)
                if mn not in self.mat_by_name:
                    raise ConfigFileException(f'Material "{mn}", used for mixing, was not (yet) defined')
                m[slot] = self.mat_by_name[mn]
            m["amount"] = float(node.req_float("amount"))
        elif kind == capi.BXDF_MIRROR:
            m["tex_color"] = self._tex_or_solid(node, ["color-texture"], ["color"], texturedir, (1.0, 1.0, 1.0))
        elif kind == capi.BXDF_DIELECTRIC:
            m["ior"] = float(node.req_float("ior"))
            m["tex_color"] = self._tex_or_solid(node, ["specular-texture", "color-texture"], ["color"], texturedir, (1.0, 1.0, 1.0))
        elif kind == capi.BXDF_TRANSPARENT:
            pass
        else:  # LTC family
            if node.has("roughness"):
                m["roughness"] = float(node.req_float("roughness"))
            elif node.has("exponent"):
                e = node.req_float("exponent")
                m["roughness"] = float(f32(math.pow(float(f32(2.0) / f32(f32(2.0) + e)), 0.5)))
            else:
                raise ConfigFileException(f'Either "roughness" or "exponent" must be present for LTC BxDF in {node.name}')
            m["tex_color"] = self._tex_or_solid(node, ["specular-texture", "color-texture"], ["color", "specular"], texturedir, (0.0, 0.0, 0.0))
            if kind in (capi.BXDF_LTC_BECKMANN_DIFFUSE, capi.BXDF_LTC_GGX_DIFFUSE):
                m["tex_diffuse"] = self._tex_or_solid(node, ["diffuse-texture"], ["diffuse"], texturedir, (0.0, 0.0, 0.0))
        return self.register_material(m, override)

    def material_from_mtl(self, name, mtl, texture_directory, texture_loader=None):
        """Material::LoadFromAiMaterial (src/bxdf/bxdf.cpp:88-184): always BxDFLTCDiffuse<GGX>;
        roughness = sqrt(2/(2+Ns)) (assimp reports 4*Ns, the reference divides by 4)."""
        load = texture_loader or (lambda fn: self.get_texture(os.path.join(texture_directory, fn)))
        m = self.new_material(name, capi.BXDF_LTC_GGX_DIFFUSE)
        m["tex_diffuse"] = self.create_solid_texture(mtl.get("Kd", (0.6, 0.6, 0.6)))
        m["tex_color"] = self.create_solid_texture(mtl.get("Ks", (0.0, 0.0, 0.0)))
        m["emission"] = tuple(float(f32(x)) for x in mtl.get("Ke", (0.0, 0.0, 0.0)))
        if mtl.get("map_Kd"):
            t = load(mtl["map_Kd"])
            if t >= 0:
                m["tex_diffuse"] = t
        if mtl.get("map_Ks"):
            t = load(mtl["map_Ks"])
            if t >= 0:
                m["tex_color"] = t
        if mtl.get("map_Bump"):
            t = load(mtl["map_Bump"])
            if t >= 0:
                m["tex_bump"] = t
        phong_exp = f32(f32(mtl.get("Ns", 0.0)) * f32(4.0)) / f32(4.0)
        m["roughness"] = float(f32(math.pow(float(f32(2.0) / f32(f32(2.0) + phong_exp)), 0.5)))
        return m

    # ---- lights / sky -----------------------------------------------------------
    def add_point_light(self, pos, color, intensity, size):
        self.pointlights.append(dict(pos=tuple(float(x) for x in pos), color=tuple(float(x) for x in color),
                                     intensity=float(intensity), size=float(size)))

    def set_skybox_color(self, color, intensity):
        self.sky.update(mode=capi.SKY_COLOR, color=tuple(float(x) for x in color), intensity=float(intensity))

    def set_skybox_envmap(self, path, intensity, rotate):
        self.sky.update(mode=capi.SKY_ENVMAP, tex=self.get_texture(path), intensity=float(intensity), rotate=float(rotate))

    # ---- geometry -----------------------------------------------------------------
    def add_mesh(self, pos, nrm, uv, tan, faces, mat_index):
        """Append one mesh (already transformed).  Emissive material => one areal light
        listing its triangles (LoadAiMesh / AddPrimitive tail, src/scene.cpp:152,204-207)."""
        off = self.n_verts
        self.vertices.append(np.asarray(pos, dtype=f32))
        self.normals.append(np.asarray(nrm, dtype=f32))
        self.tangents.append(np.asarray(tan, dtype=f32))
        self.texcoords.append(np.asarray(uv, dtype=f32))
        self.n_verts += len(pos)
        faces = np.asarray(faces, dtype=np.uint32).reshape(-1, 3)
        self.tri_idx.append(faces + np.uint32(off))
        self.tri_mat.append(np.full(len(faces), mat_index, dtype=np.uint32))
        e = self.materials[mat_index]["emission"]
        if (e[0] > 0 or e[1] > 0 or e[2] > 0) and len(faces) > 0:
            self.areal.append(list(range(self.n_tris, self.n_tris + len(faces))))
        self.n_tris += len(faces)

    def object_transform(self, T, obj):
        """scale, rotate z/y/x (about the negative axes, degrees*0.0174533), translate --
        src/config.cpp:455-470 / :508-519."""
        scale = obj.opt_vec3("scale", (1.0, 1.0, 1.0))
        translate = obj.opt_vec3("translate", (0.0, 0.0, 0.0))
        rotate = obj.opt_vec3("rotate", (0.0, 0.0, 0.0))
        T = glm_mat4_mul(glm_scale(scale), T)
        T = glm_mat4_mul(glm_rotate(f32(0.0174533) * rotate[2], (0.0, 0.0, -1.0)), T)
        T = glm_mat4_mul(glm_rotate(f32(0.0174533) * rotate[1], (0.0, -1.0, 0.0)), T)
        T = glm_mat4_mul(glm_rotate(f32(0.0174533) * rotate[0], (-1.0, 0.0, 0.0)), T)
        T = glm_mat4_mul(glm_translate(translate), T)
        return T

    def add_primitive_from_json(self, obj):
        """The "primitive" branch of ConfigJSON::InstallScene (src/config.cpp:484-533)."""
        from .config import ConfigFileException
        kind = obj.req_str("primitive")
        T = np.eye(4, dtype=f32)
        if kind not in ("plane", "tri", "cube"):
            raise ConfigFileException(f"Value \"primitive\" in {obj.name} must be either 'cube' or 'plane'.")
        if kind == "cube":
            T = glm_mat4_mul(glm_scale((0.5, 0.5, 0.5)), T)
        axis = obj.opt_str("axis", "Y")
        if axis == "X":
            T = glm_mat4_mul(glm_rotate(f32(math.pi) / f32(2.0), (0.0, 0.0, 1.0)), T)
        elif axis == "Z":
            T = glm_mat4_mul(glm_rotate(f32(math.pi) / f32(2.0), (1.0, 0.0, 0.0)), T)
        elif axis != "Y":
            raise ConfigFileException(f'Optional value "axis" in {obj.name} must be either X, Y or Z.')
        T = self.object_transform(T, obj)
        texscale = obj.opt_vec3("texture-scale", (1.0, 1.0, 1.0))
        material = obj.req_str("material")
        self.add_primitive(kind, T, material, texscale)

    def add_primitive(self, kind, T, material, texscale=(1.0, 1.0, 1.0)):
        """Scene::AddPrimitive (src/scene.cpp:210-248)."""
        pos, nrm, uv, tan = primitive_data(kind)
        mi = self.material_index(material)
        pos = xform_point(T, pos)
        nrm = normalize_rows(xform_dir(T, nrm))
        tan = normalize_rows(xform_dir(T, tan))
        ts = np.asarray(texscale, dtype=f32)
        uv = (uv * ts[None, :2]).astype(f32)
        faces = np.arange(len(pos), dtype=np.uint32).reshape(-1, 3)
        self.add_mesh(pos, nrm, uv, tan, faces, mi)

    # ---- OBJ / MTL -------------------------------------------------------------------
    def load_obj(self, path, T, import_materials=True, override_materials=False, force_mat="",
                 smooth_normals=False):
        from .objload import load_obj_file
        meshes, mtls = load_obj_file(path, smooth_normals)
        texdir = os.path.dirname(path) + "/"
        if import_materials:
            for name, mtl in mtls.items():
                if name in self.mat_by_name and not override_materials:
                    continue
                self.register_material(self.material_from_mtl(name, mtl, texdir), override_materials)
        for mesh in meshes:
            name = force_mat if force_mat != "" else mesh["material"]
            mi = self.material_index(name)
            pos = xform_point(T, mesh["pos"])
            nrm = xform_dir(T, mesh["nrm"])
            tan = xform_dir(T, mesh["tan"])
            self.add_mesh(pos, nrm, mesh["uv"], tan, mesh["faces"], mi)

    # ---- flatten ------------------------------------------------------------------------
    def finalize(self):
        cat = lambda xs, w, dt: (np.ascontiguousarray(np.concatenate(xs).reshape(-1, w), dtype=dt)
                                 if xs else np.zeros((0, w), dtype=dt))
        self.V = cat(self.vertices, 3, f32)
        self.N = cat(self.normals, 3, f32)
        self.T = cat(self.tangents, 3, f32)
        self.UV = cat(self.texcoords, 2, f32)
        self.F = cat(self.tri_idx, 3, np.uint32)
        self.FM = np.ascontiguousarray(np.concatenate(self.tri_mat), dtype=np.uint32) if self.tri_mat else np.zeros(0, np.uint32)
        return self

    # ---- flat-array fixtures (tests/golden/*.npz): the built scene, no pickles -------------
    def save_npz(self, path, extra=None):
        import json
        self.finalize()
        meta = dict(materials=self.materials, pointlights=self.pointlights, areal=self.areal, sky=self.sky,
                    geometry_label=self.geometry_label,
                    textures=[dict(kind=t["kind"], color=t["color"]) for t in self.textures], extra=extra or {})
        arrays = dict(V=self.V, N=self.N, T=self.T, UV=self.UV, F=self.F, FM=self.FM,
                      meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
        for i, t in enumerate(self.textures):
            if t["kind"] in (capi.TEX_RGB32F, capi.TEX_RGB8):
                arrays[f"tex{i}"] = t["data"]
                if t["kind"] == capi.TEX_RGB8:
                    arrays[f"lut{i}"] = t["lut"]
        np.savez_compressed(path, **arrays)

    @classmethod
    def load_npz(cls, path):
        import json
        z = np.load(path, allow_pickle=False)
        meta = json.loads(bytes(z["meta"]).decode())
        sb = cls()
        sb.vertices, sb.normals, sb.tangents, sb.texcoords = [z["V"]], [z["N"]], [z["T"]], [z["UV"]]
        sb.tri_idx, sb.tri_mat = [z["F"]], [z["FM"]]
        sb.n_verts, sb.n_tris = len(z["V"]), len(z["F"])
        sb.materials = meta["materials"]
        for m in sb.materials:
            m["emission"] = tuple(m["emission"])
        sb.mat_by_name = {m["name"]: i for i, m in enumerate(sb.materials)}
        sb.pointlights, sb.areal, sb.sky = meta["pointlights"], meta["areal"], meta["sky"]
        sb.geometry_label = meta.get("geometry_label", "real")
        sb.textures = [dict(kind=t["kind"], color=tuple(t["color"]),
                            data=(np.ascontiguousarray(z[f"tex{i}"]) if t["kind"] != capi.TEX_SOLID else None),
                            lut=(np.ascontiguousarray(z[f"lut{i}"]) if t["kind"] == capi.TEX_RGB8 else None))
                       for i, t in enumerate(meta["textures"])]
        sb.extra = meta.get("extra", {})
        return sb.finalize()

    def uses_ltc(self):
        return any(m["kind"] >= capi.BXDF_LTC_BECKMANN for m in self.materials)

    def to_desc(self):
        """Build the rgk_scene_desc (every buffer is kept alive on self and on the returned descriptor)."""
        self.finalize()
        keep = self._keep = []
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        up = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32))
        d = capi.SceneDesc()
        d.n_vertices = len(self.V)
        d.vertices, d.normals, d.tangents, d.texcoords = fp(self.V), fp(self.N), fp(self.T), fp(self.UV)
        d.n_triangles = len(self.F)
        d.tri_indices, d.tri_material = up(self.F), up(self.FM)
        mats = (capi.Material * max(1, len(self.materials)))()
        for i, m in enumerate(self.materials):
            mm = mats[i]
            mm.kind, mm.flags = m["kind"], m["flags"]
            mm.emission[:] = m["emission"]
            mm.roughness, mm.ior, mm.amount = m["roughness"], m["ior"], m["amount"]
            mm.tex_diffuse, mm.tex_color, mm.tex_bump = m["tex_diffuse"], m["tex_color"], m["tex_bump"]
            mm.mix_m1, mm.mix_m2 = m["mix_m1"], m["mix_m2"]
        d.n_materials, d.materials = len(self.materials), mats
        texs = (capi.Texture * max(1, len(self.textures)))()
        for i, t in enumerate(self.textures):
            tt = texs[i]
            tt.kind = t["kind"]
            tt.color[:] = t["color"]
            if t["kind"] == capi.TEX_RGB32F:
                tt.height, tt.width = t["data"].shape[0], t["data"].shape[1]
                tt.texels = fp(t["data"])
            elif t["kind"] == capi.TEX_RGB8:
                tt.height, tt.width = t["data"].shape[0], t["data"].shape[1]
                tt.texels8 = t["data"].ctypes.data_as(C.POINTER(C.c_uint8))
                tt.lut = fp(t["lut"])
        d.n_textures, d.textures = len(self.textures), texs
        pls = (capi.PointLight * max(1, len(self.pointlights)))()
        for i, l in enumerate(self.pointlights):
            pls[i].pos[:] = l["pos"]
            pls[i].color[:] = l["color"]
            pls[i].intensity, pls[i].size = l["intensity"], l["size"]
        d.n_pointlights, d.pointlights = len(self.pointlights), pls
        offs = np.zeros(len(self.areal) + 1, dtype=np.uint32)
        for i, a in enumerate(self.areal):
            offs[i + 1] = offs[i] + len(a)
        tris = np.array([t for a in self.areal for t in a], dtype=np.uint32)
        d.n_areal_lights = len(self.areal)
        d.areal_offsets, d.areal_tris = up(offs), up(tris)
        d.sky_mode = self.sky["mode"]
        d.sky_color[:] = self.sky["color"]
        d.sky_intensity, d.sky_rotate, d.sky_texture = self.sky["intensity"], self.sky["rotate"], self.sky["tex"]
        ggx = np.fromfile(os.path.join(HERE, "data", "ltc_ggx.f32"), dtype=f32)
        bek = np.fromfile(os.path.join(HERE, "data", "ltc_beckmann.f32"), dtype=f32)
        assert ggx.size == 5 * 4096 and bek.size == 5 * 4096
        d.ltc_ggx, d.ltc_beckmann = fp(ggx), fp(bek)
        env = os.environ.get("RGK_BVH_BUILD", "").lower()  # gpu | host: force a builder (default: by size, RGK_BUILD_AUTO)
        d.build_flags = capi.BUILD_DEVICE if env in ("gpu", "device") else (capi.BUILD_HOST_SAH if env in ("host", "cpu", "sah") else getattr(self, "build_flags", capi.BUILD_AUTO))
        keep += [mats, texs, pls, offs, tris, ggx, bek]
        d._keep = keep  # the descriptor owns its buffers too: `builder.to_desc()` on a temporary builder stays valid
        return d
