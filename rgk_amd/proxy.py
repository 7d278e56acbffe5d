"""Deterministic procedural stand-ins for the assets missing from the reference checkout.

`sponza.obj` (and dragon.obj, cloudy1.hdr) are not in the reference tree
(.MISSING_LARGE_BLOBS) and nothing can be downloaded, so configs 3-5 of
BASELINE.json have no geometry.  SURVEY 8(d) prescribes a labelled proxy:

  sponza_proxy(sb):  a two-storey colonnaded atrium with an open roof, ~66 k triangles,
      uv-mapped, using the 20 material definitions of scenes/sponza-fixed/sponza.mtl
      (names, Ns, Kd, Ks and which maps they bind -- transcribed as a table) and the
      17 JPGs the reference does ship for them (decoded bytes, rgk_amd/data/sponza_textures_u8.npz;
      gamma-2.2 decoded through the same LUT path as the reference's JPEG loader).

Every number measured on it is labelled geometry="proxy".  If RGK_ASSET_DIR holds the
real mesh the config loader uses that instead (rgk_amd.config._resolve).
"""
import math

import numpy as np

from .scene import gamma_decode_u8

f32 = np.float32

# name: (Ns, Kd, Ks, map_Kd, map_Bump)   -- scenes/sponza-fixed/sponza.mtl
SPONZA_MTL = {
    "sp_00_luk_mal1": (49.019608, (0.745098, 0.709804, 0.674510), (0, 0, 0), "01_St_kp.JPG", "01_St_kp-bump.jpg"),
    "sp_00_luk_mali": (49.019608, (0.745098, 0.709804, 0.674510), (0, 0, 0), "sp_luk.JPG", "sp_luk-bump.JPG"),
    "sp_00_pod": (200.019608, (0.627451, 0.572549, 0.560784), (0.317451, 0.282549, 0.270784), "KAMEN.JPG", "KAMEN-bump.jpg"),
    "sp_00_prozor": (49.019608, (1.0, 1.0, 1.0), (0, 0, 0), "prozor1.JPG", "prozor1.JPG"),
    "sp_00_stup": (49.019608, (0.737255, 0.709804, 0.670588), (0, 0, 0), "01_STUB.JPG", "01_STUB-bump.jpg"),
    "sp_00_svod": (0.0, (0.941177, 0.866667, 0.737255), (0.034039, 0.032314, 0.029333), "KAMEN-stup.JPG", "KAMEN-stup.JPG"),
    "sp_00_vrata_kock": (19.607843, (0.784314, 0.784314, 0.784314), (0, 0, 0), "vrata_ko.JPG", "vrata_ko.JPG"),
    "sp_00_vrata_krug": (19.607843, (0.784314, 0.784314, 0.784314), (0, 0, 0), "vrata_kr.JPG", "vrata_kr.JPG"),
    "sp_00_zid": (49.019608, (0.627451, 0.572549, 0.560784), (0, 0, 0), "KAMEN.JPG", "KAMEN-bump.jpg"),
    "sp_01_luk_a": (49.019608, (0.745098, 0.709804, 0.674510), (0, 0, 0), "sp_luk.JPG", "sp_luk-bump.JPG"),
    "sp_01_stub": (49.019608, (0.737255, 0.709804, 0.670588), (0, 0, 0), "01_STUB.JPG", "01_STUB-bump.jpg"),
    "sp_01_stub_baza": (49.019608, (0.800000, 0.784314, 0.749020), (0, 0, 0), "01_S_ba.JPG", "01_S_ba.JPG"),
    "sp_01_stub_baza_": (19.607843, (0.784314, 0.784314, 0.784314), (0, 0, 0), None, None),
    "sp_01_stub_kut": (49.019608, (0.737255, 0.709804, 0.670588), (0, 0, 0), "01_STUB.JPG", "01_STUB-bump.jpg"),
    "sp_01_stup": (49.019608, (0.827451, 0.800000, 0.768628), (0, 0, 0), "x01_st.JPG", None),
    "sp_01_stup_baza": (49.019608, (0.800000, 0.784314, 0.749020), (0, 0, 0), "01_S_ba.JPG", "01_S_ba.JPG"),
    "sp_02_reljef": (49.019608, (0.529412, 0.498039, 0.490196), (0, 0, 0), "reljef.JPG", "reljef-bump.jpg"),
    "sp_svod_kapitel": (49.019608, (0.713726, 0.705882, 0.658824), (0, 0, 0), "00_skap.JPG", "00_skap.JPG"),
    "sp_vijenac": (49.019608, (0.713726, 0.705882, 0.658824), (0, 0, 0), "00_skap.JPG", "00_skap.JPG"),
    "sp_zid_vani": (49.019608, (0.627451, 0.572549, 0.560784), (0, 0, 0), "KAMEN.JPG", "KAMEN-bump.jpg"),
}
# (width, height) of the shipped JPGs (PIL Image.size)
SPONZA_TEX_DIMS = {
    "00_skap.JPG": (903, 99), "01_STUB-bump.jpg": (1024, 704), "01_STUB.JPG": (1024, 704), "01_S_ba.JPG": (155, 23),
    "01_St_kp-bump.jpg": (846, 94), "01_St_kp.JPG": (423, 47), "KAMEN-bump.jpg": (640, 477), "KAMEN-stup.JPG": (640, 477),
    "KAMEN.JPG": (640, 477), "prozor1.JPG": (305, 357), "reljef-bump.jpg": (294, 251), "reljef.JPG": (512, 437),
    "sp_luk-bump.JPG": (1024, 150), "sp_luk.JPG": (1021, 150), "vrata_ko.JPG": (410, 489), "vrata_kr.JPG": (423, 807),
    "x01_st.JPG": (207, 477),
}


def _value_noise(rng, w, h, cells):
    """Smooth tileable value noise in [0,1], (h, w)."""
    g = rng.random((cells + 1, cells + 1))
    g[-1, :] = g[0, :]
    g[:, -1] = g[:, 0]
    ys = np.linspace(0, cells, h, endpoint=False)
    xs = np.linspace(0, cells, w, endpoint=False)
    y0, x0 = ys.astype(int), xs.astype(int)
    fy, fx = ys - y0, xs - x0
    fy, fx = fy * fy * (3 - 2 * fy), fx * fx * (3 - 2 * fx)
    a = g[np.ix_(y0, x0)] * (1 - fx)[None, :] + g[np.ix_(y0, x0 + 1)] * fx[None, :]
    b = g[np.ix_(y0 + 1, x0)] * (1 - fx)[None, :] + g[np.ix_(y0 + 1, x0 + 1)] * fx[None, :]
    return a * (1 - fy)[:, None] + b * fy[:, None]


def procedural_texture(name, w, h):
    """An 8-bit stone-like RGB image (h, w, 3) uint8, deterministic in `name`."""
    seed = sum((i + 1) * ord(c) for i, c in enumerate(name)) & 0xFFFFFFFF
    rng = np.random.default_rng(seed)
    n = 0.55 * _value_noise(rng, w, h, 8) + 0.3 * _value_noise(rng, w, h, 32) + 0.15 * rng.random((h, w))
    # mortar lines / blocks
    by, bx = max(8, h // 6), max(8, w // 8)
    yy, xx = np.mgrid[0:h, 0:w]
    mortar = ((yy % by) < 2) | (((xx + (yy // by % 2) * (bx // 2)) % bx) < 2)
    n = np.where(mortar, n * 0.45, n)
    bump = "bump" in name.lower()
    if bump:
        img = np.repeat((n * 255).astype(np.uint8)[..., None], 3, axis=2)
    else:
        tint = 0.75 + 0.25 * rng.random(3)
        img = (np.clip(0.35 + 0.6 * n[..., None] * tint[None, None, :], 0, 1) * 255).astype(np.uint8)
    return img


# ----------------------------------------------------------------------- mesh helpers
class Mesh:
    def __init__(self):
        self.p, self.n, self.uv, self.t, self.f = [], [], [], [], []
        self.nv = 0

    def add(self, p, n, uv, t, f):
        p = np.asarray(p, dtype=f32).reshape(-1, 3)
        self.p.append(p)
        self.n.append(np.asarray(n, dtype=f32).reshape(-1, 3))
        self.uv.append(np.asarray(uv, dtype=f32).reshape(-1, 2))
        self.t.append(np.asarray(t, dtype=f32).reshape(-1, 3))
        self.f.append(np.asarray(f, dtype=np.int64).reshape(-1, 3) + self.nv)
        self.nv += len(p)

    def ntris(self):
        return sum(len(x) for x in self.f)

    def emit(self, sb, material):
        if not self.p:
            return
        sb.add_mesh(np.concatenate(self.p), np.concatenate(self.n), np.concatenate(self.uv), np.concatenate(self.t),
                    np.concatenate(self.f).astype(np.uint32), sb.material_index(material))


def _grid_faces(nu, nv):
    """(nu+1) x (nv+1) vertex grid, row-major in v then u -> 2*nu*nv triangles."""
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    a = (i * (nv + 1) + j).ravel()
    b, c, d = a + 1, a + (nv + 1), a + (nv + 2)
    return np.concatenate([np.stack([a, c, b], 1), np.stack([b, c, d], 1)])


def quad(mesh, origin, eu, ev, nu, nv, uvscale=0.5, flip=False):
    """Planar patch origin + u*eu + v*ev, u,v in [0,1], subdivided nu x nv; normal = eu x ev."""
    origin, eu, ev = (np.asarray(x, dtype=np.float64) for x in (origin, eu, ev))
    u, v = np.meshgrid(np.linspace(0, 1, nu + 1), np.linspace(0, 1, nv + 1), indexing="ij")
    p = origin[None, :] + u.reshape(-1, 1) * eu[None, :] + v.reshape(-1, 1) * ev[None, :]
    n = np.cross(eu, ev)
    n /= np.linalg.norm(n)
    lu, lv = np.linalg.norm(eu), np.linalg.norm(ev)
    uv = np.stack([u.ravel() * lu * uvscale, v.ravel() * lv * uvscale], 1)
    f = _grid_faces(nu, nv)
    if flip:
        n = -n
        f = f[:, ::-1]
    t = eu / lu
    mesh.add(p, np.tile(n, (len(p), 1)), uv, np.tile(t, (len(p), 1)), f)


def box(mesh, lo, hi, sub=1, uvscale=0.5):
    lo, hi = np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)
    d = hi - lo
    X, Y, Z = np.array([d[0], 0, 0]), np.array([0, d[1], 0]), np.array([0, 0, d[2]])
    quad(mesh, lo, Z, Y, sub, sub, uvscale)                       # -x
    quad(mesh, lo + X, Y, Z, sub, sub, uvscale)                   # +x
    quad(mesh, lo, X, Z, sub, sub, uvscale)                       # -y
    quad(mesh, lo + Y, Z, X, sub, sub, uvscale)                   # +y
    quad(mesh, lo, Y, X, sub, sub, uvscale)                       # -z
    quad(mesh, lo + Z, X, Y, sub, sub, uvscale)                   # +z


def cylinder(mesh, base, radius, height, segs, rings, taper=0.92, uvscale=0.5):
    """Smooth-shaded column shaft along +y."""
    th = np.linspace(0, 2 * math.pi, segs + 1)
    ys = np.linspace(0, 1, rings + 1)
    T, Y = np.meshgrid(th, ys, indexing="ij")
    r = radius * (1 - (1 - taper) * Y)
    p = np.stack([base[0] + r * np.cos(T), base[1] + Y * height, base[2] + r * np.sin(T)], -1).reshape(-1, 3)
    n = np.stack([np.cos(T), np.zeros_like(T), np.sin(T)], -1).reshape(-1, 3)
    uv = np.stack([T.ravel() / (2 * math.pi) * 2 * math.pi * radius * uvscale, Y.ravel() * height * uvscale], 1)
    t = np.stack([-np.sin(T), np.zeros_like(T), np.cos(T)], -1).reshape(-1, 3)
    mesh.add(p, n, uv, t, _grid_faces(segs, rings)[:, ::-1])


def arch(mesh, c0, c1, y0, radius, thick, depth, segs, uvscale=0.5):
    """Semicircular arch between columns at c0 and c1 (x,z), springing at height y0: the
    underside (smooth) plus the two faces above it up to the crown, as strips."""
    c0, c1 = np.asarray(c0, dtype=np.float64), np.asarray(c1, dtype=np.float64)
    mid = 0.5 * (c0 + c1)
    axis = (c1 - c0)
    span = np.linalg.norm(axis)
    axis /= span
    side = np.array([-axis[1], axis[0]])  # in xz
    a = np.linspace(0, math.pi, segs + 1)
    rr = span / 2 - radius
    for s, sgn in ((-1, -1.0), (1, 1.0)):
        # face strips from the intrados up to a flat top at y0 + rr + thick
        xs = mid[None, :] + (-np.cos(a))[:, None] * rr * axis[None, :] + s * (depth / 2) * side[None, :]
        yi = y0 + np.sin(a) * rr
        top = np.full_like(yi, y0 + rr + thick)
        p = np.concatenate([np.stack([xs[:, 0], yi, xs[:, 1]], 1), np.stack([xs[:, 0], top, xs[:, 1]], 1)])
        n = np.tile(np.array([sgn * side[0], 0, sgn * side[1]]), (len(p), 1))
        uv = np.stack([(p[:, 0] * axis[0] + p[:, 2] * axis[1]) * uvscale, p[:, 1] * uvscale], 1)
        t = np.tile(np.array([axis[0], 0, axis[1]]), (len(p), 1))
        i = np.arange(segs)
        f = np.concatenate([np.stack([i, i + 1, i + segs + 1], 1), np.stack([i + 1, i + segs + 2, i + segs + 1], 1)])
        mesh.add(p, n, uv, t, f if s > 0 else f[:, ::-1])
    # intrados (underside), smooth normals pointing to the centre
    A, S = np.meshgrid(a, np.linspace(-0.5, 0.5, 3), indexing="ij")
    px = mid[0] - np.cos(A) * rr * axis[0] + S * depth * side[0]
    pz = mid[1] - np.cos(A) * rr * axis[1] + S * depth * side[1]
    py = y0 + np.sin(A) * rr
    p = np.stack([px, py, pz], -1).reshape(-1, 3)
    n = np.stack([np.cos(A) * axis[0], -np.sin(A), np.cos(A) * axis[1]], -1).reshape(-1, 3)
    uv = np.stack([A.ravel() * rr * uvscale, (S.ravel() + 0.5) * depth * uvscale], 1)
    t = np.stack([np.sin(A) * axis[0], np.cos(A), np.sin(A) * axis[1]], -1).reshape(-1, 3)
    mesh.add(p, n, uv, t, _grid_faces(segs, 2))


def vault(mesh, x0, x1, z0, z1, y0, rise, nu, nv, uvscale=0.5):
    """Barrel vault over the bay [x0,x1] x [z0,z1], axis along x, seen from below."""
    u, a = np.meshgrid(np.linspace(0, 1, nu + 1), np.linspace(0, math.pi, nv + 1), indexing="ij")
    zc, zr = 0.5 * (z0 + z1), 0.5 * (z1 - z0)
    p = np.stack([x0 + u * (x1 - x0), y0 + np.sin(a) * rise, zc - np.cos(a) * zr], -1).reshape(-1, 3)
    n = np.stack([np.zeros_like(a), -np.sin(a) * zr, np.cos(a) * rise], -1).reshape(-1, 3)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    uv = np.stack([u.ravel() * (x1 - x0) * uvscale, a.ravel() * zr * uvscale], 1)
    t = np.tile(np.array([1.0, 0, 0]), (len(p), 1))
    mesh.add(p, n, uv, t, _grid_faces(nu, nv))


def icosphere(mesh, centre, radius, level, uvscale=1.0):
    t = (1 + 5 ** 0.5) / 2
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6],
                  [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7],
                  [9, 8, 1]])
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    for _ in range(level):
        m = {}
        verts = list(v)

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in m:
                q = verts[a] + verts[b]
                verts.append(q / np.linalg.norm(q))
                m[k] = len(verts) - 1
            return m[k]
        nf = []
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        v, f = np.array(verts), np.array(nf)
    p = np.asarray(centre)[None, :] + v * radius
    uv = np.stack([np.arctan2(v[:, 2], v[:, 0]) / (2 * math.pi) + 0.5, np.arccos(np.clip(v[:, 1], -1, 1)) / math.pi], 1) * uvscale
    tng = np.stack([-v[:, 2], np.zeros(len(v)), v[:, 0]], 1)
    tl = np.linalg.norm(tng, axis=1, keepdims=True)
    tng = np.where(tl > 1e-6, tng / np.maximum(tl, 1e-6), np.array([[1.0, 0, 0]]))
    mesh.add(p, v, uv, tng, f[:, ::-1])


# ----------------------------------------------------------------------- the scene
_SHIPPED = None


def shipped_sponza_textures():
    """The decoded bytes of the 17 JPGs the reference ships under scenes/sponza-fixed/ (rgk_amd/data/sponza_textures_u8.npz,
    made by tools/make_fixtures.py sponza_textures), or {} when the file is absent."""
    global _SHIPPED
    if _SHIPPED is None:
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "sponza_textures_u8.npz")
        _SHIPPED = dict(np.load(path, allow_pickle=False)) if os.path.exists(path) else {}
    return _SHIPPED


def proxy_texture(sb, name):
    """One of the shipped Sponza JPGs by file name -- its decoded bytes, or (file absent) a procedural stand-in of the same
    dimensions -- as the reference's JPEG loader stores it: flipped vertically, gamma-2.2 decoded through the byte table."""
    import os
    name = os.path.basename(name) if name else name
    if name is None or name not in SPONZA_TEX_DIMS:
        return -1
    key = "proxy:" + name
    if key not in sb.tex_by_path:
        w, h = SPONZA_TEX_DIMS[name]
        shipped = shipped_sponza_textures()
        if name in shipped:
            img = shipped[name]
            assert img.shape == (h, w, 3), (name, img.shape)
        else:
            img = procedural_texture(name, w, h)
            sb.texture_label = "procedural"
        sb.add_image_texture8(key, np.ascontiguousarray(img[::-1]))  # JPEGs are stored flipped (Q12)
    return sb.tex_by_path[key]


def install_sponza_materials(sb):
    def tex(name):
        return proxy_texture(sb, name)

    for name, (ns, kd, ks, mkd, mbump) in SPONZA_MTL.items():
        mtl = dict(Ns=ns, Kd=kd, Ks=ks, map_Kd=mkd, map_Bump=mbump)
        sb.register_material(sb.material_from_mtl(name, mtl, "", texture_loader=tex), False)


# No two faces of the proxy share a plane over a common area: where a box stands on a floor, meets a wall or a ceiling, its
# face is pulled OFF (>> 2 epsilon = 6.4e-4 for this scene) past the surface instead of lying in it.  Coincident surfaces tie
# exactly, and an exact tie is decided by traversal order -- the oracle's kd-tree and the HIP path's BVH then disagree on
# which of the two triangles was hit (round 1: 8e-4 of all rays on this scene, the whole reason its image tolerance was 1e-2).
OFF = 0.004


def sponza_proxy(sb, detail=1.0):
    """Fill `sb` with the atrium.  Axes as in Sponza: x long, y up, z across.  ~66 k triangles
    at detail=1."""
    install_sponza_materials(sb)
    sb.geometry_label = "proxy"
    D = lambda n: max(1, int(round(n * detail)))
    L, Wd, H1, H2 = 14.0, 6.0, 4.6, 9.0       # half-length, half-width, storey heights
    aisle = 2.4                               # aisle depth behind the colonnade
    nave = Wd - aisle                         # |z| of the colonnade line
    roof = 11.5
    parts = {k: Mesh() for k in SPONZA_MTL}
    # floor + upper gallery floors
    quad(parts["sp_00_pod"], (-L, 0, -Wd), (0, 0, 2 * Wd), (2 * L, 0, 0), D(30), D(70))
    for sgn in (-1, 1):
        z0, z1 = (nave, Wd) if sgn > 0 else (-Wd, -nave)
        quad(parts["sp_00_pod"], (-L, H1, z0), (0, 0, z1 - z0), (2 * L, 0, 0), D(6), D(60))
        quad(parts["sp_vijenac"], (-L, H1 - 0.25, z0), (2 * L, 0, 0), (0, 0, z1 - z0), D(60), D(6))  # slab underside
    # outer walls (facing in), two storeys + parapet
    wall = parts["sp_00_zid"]
    quad(wall, (-L, 0, -Wd), (2 * L, 0, 0), (0, roof, 0), D(60), D(24))
    quad(wall, (L, 0, Wd), (-2 * L, 0, 0), (0, roof, 0), D(60), D(24))
    quad(parts["sp_zid_vani"], (-L, 0, Wd), (0, 0, -2 * Wd), (0, roof, 0), D(30), D(28))
    quad(parts["sp_zid_vani"], (L, 0, -Wd), (0, 0, 2 * Wd), (0, roof, 0), D(30), D(28))
    # roof ring: covers the aisles, leaves the nave open to the sky/sun
    for sgn in (-1, 1):
        z0, z1 = (nave, Wd) if sgn > 0 else (-Wd, -nave)
        quad(parts["sp_00_svod"], (-L, roof, z0), (2 * L, 0, 0), (0, 0, z1 - z0), D(40), D(4))
    # colonnades
    xs = np.linspace(-L + 1.5, L - 1.5, 10)
    for storey, (y0, h, mat_col, mat_base, mat_arch) in enumerate(
            ((0.0, 3.2, "sp_00_stup", "sp_01_stub_baza", "sp_00_luk_mali"), (H1, 2.9, "sp_01_stup", "sp_01_stup_baza", "sp_01_luk_a"))):
        for sgn in (-1, 1):
            z = sgn * nave
            for i, x in enumerate(xs):
                r = 0.36 if storey == 0 else 0.28
                box(parts[mat_base], (x - r * 1.5, y0 - OFF, z - r * 1.5), (x + r * 1.5, y0 + 0.35, z + r * 1.5), D(2))
                cylinder(parts[mat_col], (x, y0 + 0.35, z), r, h - 0.7, D(24), D(12))
                box(parts["sp_svod_kapitel"], (x - r * 1.6, y0 + h - 0.35, z - r * 1.6), (x + r * 1.6, y0 + h, z + r * 1.6), D(2))
                if i + 1 < len(xs):
                    arch(parts[mat_arch], (x, z), (xs[i + 1], z), y0 + h, r * 1.2, 0.45, 0.7, D(20))
                    # aisle vault behind this bay
                    zc0, zc1 = (nave, Wd) if sgn > 0 else (-Wd, -nave)
                    vault(parts["sp_00_svod"], x, xs[i + 1], zc0, zc1, y0 + h, 0.9, D(6), D(12))
            # frieze above the arches
            yb = y0 + h + 0.5 * (xs[1] - xs[0]) + 0.2
            if sgn < 0:
                quad(parts["sp_02_reljef"], (-L, yb, z + 0.35), (2 * L, 0, 0), (0, 0.6, 0), D(50), D(3))
            else:
                quad(parts["sp_02_reljef"], (L, yb, z - 0.35), (-2 * L, 0, 0), (0, 0.6, 0), D(50), D(3))
    # doors / windows on the end walls, cornice boxes
    quad(parts["sp_00_vrata_krug"], (-L + 0.02, 0, -1.2), (0, 0, 2.4), (0, 3.6, 0), D(6), D(8), flip=True)
    quad(parts["sp_00_vrata_kock"], (L - 0.02, 0, 1.2), (0, 0, -2.4), (0, 3.6, 0), D(6), D(8), flip=True)
    for x in np.linspace(-L + 3, L - 3, 6):
        quad(parts["sp_00_prozor"], (x - 0.6, H1 + 1.0, -Wd + 0.02), (1.2, 0, 0), (0, 1.6, 0), D(3), D(4))
        quad(parts["sp_00_prozor"], (x + 0.6, H1 + 1.0, Wd - 0.02), (-1.2, 0, 0), (0, 1.6, 0), D(3), D(4))
    for sgn in (-1, 1):
        box(parts["sp_vijenac"], (-L + OFF, H1 - 0.25 - OFF, sgn * nave - 0.25), (L - OFF, H1 + 0.1, sgn * nave + 0.25), D(4))
        box(parts["sp_vijenac"], (-L + OFF, roof - 0.4, sgn * nave - 0.3), (L - OFF, roof - OFF, sgn * nave + 0.3), D(4))
    # corner pilasters, small arches at the ends, urns in the nave (fine detail, like the lion heads / vases)
    for x in (-L + 0.3 + 2 * OFF, L - 0.3 - 2 * OFF):
        for z in (-nave, nave):
            box(parts["sp_01_stub_kut"], (x - 0.3, -OFF, z - 0.32), (x + 0.3, roof - 2 * OFF, z + 0.32), D(6))
    for i, x in enumerate(np.linspace(-L + 3, L - 3, 8)):
        for z in (-1.4, 1.4):
            icosphere(parts["sp_00_luk_mal1"], (x, 0.55, z), 0.45, 2 if detail >= 0.75 else 1)
            box(parts["sp_01_stub_baza_"], (x - 0.3, -OFF, z - 0.3), (x + 0.3, 0.12, z + 0.3), 1)
            cylinder(parts["sp_01_stub"], (x, 0.1, z), 0.12, 0.2, D(12), 1)
    total = 0
    for name, m in parts.items():
        m.emit(sb, name)
        total += m.ntris()
    return total


def dragon_proxy(sb, level=7):
    """Stand-in for dragon.obj's statue: three noise-displaced icospheres on a plinth,
    3 * 20 * 4^level triangles (level 7: 983 k ~ the Stanford dragon's 871 k), one glossy material."""
    mtl = dict(Ns=100.0, Kd=(0.60, 0.55, 0.30), Ks=(0.40, 0.38, 0.30), map_Kd=None, map_Bump=None)
    sb.register_material(sb.material_from_mtl("dragon", mtl, ""), False)
    rng = np.random.default_rng(871414)
    m = Mesh()
    for (c, r) in (((0.35, 1.55, 0.15), 0.55), ((0.95, 2.05, 0.25), 0.33), ((-0.25, 1.25, -0.1), 0.38)):
        icosphere(m, c, r, level)
        p = m.p[-1]
        v = (p - np.asarray(c, dtype=f32)) / f32(r)
        k = rng.uniform(2.0, 5.0, (3, 3))
        bump = sum(np.sin(v @ k[i] * (i + 2) + i) for i in range(3)) / 3.0
        m.p[-1] = (np.asarray(c, dtype=f32) + v * (r * (1.0 + 0.12 * bump[:, None]))).astype(f32)
    m.emit(sb, "dragon")
    plinth = Mesh()
    box(plinth, (-0.6, -OFF, -0.9), (1.5, 0.7, 1.0), 4)
    plinth.emit(sb, "sp_01_stub_baza")
    return m.ntris() + plinth.ntris()


def sponza_mesh_provider(detail=1.0, dragon_level=7):
    """For Config.build_scene(mesh_provider=...): substitutes the proxies for sponza.obj / dragon.obj."""
    def provide(sb, rel, T=None, forced=""):
        if "sponza" not in rel and "dragon" not in rel:
            raise FileNotFoundError(rel)
        had = dict(sb.mat_by_name)  # JSON materials registered first win (import-materials without override)
        sponza_proxy(sb, detail)
        if "dragon" in rel:
            dragon_proxy(sb, dragon_level)
        for name, idx in had.items():
            sb.mat_by_name[name] = sb.mat_by_name.get(name, idx)
    return provide
