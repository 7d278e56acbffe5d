"""GPU suite (-m gpu): the HIP path, called through the C ABI, against the oracle on the same
seeded inputs; plus size-independent properties at larger sizes.

Bars: integer / index work bit-exact (sampler bits, triangle ids, sample counts, tile lists);
float results within the stated tolerance:
  * closest hit: same triangle except epsilon-band ties, and then t, a, b, c bit-equal;
  * image: sin / cos / acos / asin / atan2 are pinned (include/rgk_libm.h: same bits on CPU and GPU) and every other
    float operation keeps the oracle's order without contraction, so images are bit-identical pixel for pixel except where
    one path meets an epsilon-band tie the kd-tree and the BVH resolve differently, or where splats are float atomics
    (reverse > 0).  Gate: per-pixel ||d||2 <= 1e-3 ||ref||2 for >= 99.9 % of pixels and whole-image relative
    L2 <= 1e-3 -- held on every configuration incl. the benchmark workload at full size; the few-spp small-size proxies
    of the unclamped Sponza configs allow 5e-3 (one tie path is most of a pixel there).  Every test records what it
    measured (conftest.record_parity -> the table at the end of the run, gpurun_out/parity_measured.txt).
"""
import ctypes as C
import os

import numpy as np
import pytest

from rgk_amd import capi
from rgk_amd.config import make_camera, make_params
from rgk_amd.scene import SceneBuilder

from conftest import ROOT, make_rays, record_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rd(product_lib):
    from rgk_amd import render_driver
    assert product_lib.rgk_device_count() >= 1, "no HIP device: the product path has no fallback"
    return render_driver


def both(rd, oracle, wl):
    desc = wl.builder.to_desc()
    return rd.Scene(desc), oracle.OracleScene(desc)


def image_metrics(img, ref, name=None):
    """(whole-image relative L2, fraction of pixels with ||delta|| <= max(1e-3 ||ref||, 1e-6)).  Stricter than SURVEY 8(d)'s
    per-pixel gate, which also allows 4 clamp / S absolute -- meaningless at the default clamp of 1e7, so it is not used."""
    d = np.linalg.norm(img - ref, axis=2)
    r = np.linalg.norm(ref, axis=2)
    within = d <= np.maximum(1e-3 * r, 1e-6)
    rel, frac = float(np.linalg.norm(img - ref) / np.linalg.norm(ref)), float(within.mean())
    if name:
        record_parity(name, rel_l2=rel, within_1e3=frac, bit_identical=float((d == 0).mean()), size=f"{img.shape[1]}x{img.shape[0]}")
    return rel, frac


# ----------------------------------------------------------------------- K0 sampler
def test_sampler_bit_exact(rd, oracle):
    rng = np.random.default_rng(0)
    n = 100000
    seed = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    idx = np.concatenate([rng.integers(0, 4096, n // 2), rng.integers(0, 2 ** 32, n // 2, dtype=np.uint64)]).astype(np.uint32)
    dim = rng.integers(0, 80, n).astype(np.uint32)
    for is2d in (0, 1):
        a, b = rd.sampler_eval(seed, idx, dim, is2d), oracle.sampler_eval(seed, idx, dim, is2d)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_sampler_matches_reference_golden_vectors(rd):
    """Device Halton (zero rotation impossible -> compare through the oracle-free identity:
    u_cp(seed) - u_cp(seed') is index-independent) is covered by test_sampler_bit_exact + the CPU
    golden test; here: the device reproduces the raw golden values once the rotation is removed."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "halton_faure.npz"))
    idx, ref = z["index"], z["values"]
    # logical 2-D dim k uses Halton dims (3k, 3k+1); 1-D dim k uses 3k+2  (k < 64)
    seed = np.zeros(len(idx), np.uint32)
    for k in (0, 1, 7, 40, 63):
        u2 = rd.sampler_eval(seed, idx, np.full(len(idx), k, np.uint32), 1)
        u1 = rd.sampler_eval(seed, idx, np.full(len(idx), k, np.uint32), 0)[:, 0]
        for got, hd in ((u2[:, 0], 3 * k), (u2[:, 1], 3 * k + 1), (u1, 3 * k + 2)):
            rot = np.mod(got.astype(np.float64) - ref[hd].astype(np.float64), 1.0)
            rot = np.where(rot > 0.5, rot - 1.0, rot) if np.ptp(rot) > 0.5 else rot
            assert np.ptp(rot) < 3e-7, (k, hd)     # a constant toroidal shift of the golden sequence


# ----------------------------------------------------------------------- K2 / K5 traversal
def random_rays(rng, lo, hi, n):
    o = (lo + (hi - lo) * rng.uniform(0.02, 0.98, (n, 3))).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d


def check_closest(g, o, rays, eps, ignore=None, max_unexplained=2e-5, name=None):
    """Hits must agree in the triangle, and then bit for bit in t and the barycentrics.  Where the triangle differs both
    accelerators must have found hits whose distances tie within 2 epsilon (H3: the oracle's kd-tree and the BVH take
    exact / epsilon-band ties in different orders); anything else is `unexplained` and bounded by max_unexplained.  With
    `name` the measured numbers go on record, and the first unexplained cases are dumped with both hits."""
    hg, _ = g.trace_closest(rays, ignore)
    ho, _ = o.trace_closest(rays, ignore)
    same = hg["tri"] == ho["tri"]
    for k in ("t", "a", "b", "c"):
        assert np.array_equal(hg[k][same].view(np.uint32), ho[k][same].view(np.uint32)), k
    bad = ~same
    both_hit = bad & (hg["tri"] >= 0) & (ho["tri"] >= 0)
    with np.errstate(invalid="ignore"):
        tie = both_hit & (np.abs(hg["t"] - ho["t"]) <= 2 * eps)
    unexplained = bad.sum() - tie.sum()
    if name:
        record_parity(name, rays=len(rays), same_triangle=float(same.mean()), eps_band_ties=int(tie.sum()), unexplained=int(unexplained))
        for i in np.where(bad & ~tie)[:1][0][:20]:
            print(f"[{name}] unexplained ray {i}: o={rays[i, :3]} d={rays[i, 3:6]} gpu tri {hg['tri'][i]} t {hg['t'][i]!r} | oracle tri {ho['tri'][i]} t {ho['t'][i]!r} | eps {eps}")
    assert unexplained <= max(1, max_unexplained * len(rays)), (int(bad.sum()), int(tie.sum()))
    return float(same.mean())


def test_closest_hit_cornell_bit_exact(rd, oracle, cornell):
    g, o = both(rd, oracle, cornell)
    rng = np.random.default_rng(1)
    i = o.info()
    lo, hi = np.array(list(i.bbox_min)), np.array(list(i.bbox_max))
    oo, dd = random_rays(rng, lo, hi, 300000)
    frac = check_closest(g, o, make_rays(oo, dd), i.epsilon, max_unexplained=0, name="test_closest_hit_cornell_bit_exact")
    assert frac == 1.0
    # rays from outside the box, axis-parallel rays (inf reciprocals, Q11), near/far windows
    axis = np.zeros((6000, 3), np.float32); axis[np.arange(6000), np.arange(6000) % 3] = np.where(np.arange(6000) % 2, 1, -1)
    o2 = (lo + (hi - lo) * rng.uniform(-0.5, 1.5, (6000, 3))).astype(np.float32)
    assert check_closest(g, o, make_rays(o2, axis), i.epsilon) > 0.999
    near = rng.uniform(0, 1, (20000, 1)).astype(np.float32); far = near + rng.uniform(0, 2, (20000, 1)).astype(np.float32)
    rays = np.concatenate([oo[:20000], dd[:20000], near, far], axis=1)
    assert check_closest(g, o, rays, i.epsilon) > 0.999


def test_closest_hit_ignore_triangle(rd, oracle, cornell):
    g, o = both(rd, oracle, cornell)
    rng = np.random.default_rng(2)
    i = o.info()
    oo, dd = random_rays(rng, np.array(list(i.bbox_min)), np.array(list(i.bbox_max)), 50000)
    rays = make_rays(oo, dd)
    first, _ = o.trace_closest(rays)
    assert check_closest(g, o, rays, i.epsilon, ignore=first["tri"].astype(np.int32)) > 0.998   # rest: epsilon-band ties (H3): the lights lie IN the ceiling
    hg, _ = g.trace_closest(rays, first["tri"].astype(np.int32))
    hit = first["tri"] >= 0
    assert (hg["tri"][hit] != first["tri"][hit]).all()


def test_closest_hit_and_visibility_sponza_proxy(rd, oracle, sponza_small):
    g, o = both(rd, oracle, sponza_small)
    rng = np.random.default_rng(3)
    i = o.info()
    lo, hi = np.array(list(i.bbox_min)), np.array(list(i.bbox_max))
    oo, dd = random_rays(rng, lo, hi, 300000)
    frac = check_closest(g, o, make_rays(oo, dd), i.epsilon, max_unexplained=5e-5, name="test_closest_hit_and_visibility_sponza_proxy")
    assert frac > 0.9999       # no coincident surfaces left in the proxy (rgk_amd/proxy.py OFF): what remains are epsilon-band ties at edges
    a = (lo + (hi - lo) * rng.uniform(0.02, 0.98, (200000, 3))).astype(np.float32)
    b = (lo + (hi - lo) * rng.uniform(0.02, 0.98, (200000, 3))).astype(np.float32)
    vg, _ = g.visibility(a, b); vo, _ = o.visibility(a, b)
    record_parity("test_closest_hit_and_visibility_sponza_proxy:visibility", pairs=len(a), agree=float((vg == vo).mean()))
    assert (vg == vo).mean() > 0.9999
    assert g.info().epsilon == o.info().epsilon and list(g.info().bbox_min) == list(o.info().bbox_min)


def test_triangle_soup_with_degenerates(rd, oracle):
    """Random soup incl. zero-area and NaN-normal triangles (never hit: primitives.cpp:90)."""
    rng = np.random.default_rng(4)
    n = 3000
    c = rng.uniform(-5, 5, (n, 1, 3)); tri = (c + rng.normal(scale=0.4, size=(n, 3, 3))).astype(np.float32)
    tri[::97, 2] = tri[::97, 1]              # degenerate: two equal vertices
    sb = SceneBuilder(); sb.register_material(sb.new_material("m", capi.BXDF_DIFFUSE))
    pos = tri.reshape(-1, 3)
    nrm = np.tile([0, 1, 0], (len(pos), 1)).astype(np.float32)
    sb.add_mesh(pos, nrm, np.zeros((len(pos), 2), np.float32), np.tile([1, 0, 0], (len(pos), 1)).astype(np.float32),
                np.arange(len(pos)).reshape(-1, 3), 0)
    desc = sb.to_desc()
    g, o = rd.Scene(desc), oracle.OracleScene(desc)
    oo, dd = random_rays(rng, np.full(3, -6.0), np.full(3, 6.0), 200000)
    assert check_closest(g, o, make_rays(oo, dd), o.info().epsilon, max_unexplained=5e-5) > 0.999


# ----------------------------------------------------------------------- scenes the reference ships complete
REFERENCE_SCENES = {  # name -> (scale, spp, image rel-L2 bound): fixtures made by tools/make_fixtures.py
    "rubiks-bump": (0.15, 16, 1e-3),          # PNG texture + bump map, point light, bumpscale 15
    "cube3": (0.12, 16, 1e-6),                # 8966 faces, LTC Beckmann, sphere light size 0.4, russian 0.6
    "box6": (0.1, 16, 1e-5),                  # 17 k triangles with uv, emissive triangles, reverse = 3 (splats: float atomics)
    "cornell-box-spheres": (0.12, 16, 1e-3),  # LTC Beckmann + dielectric spheres (may_leak), areal lights
}


def scene_fixture(name, **kw):
    from rgk_amd.workloads import SceneFixture
    return SceneFixture(os.path.join(ROOT, "tests", "golden", "scene_%s.npz" % name), **kw)


@pytest.mark.parametrize("name", sorted(REFERENCE_SCENES))
def test_reference_scene_closest_hit(rd, oracle, name):
    wl = scene_fixture(name)
    g, o = both(rd, oracle, wl)
    i = o.info()
    lo, hi = np.array(list(i.bbox_min)), np.array(list(i.bbox_max))
    oo, dd = random_rays(np.random.default_rng(7), lo, hi, 200000)
    # different triangle only inside the epsilon tie band (cube3 has coincident faces: ~1 % of rays tie), never unexplained
    assert check_closest(g, o, make_rays(oo, dd), i.epsilon, max_unexplained=5e-5, name="test_reference_scene_closest_hit:" + name) > 0.98


@pytest.mark.parametrize("name", sorted(REFERENCE_SCENES))
def test_reference_scene_image_parity(rd, oracle, name):
    scale, spp, bound = REFERENCE_SCENES[name]
    wl = scene_fixture(name, scale=scale, spp=spp)
    img, ref, kg, ko = render_both(rd, oracle, wl)
    rel, within = image_metrics(img, ref, "test_reference_scene_image_parity" + ":" + name)
    assert np.isfinite(img).all() and ref.max() > 0
    assert rel <= bound, (name, rel, within)
    assert abs(int(kg.path_rays) - int(ko.path_rays)) <= 2e-3 * ko.path_rays


# ----------------------------------------------------------------------- whole path: images
def render_both(rd, oracle, wl, prm=None, g=None, o=None):
    if g is None:
        g, o = both(rd, oracle, wl)
    prm = prm or wl.params()
    ag, cg, kg = g.render_round(wl.camera, prm, rd.generate_task_list(wl.xres, wl.yres))
    ao, co, ko = o.render_round(wl.camera, prm, oracle.generate_task_list(wl.xres, wl.yres))
    assert np.array_equal(cg, co)                      # sample counts: exact
    return ag / cg[..., None], ao / co[..., None], kg, ko


def test_cornell_image_parity(rd, oracle):
    from rgk_amd.workloads import Workload
    wl = Workload("cornell-256", scale=0.5, spp=16)     # BASELINE configs[0] at half size
    img, ref, kg, ko = render_both(rd, oracle, wl)
    rel, within = image_metrics(img, ref, "test_cornell_image_parity")
    assert rel <= 1e-3 and within >= 0.999, (rel, within)
    assert kg.paths == ko.paths == wl.xres * wl.yres * wl.multisample
    assert abs(int(kg.path_rays) - int(ko.path_rays)) <= 1e-4 * ko.path_rays      # reference ray-count semantics
    assert kg.shadow_rays <= ko.shadow_rays                                        # zero-radiance shadow rays are skipped


def test_slot_order_does_not_change_the_image(rd, oracle):
    """The order in which (pixel, sample) pairs occupy path slots is free (PassParams::gshift): 1, 4, 8 or 64 samples of a
    pixel side by side, sample counts that are no multiple of the group (fallback to a smaller group), and passes split over
    pixels and samples all give the same bits, and the oracle's image."""
    from rgk_amd.workloads import Workload
    for spp in (12, 7, 64):
        wl = Workload("cornell-256", scale=0.25, spp=spp)
        g = rd.Scene(wl.builder.to_desc())
        tiles = rd.generate_task_list(wl.xres, wl.yres)
        imgs = {}
        for grp, batch in (("0", None), ("2", None), ("3", None), ("6", None), ("3", 20000)):
            g.set_tuning(sample_group=int(grp), batch_paths=batch or 0)     # per-scene switches (rgk_scene_set_tuning): no environment involved
            imgs[(grp, batch)], cnt, _ = g.render_round(wl.camera, wl.params(), tiles)
            assert (cnt == spp).all()
        base = imgs[("0", None)]
        assert all(np.array_equal(base, v) for v in imgs.values()), spp
        if spp == 12:
            o = oracle.OracleScene(wl.builder.to_desc())
            ref, _, _ = o.render_round(wl.camera, wl.params(), tiles)
            image_metrics(base, ref, "test_slot_order_does_not_change_the_image")
            assert np.linalg.norm(base - ref) / np.linalg.norm(ref) <= 1e-3


def test_entry_nodes_do_not_change_the_image(rd):
    """Camera rays start at their pixel group's entry nodes (from a frame's second round on: lists capped behind the group's first
    hits, with a retry from the root for a ray that finds nothing within the cap) and, in a single-light scene, the first
    vertex's shadow rays at the group's light-side entry nodes (k_entry_points, k_entry_points_light): all of it must be
    conservative, i.e. the image with them is the image without them, bit for bit -- point light (sponza proxy), sphere light
    (sponza4 proxy), ragged tiles."""
    from rgk_amd.workloads import Workload
    for name, kw in (("sponza-1080p", dict(scale=0.15, spp=16)), ("sponza4-2160p", dict(scale=0.05, spp=16)), ("sponza-1080p", dict(scale=0.0371, spp=64))):
        wl = Workload(name, **kw)
        g = rd.Scene(wl.builder.to_desc())
        tiles = rd.generate_task_list(wl.xres, wl.yres)
        imgs = []
        # ... and the walker of a frame's first round: one lane per pixel for its 8 samples (k_trace_camera_beam, beam = 1) or one lane per ray
        for cam_entry, cap, light_entry, beam in ((0, 0, 0, 0), (0, 0, 0, 1), (1, 0, 0, 1), (1, 0, 0, 0), (1, 1, 0, 1), (1, 1, 1, 1), (1, 1, 1, 0)):
            gg = rd.Scene(wl.builder.to_desc()).set_tuning(entry_points=cam_entry, entry_cap=cap, light_entry=light_entry, beam=beam)  # (a fresh scene per variant: nothing cached from the previous one)
            acc, cnt, k = gg.render_round(wl.camera, wl.params(), tiles)
            acc2, _, k2 = gg.render_round(wl.camera, wl.params(), tiles)   # second round of the frame: lists capped behind the first hits
            assert np.array_equal(acc, acc2) and k.path_rays == k2.path_rays and k.shadow_rays == k2.shadow_rays, (name, kw, cam_entry, cap, light_entry)
            imgs.append((acc, k.path_rays, k.shadow_rays))
        for acc, pr, sr in imgs[1:]:
            assert np.array_equal(acc, imgs[0][0]) and pr == imgs[0][1] and sr == imgs[0][2], (name, kw)
        # other slot orders (a wave then mixes pixel groups: the per-lane path of k_group_trange) and passes split over pixels and samples
        for grp, batch in (("0", None), ("6", None), ("3", 30000)):
            g2 = rd.Scene(wl.builder.to_desc()).set_tuning(sample_group=int(grp), batch_paths=batch or 0)     # (a fresh scene: nothing cached from the runs above)
            acc, cnt, k = g2.render_round(wl.camera, wl.params(), tiles)
            acc2, _, _ = g2.render_round(wl.camera, wl.params(), tiles)      # second round of the frame: cached entry nodes
            assert np.array_equal(acc, imgs[0][0]) and np.array_equal(acc2, imgs[0][0]), (name, kw, grp, batch)


def test_per_frame_caches_follow_the_camera_and_the_tiles(rd):
    """What a scene keeps between rounds (entry nodes of the camera rays and of the first shadow rays, their cap distances) is
    keyed by camera + tile geometry: a scene that has rendered two rounds of one frame and is then asked for another camera, or
    for half of the tiles, must give what a fresh scene gives."""
    from rgk_amd.workloads import Workload
    from rgk_amd.config import camera_from_args
    wl = Workload("sponza-1080p", scale=0.12, spp=16)
    tiles = rd.generate_task_list(wl.xres, wl.yres)
    a = wl.camera.ctor
    cam2 = camera_from_args([a["pos"][0] + 3.0, a["pos"][1] - 1.0, a["pos"][2] + 2.0], a["lookat"], a["up"], a["yview"], a["xview"], wl.xres, wl.yres, a["focus_plane"], a["lens_size"])
    half = (capi.Tile * (len(tiles) // 2))(*tiles[1::2])
    used = rd.Scene(wl.builder.to_desc())
    for _ in range(2):
        used.render_round(wl.camera, wl.params(), tiles)
    for cam, tl in ((cam2, tiles), (wl.camera, half), (cam2, half), (wl.camera, tiles)):
        fresh = rd.Scene(wl.builder.to_desc())
        want, _, kw = fresh.render_round(cam, wl.params(), tl)
        for _ in range(2):      # first round of the new frame (lists rebuilt), second (capped lists)
            got, _, kg = used.render_round(cam, wl.params(), tl)
            assert np.array_equal(got, want) and kg.path_rays == kw.path_rays and kg.shadow_rays == kw.shadow_rays


def test_cornell_against_the_frozen_oracle_image(rd):
    """The committed expected accumulator of BASELINE configs[0] at half resolution (tests/golden/cornell_config0_half.npz,
    rendered by the oracle, generator tools/make_fixtures.py): no oracle code runs in this test."""
    from rgk_amd.workloads import Workload
    z = np.load(os.path.join(ROOT, "tests", "golden", "cornell_config0_half.npz"))
    wl = Workload("cornell-256", scale=0.5)
    g = rd.Scene(wl.builder.to_desc())
    acc, cnt, k = g.render_round(wl.camera, wl.params(), rd.generate_task_list(wl.xres, wl.yres))
    assert np.array_equal(cnt, z["count"]) and k.paths == int(z["counters"][0])
    assert abs(int(k.path_rays) - int(z["counters"][1])) <= 1e-4 * int(z["counters"][1])
    assert np.linalg.norm(acc - z["accum"]) / np.linalg.norm(z["accum"]) <= 1e-3


def test_sponza_proxy_image_parity(rd, oracle, sponza_small):
    """LTC-GGX + diffuse, bilinear textures, bump mapping, point light, constant sky."""
    img, ref, kg, ko = render_both(rd, oracle, sponza_small)
    rel, within = image_metrics(img, ref, "test_sponza_proxy_image_parity")
    # 99.99 % of the pixels are bit-identical (pinned libm); what differs is one or two paths that meet an epsilon-band tie the
    # kd-tree and the BVH resolve differently -- at 8 spp and the default clamp of 1e7 a single such path is most of a pixel and
    # 2e-3 of the whole image's L2 (the full-size test holds the SURVEY 8(d) gate: 2.2e-4 at 256 spp)
    assert rel <= 5e-3 and within >= 0.999, (rel, within)
    assert abs(int(kg.path_rays) - int(ko.path_rays)) <= 1e-4 * ko.path_rays
    assert np.isfinite(img).all()
    # (clamping does not help: with clamp 5 the same 2.089e-3 -- the tie path carries an ordinary value, it is just an eighth of
    # its pixel.)  The same frame at 64 spp: the path is a 64th, and SURVEY 8(d)'s 1e-3 holds at this size too
    from rgk_amd.workloads import Workload
    wl64 = Workload("sponza-1080p", scale=0.1, spp=64)
    img, ref, kg, ko = render_both(rd, oracle, wl64)
    rel, within = image_metrics(img, ref, "test_sponza_proxy_image_parity:64spp")
    assert rel <= 1e-3 and within >= 0.999, (rel, within)


def test_sponza4_sphere_light_clamp_russian(rd, oracle):
    from rgk_amd.workloads import Workload
    wl = Workload("sponza4-2160p", scale=0.04, spp=8)   # sphere light size 1, depth 4, clamp 5, russian 0.6
    img, ref, kg, ko = render_both(rd, oracle, wl)
    rel, within = image_metrics(img, ref, "test_sponza4_sphere_light_clamp_russian")
    assert rel <= 5e-3 and within >= 0.999, (rel, within)    # measured 2.6e-3 / 0.9998 (8 spp: see test_sponza_proxy_image_parity)
    assert img.max() <= wl.clamp * (1 + 1e-6)
    # 64 spp of the same frame: one tie path is an eighth of what it was, and SURVEY 8(d)'s 1e-3 holds (clamp 5 is this config's own)
    wl64 = Workload("sponza4-2160p", scale=0.04, spp=64)
    img, ref, kg, ko = render_both(rd, oracle, wl64)
    rel, within = image_metrics(img, ref, "test_sponza4_sphere_light_clamp_russian:64spp")
    d = np.linalg.norm(img - ref, axis=2); r = np.linalg.norm(ref, axis=2)
    gate = float((d <= np.maximum(1e-3 * r, 4.0 * wl64.clamp / 64)).mean())   # SURVEY 8(d)'s per-pixel gate, clamp term included (clamp 5 here)
    assert rel <= 1e-3 and gate >= 0.999 and within >= 0.998, (rel, gate, within)   # measured 8.3e-4 / 1.0 / 0.9989 (depth 4: 14 of 13 158 pixels hold a path that went another way)


def test_dragon_sponza_config4_small(rd, oracle):
    """BASELINE configs[3] at small size: sphere light, reverse = 3, depth 40, clamp 5, russian 0.7, JSON
    material override (exponent 800 LTC) over the proxy geometry + statue stand-in."""
    from rgk_amd.workloads import Workload
    wl = Workload("dragon-sponza-1080p", scale=0.05, spp=8, dragon_level=3)
    assert wl.reverse == 3 and wl.depth == 40
    img, ref, kg, ko = render_both(rd, oracle, wl)
    rel, within = image_metrics(img, ref, "test_dragon_sponza_config4_small")
    assert rel <= 1e-3 and within >= 0.999, (rel, within)    # measured 1.3e-4 / 0.9998 (reverse = 3: splats are float atomics)
    assert abs(int(kg.path_rays) - int(ko.path_rays)) <= 1e-4 * ko.path_rays


def test_deep_tree_stack_overflow_variant(rd, oracle):
    """A tree that needs more traversal-stack entries than the 32 kept in LDS: the rest lives per lane in global memory
    (RGK_STACK_OVF=1 forces that variant, which otherwise only serves trees needing more than 64 entries)."""
    from rgk_amd.workloads import Workload
    wl = Workload("dragon-sponza-1080p", scale=0.03, spp=4, dragon_level=5)
    os.environ["RGK_STACK_OVF"] = "1"
    try:
        g = rd.Scene(wl.builder.to_desc())
    finally:
        del os.environ["RGK_STACK_OVF"]
    o = oracle.OracleScene(wl.builder.to_desc())
    gi = g.info()
    assert gi.max_depth >= 11, gi.max_depth       # 3 pushes per level: more than 32 entries possible
    lo, hi = np.array(list(gi.bbox_min)), np.array(list(gi.bbox_max))
    oo, dd = random_rays(np.random.default_rng(9), lo, hi, 200000)
    assert check_closest(g, o, make_rays(oo, dd), gi.epsilon, max_unexplained=5e-5) > 0.99
    vg, _ = g.visibility(oo[:50000], oo[50000:100000])
    vo, _ = o.visibility(oo[:50000], oo[50000:100000])
    assert (vg != vo).mean() < 2e-3


def test_envmap_sky_float_texture(rd, oracle):
    """Scene::GetSkyboxRay in envmap mode (scene.cpp:748-763): lat-long lookup with rotation into a float (HDR-style)
    texture -- the kind BASELINE configs[3] names but the reference checkout does not ship."""
    rng = np.random.default_rng(11)
    sb = SceneBuilder()
    m = sb.new_material("grey", capi.BXDF_LTC_GGX_DIFFUSE)
    m["tex_diffuse"] = sb.create_solid_texture((0.5, 0.5, 0.5)); m["tex_color"] = sb.create_solid_texture((0.3, 0.3, 0.3)); m["roughness"] = 0.4
    sb.register_material(m)
    from rgk_amd.scene import glm_mat4_mul, glm_scale, glm_translate
    sb.add_primitive("plane", glm_scale((4, 1, 4)), "grey")
    sb.add_primitive("cube", glm_mat4_mul(glm_translate((0, 0.5, 0)), glm_scale((0.5, 0.5, 0.5))), "grey")
    hdr = (rng.random((64, 128, 3)) ** 4 * 6).astype(np.float32)           # a few bright spots
    hdr[20:24, 30:36] = (40.0, 35.0, 30.0)
    tex = sb.add_image_texture("sky", hdr)
    sb.sky.update(mode=capi.SKY_ENVMAP, tex=tex, intensity=0.9, rotate=37.0)
    W, H = 160, 120
    cam = make_camera((2.5, 1.8, 3.0), (0, 0.4, 0), (0, 1, 0), fov=50, xres=W, yres=H)
    prm = make_params(W, H, 16, 4, clamp=50.0, russian=0.8)
    desc = sb.to_desc()
    g, o = rd.Scene(desc), oracle.OracleScene(desc)
    ag, cg, _ = g.render_round(cam, prm, rd.generate_task_list(W, H))
    ao, co, _ = o.render_round(cam, prm, oracle.generate_task_list(W, H))
    img, ref = ag / cg[..., None], ao / co[..., None]
    rel, within = image_metrics(img, ref, "test_envmap_sky_float_texture")
    assert ref.max() > 1.0 and rel <= 2e-3, (rel, within)   # libm atan2f / asinf differences only


def material_zoo():
    """Cornell-like box exercising mirror, dielectric, transparent, mix, ltc_beckmann, no-russian, thin lens."""
    sb = SceneBuilder()

    def mat(name, kind, **kw):
        m = sb.new_material(name, kind)
        for k, v in kw.items():
            m[k] = sb.create_solid_texture(v) if k.startswith("tex_") else v
        return sb.register_material(m)
    mat("white", capi.BXDF_DIFFUSE, tex_diffuse=(0.7, 0.7, 0.7))
    mat("red", capi.BXDF_DIFFUSE, tex_diffuse=(0.6, 0.1, 0.1))
    mat("light", capi.BXDF_DIFFUSE, tex_diffuse=(0.5, 0.5, 0.5), emission=(12.0, 12.0, 10.0))
    mat("mirror", capi.BXDF_MIRROR, tex_color=(0.9, 0.9, 0.9))
    mat("glass", capi.BXDF_DIELECTRIC, tex_color=(1.0, 1.0, 1.0), ior=1.5, flags=capi.MAT_NO_RUSSIAN)
    mat("ghost", capi.BXDF_TRANSPARENT)
    mat("bek", capi.BXDF_LTC_BECKMANN, tex_color=(0.8, 0.6, 0.2), roughness=0.3)
    mat("ggxd", capi.BXDF_LTC_GGX_DIFFUSE, tex_color=(0.3, 0.3, 0.3), tex_diffuse=(0.2, 0.4, 0.6), roughness=0.15)
    mat("mix", capi.BXDF_MIX, mix_m1=sb.material_index("red"), mix_m2=sb.material_index("bek"), amount=0.4)

    def T(scale, translate, rot=None):
        from rgk_amd.scene import glm_mat4_mul, glm_rotate, glm_scale, glm_translate
        m = glm_scale(scale)
        if rot:
            m = glm_mat4_mul(glm_rotate(rot[0], rot[1]), m)
        return glm_mat4_mul(glm_translate(translate), m)
    sb.add_primitive("plane", T((2, 1, 2), (0, 0, 0)), "white")
    sb.add_primitive("plane", T((2, 1, 2), (0, 3, 0), (np.pi, (1, 0, 0))), "white")
    sb.add_primitive("plane", T((2, 1, 2), (0, 1.5, -2), (np.pi / 2, (1, 0, 0))), "ggxd")
    sb.add_primitive("plane", T((2, 1, 2), (-2, 1.5, 0), (-np.pi / 2, (0, 0, 1))), "red")
    sb.add_primitive("plane", T((2, 1, 2), (2, 1.5, 0), (np.pi / 2, (0, 0, 1))), "mirror")
    sb.add_primitive("plane", T((0.5, 1, 0.5), (0, 2.98, 0), (np.pi, (1, 0, 0))), "light")
    sb.add_primitive("cube", T((0.8, 0.8, 0.8), (-0.9, 0.4, -0.5), (0.4, (0, 1, 0))), "glass")
    sb.add_primitive("cube", T((0.7, 1.4, 0.7), (0.8, 0.7, -0.8), (-0.3, (0, 1, 0))), "mix")
    sb.add_primitive("cube", T((0.5, 0.5, 0.5), (0.2, 0.25, 0.9)), "bek")
    sb.add_primitive("plane", T((0.4, 1, 0.4), (-0.2, 1.2, 0.6), (np.pi / 2, (1, 0, 0))), "ghost")
    return sb


def test_material_zoo_image_parity(rd, oracle):
    sb = material_zoo()
    desc = sb.to_desc()
    g, o = rd.Scene(desc), oracle.OracleScene(desc)
    W, H, S = 96, 72, 32
    for lens in (0.0, 0.08):
        cam = make_camera((0, 1.5, 5.5), (0, 1.3, 0), (0, 1, 0), fov=45, xres=W, yres=H, focus_plane=5.0, lens_size=lens)
        prm = make_params(W, H, S, 8, clamp=30.0, russian=0.7)
        ag, cg, kg = g.render_round(cam, prm, rd.generate_task_list(W, H))
        ao, co, ko = o.render_round(cam, prm, oracle.generate_task_list(W, H))
        img, ref = ag / cg[..., None], ao / co[..., None]
        rel, within = image_metrics(img, ref, "test_material_zoo_image_parity")
        assert rel <= 1e-4 and within >= 0.999, (lens, rel, within)   # measured 1.2e-5 / 0.9999
        assert abs(int(kg.path_rays) - int(ko.path_rays)) <= 1e-4 * ko.path_rays


def test_rounds_are_repeatable_bit_for_bit(rd):
    """round 2 saw LTC results change from run to run with one form of the pinned sqrt and filed it as a compiler
    finding; a fixed binary on fixed input cannot do that -- it was a read of something not yet written (the form does NOT
    reproduce it on today's kernels, with or without RGK_POISON: tools/gpu_zoo_debug.py on a -DRGK_LIBM_BUILTIN_SQRT build).
    Whatever the cause was, this is the guard: the material zoo (every BxDF kind, textures, bump, lens) rendered twice at depth 1
    and 8 by one scene and once by a fresh scene gives the same bits and the same ray counts."""
    sb = material_zoo()
    W, H, S = 96, 72, 32
    cam = make_camera((0, 1.5, 5.5), (0, 1.3, 0), (0, 1, 0), fov=45, xres=W, yres=H, focus_plane=5.0, lens_size=0.0)
    g = rd.Scene(sb.to_desc())
    for depth in (1, 8):
        prm = make_params(W, H, S, depth, clamp=30.0, russian=0.7)
        tiles = rd.generate_task_list(W, H)
        a1, c1, k1 = g.render_round(cam, prm, tiles)
        a2, c2, k2 = g.render_round(cam, prm, tiles)
        a3, c3, k3 = rd.Scene(sb.to_desc()).render_round(cam, prm, tiles)
        assert np.array_equal(a1, a2) and np.array_equal(a1, a3), depth
        assert (k1.path_rays, k1.shadow_rays) == (k2.path_rays, k2.shadow_rays) == (k3.path_rays, k3.shadow_rays)


def test_bidirectional_reverse_parity(rd, oracle):
    """reverse > 0 (BASELINE configs[3] feature): light sub-path, light-tracing splats (count 0, may land on
    any pixel), connections of every camera vertex to every light vertex (path_tracer.cpp:336-398,463-480)."""
    from rgk_amd.workloads import Workload
    wl = Workload("cornell-256", scale=0.25, spp=16)
    g, o = both(rd, oracle, wl)
    for reverse, depth in ((1, 3), (3, 5)):
        prm = make_params(wl.xres, wl.yres, wl.multisample, depth, clamp=20.0, russian=0.7, reverse=reverse)
        ag, cg, kg = g.render_round(wl.camera, prm, rd.generate_task_list(wl.xres, wl.yres))
        ao, co, ko = o.render_round(wl.camera, prm, oracle.generate_task_list(wl.xres, wl.yres))
        assert np.array_equal(cg, co)                               # splats add radiance with count 0
        rel = np.linalg.norm(ag - ao) / np.linalg.norm(ao)
        assert rel <= 2e-3, (reverse, rel)                          # float atomics reorder the splat sums
        assert abs(int(kg.path_rays) - int(ko.path_rays)) <= 1e-3 * ko.path_rays   # camera + light sub-path rays
    # zoo: delta + LTC materials on both sub-paths, thin lens (camera position per sample)
    sb = material_zoo()
    desc = sb.to_desc()
    g2, o2 = rd.Scene(desc), oracle.OracleScene(desc)
    W, H, S = 64, 48, 32
    cam = make_camera((0, 1.5, 5.5), (0, 1.3, 0), (0, 1, 0), fov=45, xres=W, yres=H, focus_plane=5.0, lens_size=0.05)
    prm = make_params(W, H, S, 6, clamp=30.0, russian=0.7, reverse=2)
    ag, cg, kg = g2.render_round(cam, prm, rd.generate_task_list(W, H))
    ao, co, ko = o2.render_round(cam, prm, oracle.generate_task_list(W, H))
    assert np.array_equal(cg, co)
    assert np.linalg.norm(ag - ao) / np.linalg.norm(ao) <= 3e-2
    # no lights at all: reverse has no effect (no light sub-path is built)
    sb0 = SceneBuilder(); m = sb0.new_material("m", capi.BXDF_DIFFUSE); m["tex_diffuse"] = sb0.create_solid_texture((0.5, 0.5, 0.5))
    sb0.register_material(m); sb0.add_primitive("cube", np.eye(4, dtype=np.float32), "m"); sb0.set_skybox_color((0.5, 0.6, 0.7), 1.0)
    g3 = rd.Scene(sb0.to_desc())
    cam3 = make_camera((0, 0.3, 3), (0, 0, 0), (0, 1, 0), fov=40, xres=32, yres=32)
    a0 = g3.render_round(cam3, make_params(32, 32, 4, 3, reverse=0), rd.generate_task_list(32, 32))[0]
    a2 = g3.render_round(cam3, make_params(32, 32, 4, 3, reverse=2), rd.generate_task_list(32, 32))[0]
    assert np.array_equal(a0, a2)


# ----------------------------------------------------------------------- boundary behaviour, properties
FULL_SIZE_SPONZA_REL, FULL_SIZE_SPONZA_WITHIN = 1e-3, 0.999   # SURVEY 8(d)'s gate, on the benchmark workload at full size (measured 2.2e-4 / 0.9993)


def test_cornell_config2_full_size_256spp(rd, oracle):
    """BASELINE configs[1] as quoted: cornell-box 1024 x 1024 x 256 spp, russian 0.75 (268 M paths).  Whole frame: counts,
    range, idempotence, the two-halves deal; 64 tiles of it (16.8 M paths) against the oracle at the full sample count."""
    from rgk_amd.workloads import Workload
    wl = Workload("cornell-1024")
    assert (wl.xres, wl.yres, wl.multisample) == (1024, 1024, 256) and abs(wl.russian - 0.75) < 1e-6
    g = rd.Scene(wl.builder.to_desc())
    prm = wl.params()
    tiles = rd.generate_task_list(wl.xres, wl.yres)
    acc, cnt, k = g.render_round(wl.camera, prm, tiles)
    assert (cnt == 256).all() and k.paths == 1024 * 1024 * 256
    assert np.isfinite(acc).all() and (acc >= 0).all() and acc.max() <= 256 * wl.clamp * (1 + 1e-6)
    a3 = np.zeros_like(acc); c3 = np.zeros_like(cnt)
    ev = (capi.Tile * ((len(tiles) + 1) // 2))(*tiles[0::2]); od = (capi.Tile * (len(tiles) // 2))(*tiles[1::2])
    g.render_round(wl.camera, prm, ev, a3, c3); g.render_round(wl.camera, prm, od, a3, c3)
    assert np.array_equal(acc, a3) and np.array_equal(cnt, c3)
    sub = (capi.Tile * 64)(*tiles[:64])
    o = oracle.OracleScene(wl.builder.to_desc())
    ao = np.zeros_like(acc); co = np.zeros_like(cnt); o.render_round(wl.camera, prm, sub, ao, co)
    m = co > 0
    assert m.sum() == 64 * 1024 and np.array_equal(cnt[m], co[m])
    d = np.linalg.norm(acc[m] - ao[m], axis=1); r = np.linalg.norm(ao[m], axis=1)
    rel = float(np.linalg.norm(acc[m] - ao[m]) / np.linalg.norm(ao[m])); within = float((d <= np.maximum(1e-3 * r, 1e-6)).mean())
    record_parity("test_cornell_config2_full_size_256spp", rel_l2_64_tiles=rel, within_1e3=within, bit_identical=float((d == 0).mean()), size="1024x1024x256")
    assert rel <= 1e-3 and within >= 0.999


def test_round_properties_cornell_config2_size(rd, oracle):
    """BASELINE configs[1] geometry at 1024x1024 (16 spp to stay quick): size-independent properties."""
    from rgk_amd.workloads import Workload
    wl = Workload("cornell-1024", spp=16)
    g = rd.Scene(wl.builder.to_desc())
    prm = wl.params()
    tiles = rd.generate_task_list(wl.xres, wl.yres)
    acc, cnt, k = g.render_round(wl.camera, prm, tiles)
    assert (cnt == 16).all() and k.paths == 1024 * 1024 * 16
    assert np.isfinite(acc).all() and (acc >= 0).all() and acc.max() <= 16 * wl.clamp * (1 + 1e-6)
    # idempotent: same seeds -> same bits; tile-shard invariant: two halves == whole (the multi-GPU deal)
    acc2, cnt2, _ = g.render_round(wl.camera, prm, tiles)
    assert np.array_equal(acc, acc2)
    a3 = np.zeros_like(acc); c3 = np.zeros_like(cnt)
    ev = (capi.Tile * ((len(tiles) + 1) // 2))(*tiles[0::2]); od = (capi.Tile * (len(tiles) // 2))(*tiles[1::2])
    g.render_round(wl.camera, prm, ev, a3, c3); g.render_round(wl.camera, prm, od, a3, c3)
    assert np.array_equal(acc, a3) and np.array_equal(cnt, c3)
    # small batches (many passes over pixels and samples) change nothing
    g.set_tuning(batch_paths=300000)
    a4, c4, _ = g.render_round(wl.camera, prm, tiles)
    g.set_tuning(batch_paths=0)
    assert np.array_equal(acc, a4)
    # the two halves of the pixel list side by side on two streams (an experiment that did not pay; kept switchable): the same bits
    g.set_tuning(two_lanes=1)
    a6, c6, k6 = g.render_round(wl.camera, prm, tiles)
    g.set_tuning(two_lanes=0)
    assert np.array_equal(acc, a6) and np.array_equal(cnt, c6) and (k6.path_rays, k6.shadow_rays) == (k.path_rays, k.shadow_rays)
    # linearity of the accumulator: a second round adds
    tiles_b = rd.generate_task_list(wl.xres, wl.yres, seedcount_base=len(tiles))
    a5 = acc.copy(); c5 = cnt.copy()
    g.render_round(wl.camera, prm, tiles_b, a5, c5)
    assert (c5 == 32).all() and (a5 >= acc).all()
    # energy check against the oracle on a 64-tile sample of the same frame
    sub = (capi.Tile * 64)(*tiles[:64])
    ag = np.zeros_like(acc); cg = np.zeros_like(cnt); g.render_round(wl.camera, prm, sub, ag, cg)
    o = oracle.OracleScene(wl.builder.to_desc())
    ao = np.zeros_like(acc); co = np.zeros_like(cnt); o.render_round(wl.camera, prm, sub, ao, co)
    rel = float(np.linalg.norm(ag - ao) / np.linalg.norm(ao))
    record_parity("test_round_properties_cornell_config2_size", rel_l2_64_tiles=rel, size="1024x1024x16")
    assert np.array_equal(cg, co) and rel <= 1e-3


def test_round_properties_sponza_config3_full_size(rd, oracle):
    """BASELINE configs[2] -- the benchmark workload -- at its full 1920x1080x256 spp: size-independent properties of a
    whole round, and the oracle on a 24-tile sample of the same frame at the same 256 spp."""
    from rgk_amd.workloads import Workload
    wl = Workload("sponza-1080p")
    assert (wl.xres, wl.yres, wl.multisample, wl.depth) == (1920, 1080, 256, 2)
    g = rd.Scene(wl.builder.to_desc())
    prm = wl.params()
    tiles = rd.generate_task_list(wl.xres, wl.yres)
    assert len(tiles) == 2040
    acc, cnt, k = g.render_round(wl.camera, prm, tiles)
    assert (cnt == 256).all() and k.paths == 1920 * 1080 * 256
    assert np.isfinite(acc).all() and (acc >= 0).all() and acc.max() <= 256 * wl.clamp * (1 + 1e-6)
    assert k.paths <= k.path_rays <= 2 * k.paths                         # depth 2: one or two path rays per path
    # tile-shard invariance, the multi-GPU deal (tile i -> rank i mod 2), bit for bit
    a2 = np.zeros_like(acc); c2 = np.zeros_like(cnt)
    ev = (capi.Tile * ((len(tiles) + 1) // 2))(*tiles[0::2]); od = (capi.Tile * (len(tiles) // 2))(*tiles[1::2])
    g.render_round(wl.camera, prm, ev, a2, c2); g.render_round(wl.camera, prm, od, a2, c2)
    assert np.array_equal(acc, a2) and np.array_equal(cnt, c2)
    # the oracle on every 85th tile of the centre-out list
    sub = (capi.Tile * 24)(*tiles[0::85])
    ag = np.zeros_like(acc); cg = np.zeros_like(cnt); g.render_round(wl.camera, prm, sub, ag, cg)
    o = oracle.OracleScene(wl.builder.to_desc())
    ao = np.zeros_like(acc); co = np.zeros_like(cnt); o.render_round(wl.camera, prm, sub, ao, co)
    assert np.array_equal(cg, co)
    m = co > 0
    assert np.array_equal(ag[m], acc[m])                                 # a tile's pixels do not depend on the other tiles
    rel = float(np.linalg.norm(ag - ao) / np.linalg.norm(ao))
    d = np.linalg.norm(ag - ao, axis=2)[m]; r = np.linalg.norm(ao, axis=2)[m]
    within = float((d <= np.maximum(1e-3 * r, 1e-6)).mean())
    record_parity("test_round_properties_sponza_config3_full_size", rel_l2_24_tiles=rel, within_1e3=within, bit_identical=float((d == 0).mean()), size="1920x1080x256")
    assert rel <= FULL_SIZE_SPONZA_REL and within >= FULL_SIZE_SPONZA_WITHIN


def test_round_properties_sponza4_config5_full_size(rd, oracle):
    """BASELINE configs[4] at its full 3840x2160x1024 spp (8.5 G paths, what the 8 GPUs share) on ONE GPU: counts, range, the
    tile deal over 8 ranks done one rank after the other == the whole frame bit for bit, and the oracle on 12 tiles at the full
    1024 spp.  (The sharded form with RCCL needs the 8-GPU node the builder does not have.)"""
    from rgk_amd.workloads import Workload
    wl = Workload("sponza4-2160p")
    assert (wl.xres, wl.yres, wl.multisample, wl.depth) == (3840, 2160, 1024, 4)
    g = rd.Scene(wl.builder.to_desc())
    prm = wl.params()
    tiles = rd.generate_task_list(wl.xres, wl.yres)
    assert len(tiles) == 8160
    acc, cnt, k = g.render_round(wl.camera, prm, tiles)
    assert (cnt == 1024).all() and k.paths == 3840 * 2160 * 1024
    assert np.isfinite(acc).all() and (acc >= 0).all() and acc.max() <= 1024 * wl.clamp * (1 + 1e-6)
    a2 = np.zeros_like(acc); c2 = np.zeros_like(cnt)
    for rank in range(8):
        mine = rd.shard_tiles(tiles, rank, 8)
        assert len(mine) == 1020
        g.render_round(wl.camera, prm, mine, a2, c2)
    assert np.array_equal(acc, a2) and np.array_equal(cnt, c2)
    sub = (capi.Tile * 12)(*tiles[0::680])
    o = oracle.OracleScene(wl.builder.to_desc())
    ao = np.zeros_like(acc); co = np.zeros_like(cnt); o.render_round(wl.camera, prm, sub, ao, co)
    m = co > 0
    assert m.sum() == 12 * 1024 and np.array_equal(cnt[m], co[m])
    d = np.linalg.norm(acc[m] - ao[m], axis=1); r = np.linalg.norm(ao[m], axis=1)
    rel = float(np.linalg.norm(acc[m] - ao[m]) / np.linalg.norm(ao[m])); strict = float((d <= np.maximum(1e-3 * r, 1e-6)).mean())
    # SURVEY 8(d)'s per-pixel gate on the accumulated sums: max(1e-3 |ref|, 4 clamp) -- here the clamp is 5, so the second term
    # means "about two paths of this pixel's 1024 went another way" (one path in ~2e5 takes a different discrete decision
    # somewhere along its up to four vertices, DESIGN.md 4: ~0.5 % of the pixels at this sample count; measured 0.995 strict)
    gate = float((d <= np.maximum(1e-3 * r, 4.0 * wl.clamp)).mean())
    record_parity("test_round_properties_sponza4_config5_full_size", rel_l2_12_tiles=rel, within_1e3_strict=strict, within_survey_gate=gate,
                  bit_identical=float((d == 0).mean()), size="3840x2160x1024")
    assert rel <= 1e-3 and gate >= 0.999 and strict >= 0.99


def test_round_properties_dragon_sponza_config4_full_size(rd, oracle):
    """BASELINE configs[3] at its full 1920x1080x512 spp, reverse 3, depth 40 on the 1.05 M-triangle scene (proxy atrium + statue
    stand-in, the geometry the bench line runs): a whole bidirectional round -- counts, range, the two-halves tile deal -- and the
    oracle on 8 tiles of the same frame at the full 512 spp.  Splats (light-tracing side effects, tracer.cpp:20-26) are float
    atomics on the GPU: their order varies from run to run, so sums that include them agree to float re-association (measured
    ~1e-7 relative), not bit for bit; everything a path adds to its own pixel keeps the oracle's order."""
    from rgk_amd.workloads import Workload
    wl = Workload("dragon-sponza-1080p")
    assert (wl.xres, wl.yres, wl.multisample, wl.depth, wl.reverse) == (1920, 1080, 512, 40, 3)
    wl.builder.finalize()
    assert len(wl.builder.F) > 1000000
    g = rd.Scene(wl.builder.to_desc())
    prm = wl.params()
    tiles = rd.generate_task_list(wl.xres, wl.yres)
    assert len(tiles) == 2040
    acc, cnt, k = g.render_round(wl.camera, prm, tiles)
    assert (cnt == 512).all() and k.paths == 1920 * 1080 * 512          # splats add radiance with count 0
    assert np.isfinite(acc).all() and (acc >= 0).all()
    assert k.path_rays >= k.paths and k.shadow_rays > 0
    # the multi-GPU deal (tile i -> rank i mod 2): the halves add up to the whole; their splats land anywhere in the frame
    a2 = np.zeros_like(acc); c2 = np.zeros_like(cnt)
    ev = (capi.Tile * ((len(tiles) + 1) // 2))(*tiles[0::2]); od = (capi.Tile * (len(tiles) // 2))(*tiles[1::2])
    _, _, k_ev = g.render_round(wl.camera, prm, ev, a2, c2); _, _, k_od = g.render_round(wl.camera, prm, od, a2, c2)
    assert np.array_equal(cnt, c2) and k_ev.path_rays + k_od.path_rays == k.path_rays and k_ev.shadow_rays + k_od.shadow_rays == k.shadow_rays
    rel_halves = float(np.linalg.norm(acc - a2) / np.linalg.norm(acc))
    assert rel_halves <= 1e-5, rel_halves
    # the oracle on every 255th tile of the centre-out list, the GPU on the same 8 tiles alone (a tile's light sub-paths splat
    # into other tiles: both sides must trace the same set of paths for the frames to be comparable)
    sub = (capi.Tile * 8)(*tiles[0::255])
    ag = np.zeros_like(acc); cg = np.zeros_like(cnt); _, _, kg = g.render_round(wl.camera, prm, sub, ag, cg)
    o = oracle.OracleScene(wl.builder.to_desc())
    ao = np.zeros_like(acc); co = np.zeros_like(cnt); _, _, ko = o.render_round(wl.camera, prm, sub, ao, co)
    assert np.array_equal(cg, co) and (co > 0).sum() == 8 * 1024
    assert abs(int(kg.path_rays) - int(ko.path_rays)) <= 1e-4 * ko.path_rays
    # the reference tests Visibility first and evaluates the BxDFs after (path_tracer.cpp:431,466); the kernels evaluate first and
    # do not trace a ray that could only carry zero radiance: fewer shadow rays, the same sums
    assert 0.5 * ko.shadow_rays <= kg.shadow_rays <= ko.shadow_rays
    m = co > 0
    rel = float(np.linalg.norm(ag - ao) / np.linalg.norm(ao))
    rel_tiles = float(np.linalg.norm(ag[m] - ao[m]) / np.linalg.norm(ao[m]))
    d = np.linalg.norm(ag - ao, axis=2)[m]; r = np.linalg.norm(ao, axis=2)[m]
    strict = float((d <= np.maximum(1e-3 * r, 1e-6)).mean())
    gate = float((d <= np.maximum(1e-3 * r, 4.0 * wl.clamp)).mean())    # max(1e-3 |ref|, 4 clamp / S) on the per-sample mean
    record_parity("test_round_properties_dragon_sponza_config4_full_size", rel_l2_frame=rel, rel_l2_8_tiles=rel_tiles, within_1e3_strict=strict,
                  within_survey_gate=gate, halves_vs_whole=rel_halves, bit_identical=float((d == 0).mean()), size="1920x1080x512 reverse 3")
    assert rel <= 1e-3 and rel_tiles <= 1e-3 and gate >= 0.999, (rel, rel_tiles, gate, strict)


def test_edge_cases(rd, oracle, cornell):
    g, o = both(rd, oracle, cornell)
    prm = cornell.params()
    # empty tile list: nothing happens
    acc, cnt, k = g.render_round(cornell.camera, prm, (capi.Tile * 0)())
    assert k.paths == 0 and not acc.any() and not cnt.any()
    # ragged frame (not a multiple of the tile size), 1 spp, depth 1, non-square sample count
    for (w, h, s, d) in ((70, 45, 1, 1), (33, 31, 5, 3), (1, 1, 7, 10)):
        cam = make_camera((0, 1, 6.8), (0, 1, 0), (0, 1, 0), fov=19.5, xres=w, yres=h)
        p = make_params(w, h, s, d, clamp=20.0, russian=0.74)
        ag, cg, _ = g.render_round(cam, p, rd.generate_task_list(w, h))
        ao, co, _ = o.render_round(cam, p, oracle.generate_task_list(w, h))
        assert np.array_equal(cg, co) and (cg == s).all()
        assert np.allclose(ag, ao, rtol=2e-3, atol=1e-4)
    # russian < 0 (no roulette, RTC default) and russian == 0
    for r in (-1.0, 0.0):
        p = make_params(cornell.xres, cornell.yres, 4, 4, clamp=20.0, russian=r)
        ag, cg, kg = g.render_round(cornell.camera, p, rd.generate_task_list(cornell.xres, cornell.yres))
        ao, co, ko = o.render_round(cornell.camera, p, oracle.generate_task_list(cornell.xres, cornell.yres))
        assert np.linalg.norm(ag - ao) / np.linalg.norm(ao) <= 1e-3 and kg.path_rays == ko.path_rays


def test_error_codes(rd, product_lib, cornell):
    g = rd.Scene(cornell.builder.to_desc())
    prm = cornell.params()
    tiles = rd.generate_task_list(cornell.xres, cornell.yres)
    acc = np.zeros((cornell.yres, cornell.xres, 3), np.float32); cnt = np.zeros((cornell.yres, cornell.xres), np.uint32)
    bad = (capi.Tile * 1)(); bad[0].x0, bad[0].x1, bad[0].y0, bad[0].y1 = 0, cornell.xres + 5, 0, 8
    assert product_lib.rgk_render_round(g.h, C.byref(cornell.camera), C.byref(prm), bad, 1, acc.ctypes.data, cnt.ctypes.data, None) == -1
    assert b"outside the frame" in product_lib.rgk_last_error()
    p2 = cornell.params(); p2.reverse = 9          # more light-path vertices than the kernels keep per slot
    assert product_lib.rgk_render_round(g.h, C.byref(cornell.camera), C.byref(p2), tiles, len(tiles), acc.ctypes.data, cnt.ctypes.data, None) == -5
    p2b = cornell.params(); p2b.sampler = capi.SAMPLER_STRATIFIED   # mt19937 tables: oracle only
    assert product_lib.rgk_render_round(g.h, C.byref(cornell.camera), C.byref(p2b), tiles, len(tiles), acc.ctypes.data, cnt.ctypes.data, None) == -5
    p3 = cornell.params(); p3.multisample = 0
    assert product_lib.rgk_render_round(g.h, C.byref(cornell.camera), C.byref(p3), tiles, len(tiles), acc.ctypes.data, cnt.ctypes.data, None) == -1
    assert not acc.any()
    h = C.c_void_p()
    d = cornell.builder.to_desc()
    assert product_lib.rgk_scene_create(C.byref(d), 99, C.byref(h)) == -1        # device out of range
    sb = SceneBuilder.load_npz(os.path.join(ROOT, "rgk_amd", "data", "cornell_scene.npz"))
    sb.FM = sb.FM.copy(); sb.tri_mat = [np.full_like(sb.FM, 77)]
    assert product_lib.rgk_scene_create(C.byref(sb.to_desc()), 0, C.byref(h)) == -1
    assert b"material index" in product_lib.rgk_last_error()


def test_device_accumulator_entry_point(rd, oracle, cornell):
    import torch
    g, o = both(rd, oracle, cornell)
    prm = cornell.params()
    tiles = rd.generate_task_list(cornell.xres, cornell.yres)
    acc = torch.zeros((cornell.yres, cornell.xres, 3), dtype=torch.float32, device="cuda:0")
    cnt = torch.zeros((cornell.yres, cornell.xres), dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    g.render_round_device(cornell.camera, prm, tiles, acc.data_ptr(), cnt.data_ptr())
    ah, ch, _ = g.render_round(cornell.camera, prm, tiles)
    assert np.array_equal(acc.cpu().numpy(), ah) and np.array_equal(cnt.cpu().numpy().view(np.uint32), ch)


def test_render_driver_rounds(rd, oracle, cornell, tmp_path):
    """RenderFrame in rounds mode: seedcount advances across rounds like render_driver.cpp:160,222; the image file is
    rewritten after every round (:233) as Normalize(output_scale).Write."""
    import torch
    g, o = both(rd, oracle, cornell)

    class Cfg:
        xres, yres, render_rounds, render_minutes = cornell.xres, cornell.yres, 2, None
        get_params = staticmethod(lambda sampler=0, flags=0: cornell.params(sampler, flags))
    drv = rd.RenderDriver(g, Cfg, cornell.camera)
    ob = drv.render_frame(output_file=str(tmp_path / "cornell.exr"))
    n = len(oracle.generate_task_list(cornell.xres, cornell.yres))
    acc = np.zeros((cornell.yres, cornell.xres, 3), np.float32); cnt = np.zeros((cornell.yres, cornell.xres), np.uint32)
    for r in range(2):
        o.render_round(cornell.camera, cornell.params(), oracle.generate_task_list(cornell.xres, cornell.yres, seedcount_base=r * n), acc, cnt)
    got = ob.data.cpu().numpy()
    assert np.array_equal(ob.count.cpu().numpy().view(np.uint32), cnt)
    assert np.linalg.norm(got - acc) / np.linalg.norm(acc) <= 1e-3
    px = ob.get_pixels().cpu().numpy()
    assert np.allclose(px, got / cnt[..., None], rtol=1e-6, atol=1e-7)
    assert float(ob.normalize(-1.0).max()) == pytest.approx(1.0)
    img = rd.read_exr(str(tmp_path / "cornell.exr"))       # auto-normalised half-float RGBA, A = 1
    assert img.shape == (cornell.yres, cornell.xres, 4) and (img[..., 3] == 1.0).all() and img[..., :3].max() == 1.0
    assert np.allclose(img[..., :3], ob.normalize(-1.0).cpu().numpy(), rtol=2e-3, atol=1e-4)


def test_checkpoint_resume_is_bit_exact(rd, cornell, tmp_path):
    """render 2 rounds, save the raw accumulator, load it into a fresh driver, render 2 more == 4 rounds in
    one go, bit for bit (the checkpoint carries rounds_done and the running task counter that seeds the next round)."""
    g = rd.Scene(cornell.builder.to_desc())

    class Cfg:
        xres, yres, render_rounds, render_minutes = cornell.xres, cornell.yres, 4, None
        get_params = staticmethod(lambda sampler=0, flags=0: cornell.params(sampler, flags))
    whole = rd.RenderDriver(g, Cfg, cornell.camera)
    whole.render_frame(rounds=4)
    first = rd.RenderDriver(g, Cfg, cornell.camera)
    first.render_frame(rounds=2)
    ck = str(tmp_path / "frame.rgkacc")
    first.save_checkpoint(ck)
    second = rd.RenderDriver(g, Cfg, cornell.camera)
    second.load_checkpoint(ck)
    assert (second.rounds_done, second.seedcount) == (2, 2 * first.n_tasks)
    second.render_frame(rounds=2)
    assert second.rounds_done == 4
    assert np.array_equal(second.total_ob.count.cpu().numpy(), whole.total_ob.count.cpu().numpy())
    assert np.array_equal(second.total_ob.data.cpu().numpy().view(np.uint32), whole.total_ob.data.cpu().numpy().view(np.uint32))
    # a truncated file and a resolution mismatch are refused
    open(ck + ".bad", "wb").write(open(ck, "rb").read()[:1000])
    with pytest.raises(capi.RgkError):
        second.load_checkpoint(ck + ".bad")


def test_accumulator_object_and_rccl_reduce_world_size_1(rd, product_lib, cornell):
    """The device accumulator of the C ABI (rgk_accum_*) and the RCCL reduce behind it (rgk_comm_*, one rank: the sum of one
    contribution is itself): what a C++ host without HIP headers or torch uses (tests/cpp/caller_main.cpp drives the same calls)."""
    g = rd.Scene(cornell.builder.to_desc())
    prm = cornell.params()
    tiles = rd.generate_task_list(cornell.xres, cornell.yres)
    acc = C.c_void_p()
    assert product_lib.rgk_accum_create(cornell.xres, cornell.yres, 0, C.byref(acc)) == 0
    g.render_round_device(cornell.camera, prm, tiles, product_lib.rgk_accum_rgb(acc), product_lib.rgk_accum_count(acc))
    a = np.empty((cornell.yres, cornell.xres, 3), np.float32); c = np.empty((cornell.yres, cornell.xres), np.uint32)
    assert product_lib.rgk_accum_download(acc, a.ctypes.data, c.ctypes.data) == 0
    ah, ch, _ = g.render_round(cornell.camera, prm, tiles)
    assert np.array_equal(a, ah) and np.array_equal(c, ch)
    ident = (C.c_uint8 * 128)()
    comm = C.c_void_p()
    assert product_lib.rgk_comm_get_unique_id(ident) == 0, product_lib.rgk_last_error()
    assert product_lib.rgk_comm_create(ident, 0, 1, 0, C.byref(comm)) == 0, product_lib.rgk_last_error()
    assert product_lib.rgk_accum_reduce(comm, product_lib.rgk_accum_rgb(acc), product_lib.rgk_accum_count(acc), cornell.xres, cornell.yres, 0) == 0
    a2 = np.empty_like(a); c2 = np.empty_like(c)
    assert product_lib.rgk_accum_download(acc, a2.ctypes.data, c2.ctypes.data) == 0
    assert np.array_equal(a2, a) and np.array_equal(c2, c)
    assert product_lib.rgk_accum_reduce(comm, product_lib.rgk_accum_rgb(acc), None, cornell.xres, cornell.yres, 3) == -1  # root outside the communicator
    product_lib.rgk_comm_destroy(comm)
    assert product_lib.rgk_accum_clear(acc) == 0
    assert product_lib.rgk_accum_download(acc, a2.ctypes.data, c2.ctypes.data) == 0 and not a2.any() and not c2.any()
    product_lib.rgk_accum_destroy(acc)


def test_ragged_tiles_pixel_list(rd, oracle):
    """Frames that are not multiples of 32 (nor of the 8x8 slot blocks): the device-built pixel list must visit every pixel of
    every ragged tile once, with the seed of its row-major rank in the tile."""
    from rgk_amd.workloads import Workload
    for (w, h) in ((37, 29), (70, 33), (5, 3)):
        wl = Workload("cornell-256", scale=1.0, spp=2)
        wl.xres, wl.yres = w, h
        from rgk_amd.config import make_camera
        wl.camera = make_camera(wl.camera.ctor["pos"], wl.camera.ctor["lookat"], wl.camera.ctor["up"], fov=19.5, xres=w, yres=h)
        g, o = both(rd, oracle, wl)
        prm = wl.params()
        ag, cg, _ = g.render_round(wl.camera, prm, rd.generate_task_list(w, h))
        ao, co, _ = o.render_round(wl.camera, prm, oracle.generate_task_list(w, h))
        assert np.array_equal(cg, co) and (cg == 2).all()
        rel = np.linalg.norm(ag - ao) / np.linalg.norm(ao)
        print(f"[ragged {w}x{h}] rel-L2 {rel:.2e}")
        assert rel <= 1e-3


# ----------------------------------------------------------------------- unit level: BxDF::value / sample, texture lookups (a11, a12, a14)
def _unit_dirs(rng, n):
    """Local-frame directions: mostly the upper hemisphere, some below it, some exactly grazing / along the normal."""
    v = rng.normal(size=(n, 3)).astype(np.float32)
    v[:, 2] = np.abs(v[:, 2])
    v[::11, 2] *= -1.0
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    v[::97] = (0.0, 0.0, 1.0)
    return np.ascontiguousarray(v.astype(np.float32))


def test_bxdf_value_and_sample_unit_level(rd, oracle, product_lib):
    """BxDF::value / BxDF::sample of every material kind on a direction x uv x sample grid, HIP (both routes: the generic
    code and the shading kernel's shared-fetch route) against the oracle's restatement of src/bxdf/bxdf.cpp + src/LTC/ltc.cpp.
    The LTC closed forms of the device keep the oracle's operation order, so values are bit-identical except through libm
    (acosf, sinf, cosf, sqrtf are the same IEEE results; acosf/sinf/cosf differ by <= 2 ulp between glibc and ocml)."""
    sb = material_zoo()
    # + textured LTC materials with every roughness the table's alpha axis sees, bump-free (textures are covered below)
    for k, r in enumerate((0.02, 0.0995, 0.3, 0.6, 1.0)):
        m = sb.new_material(f"ltc{k}", capi.BXDF_LTC_GGX_DIFFUSE if k % 2 else capi.BXDF_LTC_BECKMANN_DIFFUSE)
        m["tex_diffuse"] = sb.create_solid_texture((0.6, 0.5, 0.4)); m["tex_color"] = sb.create_solid_texture((0.3, 0.3, 0.25)); m["roughness"] = r
        sb.register_material(m)
    for k, kind in enumerate((capi.BXDF_LTC_GGX_DIFFUSE, capi.BXDF_LTC_BECKMANN)):   # black lobes (Ks = 0, as 18 of Sponza's 20 materials): Q6
        m = sb.new_material(f"black{k}", kind)
        m["tex_diffuse"] = sb.create_solid_texture((0.6, 0.5, 0.4)); m["tex_color"] = sb.create_solid_texture((0.0, 0.0, 0.0)); m["roughness"] = 0.3
        sb.register_material(m)
    desc = sb.to_desc()
    g, o = rd.Scene(desc), oracle.OracleScene(desc)
    L = oracle.lib()
    rng = np.random.default_rng(21)
    n_mat, n = len(sb.materials), 4000
    mat = (np.arange(n) % n_mat).astype(np.uint32)
    Vi, Vr = _unit_dirs(rng, n), _unit_dirs(rng, n)
    # delta lobes need exactly mirrored / refracted pairs to be non-zero: give a share of the inputs Vr = reflect(Vi)
    Vr[::3] = Vi[::3] * np.array([-1, -1, 1], np.float32)
    uv = rng.uniform(-1, 2, (n, 2)).astype(np.float32)
    u = rng.uniform(0, 1, (n, 2)).astype(np.float32)
    ref_v = np.zeros((n, 3), np.float32); ref_d = np.zeros((n, 3), np.float32); ref_w = np.zeros((n, 3), np.float32); ref_l = np.zeros(n, np.int32)
    for i in range(n):
        L.orc_bxdf_value(o.h, int(mat[i]), Vi[i].ctypes.data, Vr[i].ctypes.data, uv[i].ctypes.data, ref_v[i].ctypes.data)
        ml = C.c_int(0)
        L.orc_bxdf_sample(o.h, int(mat[i]), Vi[i].ctypes.data, uv[i].ctypes.data, u[i].ctypes.data, ref_d[i].ctypes.data, ref_w[i].ctypes.data, C.byref(ml))
        ref_l[i] = ml.value
    for route in (0, 1):
        val = np.zeros((n, 3), np.float32); d = np.zeros((n, 3), np.float32); w = np.zeros((n, 3), np.float32); leak = np.zeros(n, np.uint8)
        assert product_lib.rgk_bxdf_value(g.h, n, route, mat.ctypes.data, Vi.ctypes.data, Vr.ctypes.data, uv.ctypes.data, val.ctypes.data) == 0
        assert product_lib.rgk_bxdf_sample(g.h, n, route, mat.ctypes.data, Vi.ctypes.data, uv.ctypes.data, u.ctypes.data, d.ctypes.data, w.ctypes.data, leak.ctypes.data) == 0
        both_nan = np.isnan(val) & np.isnan(ref_v)          # Q6: view exactly along N -> singular LTC frame -> NaN on both sides
        ev = np.abs(val - ref_v)[~both_nan]; sv = np.maximum(np.abs(ref_v), 1e-6)[~both_nan]
        assert np.array_equal(np.isnan(val), np.isnan(ref_v))
        rel_v = float((ev / sv).max())
        exact_v = float((val.view(np.uint32) == ref_v.view(np.uint32)).all(axis=1).mean())
        ok = ~np.isnan(ref_d).any(axis=1) & (ref_w.max(axis=1) > 0)   # a zero weight ends the path: its direction is never read (path_tracer.cpp:268-279)
        err_d = float(np.abs(d[ok] - ref_d[ok]).max())
        exact_d = float((d[ok].view(np.uint32) == ref_d[ok].view(np.uint32)).all(axis=1).mean())
        record_parity(f"test_bxdf_value_and_sample_unit_level:route{route}", n=n, value_max_rel=rel_v, value_bit_identical=exact_v,
                      sample_dir_max_abs=err_d, sample_dir_bit_identical=exact_d, nonzero_values=float((ref_v.max(axis=1) > 0).mean()))
        assert rel_v <= 2e-4 and err_d <= 2e-5        # libm only: acosf (the LTC table angle: a 1-ulp theta moves the bilinear weights), sinf / cosf (disc sample)
        assert np.array_equal(w.view(np.uint32), ref_w.view(np.uint32)) or np.abs(w - ref_w).max() <= 1e-7
        assert np.array_equal(leak.astype(np.int32), ref_l)
    assert (ref_v.max(axis=1) > 0).mean() > 0.3 and ref_l.any()
    assert product_lib.rgk_bxdf_value(g.h, 1, 0, np.array([999], np.uint32).ctypes.data, Vi.ctypes.data, Vr.ctypes.data, uv.ctypes.data, val.ctypes.data) == -1


def test_texture_lookup_unit_level(rd, oracle, product_lib):
    """GetPixelInterpolated / GetSlopeRight / GetSlopeBottom (src/texture.cpp:35-102): solid, empty, float and 8-bit textures
    (the PNG of rubiks-bump and one of the shipped Sponza JPGs as decoded bytes), uv inside, outside (repeat), on texel
    centres and on the wrap seam: integer / float arithmetic only, so HIP and oracle must agree bit for bit."""
    from rgk_amd.proxy import shipped_sponza_textures
    sb = SceneBuilder()
    rng = np.random.default_rng(22)
    ids = [sb.create_solid_texture((0.2, 0.5, 0.9)), -1,
           sb.add_image_texture("f32", rng.random((37, 53, 3)).astype(np.float32) * 3.0),
           sb.add_image_texture8("u8", rng.integers(0, 256, (29, 64, 3), dtype=np.uint8))]
    jpg = shipped_sponza_textures().get("KAMEN.JPG")
    if jpg is not None:
        ids.append(sb.add_image_texture8("kamen", np.ascontiguousarray(jpg[::-1])))
    m = sb.new_material("m", capi.BXDF_DIFFUSE); m["tex_diffuse"] = ids[0]; sb.register_material(m)
    sb.add_primitive("cube", np.eye(4, dtype=np.float32), "m")
    desc = sb.to_desc()
    g, o = rd.Scene(desc), oracle.OracleScene(desc)
    L = oracle.lib()
    n = 6000
    tex = np.array([ids[i % len(ids)] for i in range(n)], dtype=np.int32)
    uv = rng.uniform(-2, 3, (n, 2)).astype(np.float32)
    uv[::7] = np.round(uv[::7] * 8) / 8                      # seams and exact fractions
    uv[::13] = (np.floor(uv[::13] * 53) + 0.5) / 53          # texel centres of the 53-wide image
    rgb = np.zeros((n, 3), np.float32); sr = np.zeros(n, np.float32); sbm = np.zeros(n, np.float32)
    assert product_lib.rgk_texture_sample(g.h, n, tex.ctypes.data, uv.ctypes.data, rgb.ctypes.data, sr.ctypes.data, sbm.ctypes.data) == 0
    ref = np.zeros((n, 3), np.float32); rr = np.zeros(n, np.float32); rb = np.zeros(n, np.float32)
    for i in range(n):
        a, b = C.c_float(0), C.c_float(0)
        L.orc_texture_sample(o.h, int(tex[i]), uv[i].ctypes.data, ref[i].ctypes.data, C.byref(a), C.byref(b))
        rr[i], rb[i] = a.value, b.value
    record_parity("test_texture_lookup_unit_level", n=n, rgb_bit_identical=float((rgb.view(np.uint32) == ref.view(np.uint32)).all(axis=1).mean()),
                  slopes_bit_identical=float(((sr.view(np.uint32) == rr.view(np.uint32)) & (sbm.view(np.uint32) == rb.view(np.uint32))).mean()))
    assert np.array_equal(rgb.view(np.uint32), ref.view(np.uint32))
    assert np.array_equal(sr.view(np.uint32), rr.view(np.uint32)) and np.array_equal(sbm.view(np.uint32), rb.view(np.uint32))
    assert product_lib.rgk_texture_sample(g.h, 1, np.array([77], np.int32).ctypes.data, uv.ctypes.data, rgb.ctypes.data, sr.ctypes.data, sbm.ctypes.data) == -1


def test_pinned_libm_same_bits_on_gpu_and_cpu(oracle, product_lib):
    """include/rgk_libm.h compiled by hipcc for gfx950 and by g++ for the host: the same bits for every input (1 M per
    function, the ranges the path produces plus edge cases) -- what makes whole images bit-identical."""
    L = oracle.lib()
    rng = np.random.default_rng(31)
    x = np.concatenate([rng.uniform(0, 2 * np.pi, 600000), rng.uniform(-50, 50, 200000), [0.0, -0.0, np.pi, np.pi / 2, 1e-30, 6.2831855]]).astype(np.float32)
    u = np.concatenate([rng.uniform(-1, 1, 600000), 1 - rng.uniform(0, 1e-4, 100000), -1 + rng.uniform(0, 1e-4, 100000), [1, -1, 0, -0.0, 0.5, -0.5, 1.5, np.nan]]).astype(np.float32)
    d = rng.normal(size=(800000, 2)).astype(np.float32)
    d[:1000, 0] = 0.0; d[1000:2000, 1] = 0.0; d[2000:2100] *= -0.0
    for fn, a, b in ((0, x, None), (1, x, None), (2, u, None), (3, u, None), (4, d[:, 0].copy(), d[:, 1].copy())):
        a = np.ascontiguousarray(a); b = a if b is None else np.ascontiguousarray(b)
        cpu = np.zeros(len(a), np.float32); gpu = np.zeros(len(a), np.float32)
        assert L.orc_libm(fn, len(a), a.ctypes.data, b.ctypes.data, cpu.ctypes.data) == 0
        assert product_lib.rgk_libm_eval(fn, len(a), a.ctypes.data, b.ctypes.data, gpu.ctypes.data) == 0
        same = cpu.view(np.uint32) == gpu.view(np.uint32)
        both_nan = np.isnan(cpu) & np.isnan(gpu)
        record_parity(f"test_pinned_libm_same_bits_on_gpu_and_cpu:fn{fn}", n=len(a), bit_identical=float((same | both_nan).mean()))
        assert (same | both_nan).all(), (fn, a[~(same | both_nan)][:5], cpu[~(same | both_nan)][:5], gpu[~(same | both_nan)][:5])


# ----------------------------------------------------------------------- f2: the accelerator built on the device
def test_device_built_bvh_gives_the_same_hits(rd, oracle):
    """the LBVH built on the GPU (Morton sort, Karras hierarchy, refit with tree rotations, collapse + quantisation
    on the device, rgk_build.hip) against the ORACLE's kd-tree walk (same bar as the host-built tree: same triangle except inside
    the epsilon tie band, then t and barycentrics bit for bit) and against the host's binned-SAH tree (another tree over the same
    triangle records) -- on cube3, box6, the Sponza proxy and the 1.05 M-triangle dragon scene; images equal too.  Records build
    time and node visits per ray of the host tree, the device tree, and the device tree without its rotation step."""
    import time
    from rgk_amd.workloads import Workload
    cases = [("cube3", scene_fixture("cube3", scale=0.1, spp=4)), ("box6", scene_fixture("box6", scale=0.1, spp=4)),
             ("sponza-proxy", Workload("sponza-1080p", scale=0.1, spp=4)), ("dragon-sponza-proxy 1.05 M", Workload("dragon-sponza-1080p", scale=0.05, spp=2))]
    for name, wl in cases:
        sb = wl.builder
        sb.build_flags = capi.BUILD_HOST_SAH
        t0 = time.time(); gh = rd.Scene(sb.to_desc()); th = time.time() - t0
        sb.build_flags = capi.BUILD_DEVICE
        t0 = time.time(); gd = rd.Scene(sb.to_desc()); td = time.time() - t0
        os.environ["RGK_LBVH_ROTATE"] = "0"      # (read once, in rgk_scene_create)
        try:
            gp = rd.Scene(sb.to_desc())
        finally:
            del os.environ["RGK_LBVH_ROTATE"]
        sb.build_flags = capi.BUILD_HOST_SAH
        o = oracle.OracleScene(sb.to_desc())
        ih, idv = gh.info(), gd.info()
        assert ih.epsilon == idv.epsilon and list(ih.bbox_min) == list(idv.bbox_min) and idv.n_leaf_refs == ih.n_leaf_refs
        lo, hi = np.array(list(ih.bbox_min)), np.array(list(ih.bbox_max))
        oo, dd = random_rays(np.random.default_rng(41), lo, hi, 200000)
        rays = make_rays(oo, dd)
        # the oracle on the other side (cube3 is full of coincident faces: ties inside the epsilon band are what check_closest allows)
        check_closest(gd, o, rays, ih.epsilon, max_unexplained=5e-5, name="test_device_built_bvh_gives_the_same_hits:" + name + ":vs-oracle")
        hh, ch = gh.trace_closest(rays, count=True)
        hd, cd = gd.trace_closest(rays, count=True)
        hp, cp = gp.trace_closest(rays, count=True)
        same = hh["tri"] == hd["tri"]
        for k in ("t", "a", "b", "c"):
            assert np.array_equal(hh[k][same].view(np.uint32), hd[k][same].view(np.uint32)), (name, k)
        assert np.array_equal(hd["tri"], hp["tri"]) and np.array_equal(hd["t"].view(np.uint32), hp["t"].view(np.uint32)), name   # rotations change the tree, not a hit
        bad = ~same
        with np.errstate(invalid="ignore"):
            tie = bad & (hh["tri"] >= 0) & (hd["tri"] >= 0) & (np.abs(hh["t"] - hd["t"]) <= 2 * ih.epsilon)
        record_parity("test_device_built_bvh_gives_the_same_hits:" + name, triangles=len(sb.F), same_triangle=float(same.mean()), eps_band_ties=int(tie.sum()),
                      unexplained=int(bad.sum() - tie.sum()), host_scene_s=th, device_scene_s=td, host_nodes=ih.n_nodes, device_nodes=idv.n_nodes,
                      host_nodes_per_ray=ch.node_visits / len(rays), device_nodes_per_ray=cd.node_visits / len(rays),
                      device_plain_lbvh_nodes_per_ray=cp.node_visits / len(rays),
                      host_trace_ms=ch.ms_trace, device_trace_ms=cd.ms_trace, device_plain_trace_ms=cp.ms_trace, device_levels=idv.max_depth)
        assert bad.sum() - tie.sum() <= max(1, 5e-5 * len(rays)), (name, int(bad.sum()), int(tie.sum()))
        a = (lo + (hi - lo) * np.random.default_rng(42).uniform(0.02, 0.98, (100000, 3))).astype(np.float32)
        b = (lo + (hi - lo) * np.random.default_rng(43).uniform(0.02, 0.98, (100000, 3))).astype(np.float32)
        assert (gh.visibility(a, b)[0] == gd.visibility(a, b)[0]).mean() > 0.9999
        assert (o.visibility(a, b)[0] != gd.visibility(a, b)[0]).mean() < 2e-3
        prm = wl.params()
        tiles = rd.generate_task_list(wl.xres, wl.yres)
        ah = gh.render_round(wl.camera, prm, tiles)[0]
        ad = gd.render_round(wl.camera, prm, tiles)[0]
        rel = float(np.linalg.norm(ah - ad) / np.linalg.norm(ah))
        record_parity("test_device_built_bvh_gives_the_same_hits:" + name + ":image", rel_l2=rel, bit_identical=float((np.abs(ah - ad).max(axis=2) == 0).mean()))
        assert rel <= 5e-3
        gh.close(); gd.close(); gp.close()


# ----------------------------------------------------------------------- textures as a reference host hands them over
def _textures_as_float(sb):
    """What tests/cpp/rgk_binding.inc hands over: every FileTexture as RGK_TEX_RGB32F (the reference keeps only the decoded
    floats, src/texture.cpp:203,252-254)."""
    n = 0
    for t in sb.textures:
        if t["kind"] == capi.TEX_RGB8:
            t["data"] = np.ascontiguousarray(t["lut"][t["data"]], dtype=np.float32)
            t["kind"], t["lut"] = capi.TEX_RGB32F, None
            n += 1
    return n


def test_float_textures_with_few_values_take_the_byte_path(rd):
    """the headline ran on 8-bit textures (bytes + byte -> float table), a format the reference-side binding
    cannot deliver -- it has floats only.  rgk_scene_create now recognises a float texture whose channels take <= 256 distinct
    values and stores it as bytes + the table of those very floats: the same texel values (so the same image, bit for bit, as
    with RGK_TEX_RGB8 input and as with float4 storage), at the 8-bit path's cost.  A texture with more values stays float."""
    from rgk_amd.workloads import SceneFixture, Workload
    for name, mk in (("sponza-proxy", lambda: Workload("sponza-1080p", scale=0.1, spp=8)),
                     ("rubiks-bump", lambda: SceneFixture(os.path.join(ROOT, "tests", "golden", "scene_rubiks-bump.npz"), scale=0.1, spp=4, depth=4))):
        a, b = mk(), mk()
        n_conv = _textures_as_float(b.builder)
        assert n_conv >= 1
        prm = a.params()
        tiles = rd.generate_task_list(prm.xres, prm.yres)
        ga = rd.Scene(a.builder.to_desc())
        gb = rd.Scene(b.builder.to_desc())
        b.builder.build_flags = capi.BUILD_KEEP_FLOAT_TEXTURES
        gc = rd.Scene(b.builder.to_desc())
        b.builder.build_flags = capi.BUILD_HOST_SAH
        ia, ib, ic = ga.info(), gb.info(), gc.info()
        assert (ia.n_float_textures, ia.n_palettized_textures) == (0, 0)
        assert ib.n_float_textures == n_conv and ib.n_palettized_textures == n_conv, (name, ib.n_float_textures, ib.n_palettized_textures)
        assert ic.n_float_textures == n_conv and ic.n_palettized_textures == 0
        ra, rb, rc = (g.render_round(a.camera, prm, tiles)[0] for g in (ga, gb, gc))
        record_parity("test_float_textures_with_few_values_take_the_byte_path:" + name, textures=n_conv,
                      u8_vs_palettized_bit_identical=float((ra == rb).all(axis=2).mean()), u8_vs_float4_bit_identical=float((ra == rc).all(axis=2).mean()))
        assert np.array_equal(ra, rb) and np.array_equal(ra, rc), name
        for g in (ga, gb, gc):
            g.close()
    # texel-level: bytes, palettized floats and float4 storage return the same bits from GetPixelInterpolated / GetSlope*
    lib = capi.load_product()
    rng = np.random.default_rng(77)
    sb = SceneBuilder()
    many = rng.random((40, 50, 3)).astype(np.float32)                       # 6000 distinct values: stays float
    few = rng.choice(rng.random(200).astype(np.float32), size=(33, 47, 3))  # 200 distinct values: byte path
    m = sb.new_material("m", capi.BXDF_DIFFUSE)
    m["tex_diffuse"] = sb.add_image_texture("many", many)
    m2 = sb.new_material("m2", capi.BXDF_DIFFUSE)
    m2["tex_diffuse"] = sb.add_image_texture("few", few)
    sb.register_material(m); sb.register_material(m2)
    from rgk_amd.scene import glm_scale
    sb.add_primitive("plane", glm_scale((1, 1, 1)), "m"); sb.add_primitive("plane", glm_scale((2, 1, 2)), "m2")
    g1 = rd.Scene(sb.to_desc())
    sb.build_flags = capi.BUILD_KEEP_FLOAT_TEXTURES
    g2 = rd.Scene(sb.to_desc())
    i1, i2 = g1.info(), g2.info()
    assert (i1.n_float_textures, i1.n_palettized_textures) == (2, 1) and (i2.n_float_textures, i2.n_palettized_textures) == (2, 0)
    uv = rng.uniform(-2, 3, (50000, 2)).astype(np.float32)
    for tex in (0, 1):
        out = []
        for g in (g1, g2):
            rgb = np.zeros((len(uv), 3), np.float32); sr = np.zeros(len(uv), np.float32); sbm = np.zeros(len(uv), np.float32)
            assert lib.rgk_texture_sample(g.h, len(uv), np.full(len(uv), tex, np.int32).ctypes.data, uv.ctypes.data, rgb.ctypes.data, sr.ctypes.data, sbm.ctypes.data) == 0
            out.append((rgb, sr, sbm))
        for x, y in zip(out[0], out[1]):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), tex


# ----------------------------------------------------------------------- f2: refit for moved vertices
def test_refit_equals_a_fresh_scene(rd, oracle):
    """rgk_scene_refit ("+ refit"): vertices moved, triangles kept.  The refitted scene -- host-built tree and
    device-built tree -- must answer like a scene created from the moved vertices: Commit's scalars (epsilon, padded box, areal
    power) equal, closest hits equal to the ORACLE's on the moved mesh under the usual bar and equal bit for bit to the fresh
    scene's, the same image; and refitting back gives the original scene's bits again.  Records what the old topology costs on
    the new positions (node visits per ray, refitted vs rebuilt)."""
    import copy, time
    from rgk_amd.workloads import Workload
    cases = [("box6 (emissive triangles)", scene_fixture("box6", scale=0.1, spp=4)), ("sponza-proxy", Workload("sponza-1080p", scale=0.1, spp=4)),
             ("dragon-sponza-proxy 1.05 M", Workload("dragon-sponza-1080p", scale=0.05, spp=2))]
    for name, wl in cases:
        sb = wl.builder
        sb.finalize()
        V0 = sb.V.copy()
        ctr, ext = V0.mean(axis=0), np.ptp(V0, axis=0).max()
        # a smooth, non-rigid move: a twist about the vertical axis growing with height + a swell, a few per cent of the scene size
        ang = 0.25 * (V0[:, 1] - ctr[1]) / ext
        c, s_ = np.cos(ang), np.sin(ang)
        V1 = V0.copy()
        V1[:, 0] = ctr[0] + c * (V0[:, 0] - ctr[0]) - s_ * (V0[:, 2] - ctr[2])
        V1[:, 2] = ctr[2] + s_ * (V0[:, 0] - ctr[0]) + c * (V0[:, 2] - ctr[2])
        V1 = (V1 + 0.02 * ext * np.sin(3.0 * V0[:, [1, 2, 0]] / ext)).astype(np.float32)
        prm = wl.params()
        tiles = rd.generate_task_list(wl.xres, wl.yres)
        for flags, tag in ((capi.BUILD_HOST_SAH, "host tree"), (capi.BUILD_DEVICE, "device tree")):
            sb.vertices = [V0]; sb.build_flags = flags     # (to_desc() re-concatenates the vertex chunks)
            g = rd.Scene(sb.to_desc())
            img0 = g.render_round(wl.camera, prm, tiles)[0]
            t0 = time.time(); g.refit(V1); t_refit = time.time() - t0
            sb.vertices = [V1]
            t0 = time.time(); f = rd.Scene(sb.to_desc()); t_fresh = time.time() - t0
            o = oracle.OracleScene(sb.to_desc())
            ig, if_, io = g.info(), f.info(), o.info()
            assert ig.epsilon == if_.epsilon == io.epsilon and list(ig.bbox_min) == list(if_.bbox_min) and list(ig.bbox_max) == list(if_.bbox_max)
            assert ig.total_areal_power == if_.total_areal_power
            lo, hi = np.array(list(ig.bbox_min)), np.array(list(ig.bbox_max))
            oo, dd = random_rays(np.random.default_rng(51), lo, hi, 200000)
            rays = make_rays(oo, dd)
            check_closest(g, o, rays, ig.epsilon, max_unexplained=5e-5, name=f"test_refit_equals_a_fresh_scene:{name}:{tag}:vs-oracle")
            hg, cg = g.trace_closest(rays, count=True)
            hf, cf = f.trace_closest(rays, count=True)
            same = hg["tri"] == hf["tri"]
            for k in ("t", "a", "b", "c"):
                assert np.array_equal(hg[k][same].view(np.uint32), hf[k][same].view(np.uint32)), (name, tag, k)
            with np.errstate(invalid="ignore"):
                tie = ~same & (hg["tri"] >= 0) & (hf["tri"] >= 0) & (np.abs(hg["t"] - hf["t"]) <= 2 * ig.epsilon)
            assert (~same).sum() - tie.sum() <= max(1, 5e-5 * len(rays)), (name, tag, int((~same).sum()), int(tie.sum()))
            a = (lo + (hi - lo) * np.random.default_rng(52).uniform(0.02, 0.98, (100000, 3))).astype(np.float32)
            b = (lo + (hi - lo) * np.random.default_rng(53).uniform(0.02, 0.98, (100000, 3))).astype(np.float32)
            assert (g.visibility(a, b)[0] == f.visibility(a, b)[0]).mean() > 0.9999
            ig_img = g.render_round(wl.camera, prm, tiles)[0]
            if_img = f.render_round(wl.camera, prm, tiles)[0]
            rel = float(np.linalg.norm(ig_img - if_img) / np.linalg.norm(if_img))
            record_parity(f"test_refit_equals_a_fresh_scene:{name}:{tag}", triangles=len(sb.F), refit_s=t_refit, fresh_scene_s=t_fresh, same_triangle=float(same.mean()),
                          eps_band_ties=int(tie.sum()), refit_nodes_per_ray=cg.node_visits / len(rays), rebuilt_nodes_per_ray=cf.node_visits / len(rays),
                          image_rel_l2=rel, image_bit_identical=float((np.abs(ig_img - if_img).max(axis=2) == 0).mean()))
            assert rel <= 5e-3 and not np.array_equal(ig_img, img0)
            g.refit(V0)                                        # and back: the original scene's bits (reverse > 0: up to the order of the splats' float atomics)
            back = g.render_round(wl.camera, prm, tiles)[0]
            assert np.array_equal(back, img0) if wl.reverse == 0 else np.linalg.norm(back - img0) / np.linalg.norm(img0) <= 1e-6, (name, tag)
            g.close(); f.close()
        sb.vertices = [V0]; sb.build_flags = capi.BUILD_AUTO
    # argument errors
    cube = scene_fixture("cube3", scale=0.1, spp=4)
    g = rd.Scene(cube.builder.to_desc())
    lib = capi.load_product()
    assert lib.rgk_scene_refit(g.h, None, None, None) == -1 and lib.rgk_scene_refit(None, None, None, None) == -1
