#!/usr/bin/env python3
"""First-light check on a GPU box: sampler, closest-hit and image parity against the oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from rgk_amd import capi
from rgk_amd.scene import SceneBuilder
from rgk_amd.config import make_camera, make_params
from rgk_amd import render_driver as rd
from oracle import rgk_oracle as O

sb = SceneBuilder.load_npz(os.path.join(ROOT, "tests/golden/cornell_scene.npz"))
ex = sb.extra
W = H = int(os.environ.get("RES", "128")); S = int(os.environ.get("SPP", "16"))
cam = make_camera(ex["camera"]["pos"], ex["camera"]["lookat"], ex["camera"]["up"], fov=ex["camera"]["fov"], xres=W, yres=H)
prm = make_params(W, H, S, ex["depth"], ex["clamp"], ex["russian"], ex["bumpscale"])

# 1. sampler
rng = np.random.default_rng(0)
n = 20000
seed = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
idx = rng.integers(0, 4096, n).astype(np.uint32)
dim = rng.integers(0, 70, n).astype(np.uint32)
for is2d in (0, 1):
    a = rd.sampler_eval(seed, idx, dim, is2d); b = O.sampler_eval(seed, idx, dim, is2d)
    print("sampler is2d", is2d, "bit-exact", np.array_equal(a.view(np.uint32), b.view(np.uint32)), "maxdiff", np.abs(a-b).max())

# 2. closest hit
desc = sb.to_desc()
osc = O.OracleScene(desc)
gsc = rd.Scene(desc)
gi, oi = gsc.info(), osc.info()
print("eps", gi.epsilon, oi.epsilon, "bbox", list(gi.bbox_min), list(oi.bbox_min), "nodes", gi.n_nodes, "depth", gi.max_depth)
n = 200000
o = rng.uniform(-0.9, 0.9, (n, 3)).astype(np.float32); o[:, 1] = o[:, 1] + 1.0
d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.concatenate([o, d, np.zeros((n, 1), np.float32), np.full((n, 1), 10000.0, np.float32)], axis=1).astype(np.float32)
hg, cg = gsc.trace_closest(rays, count=True); ho, co = osc.trace_closest(rays)
same = hg["tri"] == ho["tri"]
print("closest: same tri", same.mean(), "t bit-exact among same", np.array_equal(hg["t"][same].view(np.uint32), ho["t"][same].view(np.uint32)),
      "abc exact", all(np.array_equal(hg[k][same].view(np.uint32), ho[k][same].view(np.uint32)) for k in "abc"))
bad = np.where(~same)[0]
print("mismatch", len(bad), [(int(hg['tri'][i]), int(ho['tri'][i]), float(hg['t'][i]), float(ho['t'][i])) for i in bad[:8]])
print("gpu nodes/ray", cg.node_visits / n, "tris/ray", cg.tri_tests / n, "oracle nodes/ray", co.node_visits / n, co.tri_tests / n)
a = rng.uniform(-0.9, 0.9, (n, 3)).astype(np.float32); a[:, 1] += 1.0
b = rng.uniform(-0.9, 0.9, (n, 3)).astype(np.float32); b[:, 1] += 1.0
vg, _ = gsc.visibility(a, b); vo, _ = osc.visibility(a, b)
print("visibility agree", (vg == vo).mean(), "visible frac", vo.mean())

# 3. image
tiles = rd.generate_task_list(W, H)
t = time.time(); ag, cgc, cntg = gsc.render_round(cam, prm, tiles); tg = time.time() - t
t = time.time(); ag, cgc, cntg = gsc.render_round(cam, prm, tiles, None, None); tg = time.time() - t
otiles = O.generate_task_list(W, H)
t = time.time(); ao, coc, cnto = osc.render_round(cam, prm, otiles); to = time.time() - t
ig, io = ag / cgc[..., None], ao / coc[..., None]
print("gpu time", tg, "Mpaths/s", cntg.paths / tg / 1e6, "oracle time", to, "Mpaths/s", cnto.paths / to / 1e6)
print("counters gpu", cntg.path_rays, cntg.shadow_rays, "oracle", cnto.path_rays, cnto.shadow_rays)
dd = np.linalg.norm(ig - io, axis=2); rr = np.linalg.norm(io, axis=2)
print("mean", ig.mean(axis=(0, 1)), io.mean(axis=(0, 1)))
print("rel L2 image", np.linalg.norm(ig - io) / np.linalg.norm(io), "exact pixels", (dd == 0).mean(),
      "pixels within 1e-3 rel", (dd <= 1e-3 * rr + 1e-6).mean(), "max abs", dd.max())
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "first_images.npz"), gpu=ig, oracle=io)
