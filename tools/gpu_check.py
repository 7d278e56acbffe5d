#!/usr/bin/env python3
"""Ad-hoc check on a GPU box: sampler, closest-hit, visibility and image parity vs the oracle.

  WORKLOAD=sponza-1080p SCALE=0.15 SPP=16 python tools/gpu_check.py
"""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from rgk_amd import capi
from rgk_amd.workloads import Workload
from rgk_amd import render_driver as rd
from oracle import rgk_oracle as O

name = os.environ.get("WORKLOAD", "cornell-256")
wl = Workload(name, scale=float(os.environ.get("SCALE", "0.5")), spp=int(os.environ.get("SPP", "16")))
W, H = wl.xres, wl.yres
cam, prm = wl.camera, wl.params(flags=capi.FLAG_COUNT_TRAVERSAL)
print("workload", name, W, H, wl.multisample, "geometry", wl.geometry)
rng = np.random.default_rng(0)
desc = wl.builder.to_desc()
t = time.time(); osc = O.OracleScene(desc); print("oracle commit s", time.time() - t)
t = time.time(); gsc = rd.Scene(desc); print("gpu scene create s", time.time() - t)
gi, oi = gsc.info(), osc.info()
print("eps", gi.epsilon, oi.epsilon, "bbox eq", list(gi.bbox_min) == list(oi.bbox_min), list(gi.bbox_max) == list(oi.bbox_max),
      "bvh nodes", gi.n_nodes, "depth", gi.max_depth, "leaf refs", gi.n_leaf_refs, "| kd nodes", oi.n_nodes, "depth", oi.max_depth, "refs", oi.n_leaf_refs)
lo, hi = np.array(list(oi.bbox_min)), np.array(list(oi.bbox_max))
n = 200000
o = (lo + (hi - lo) * rng.uniform(0.05, 0.95, (n, 3))).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.concatenate([o, d, np.zeros((n, 1), np.float32), np.full((n, 1), 10000.0, np.float32)], axis=1).astype(np.float32)
t = time.time(); hg, cg = gsc.trace_closest(rays, count=True); tg = time.time() - t
t = time.time(); ho, co = osc.trace_closest(rays); to = time.time() - t
same = hg["tri"] == ho["tri"]
print("closest: same tri", same.mean(), "t bit-exact among same", np.array_equal(hg["t"][same].view(np.uint32), ho["t"][same].view(np.uint32)),
      "abc exact", all(np.array_equal(hg[k][same].view(np.uint32), ho[k][same].view(np.uint32)) for k in "abc"), "gpu s", tg, "oracle s", to)
bad = np.where(~same)[0]
dt = np.abs(hg["t"][bad] - ho["t"][bad])
print("mismatch", len(bad), "of which |dt| <= 2eps:", int((dt <= 2 * gi.epsilon).sum()),
      [(int(hg['tri'][i]), int(ho['tri'][i]), float(hg['t'][i]), float(ho['t'][i])) for i in bad[:6]])
print("gpu nodes/ray", cg.node_visits / n, "tris/ray", cg.tri_tests / n, "| kd nodes/ray", co.node_visits / n, "tris/ray", co.tri_tests / n)
a = (lo + (hi - lo) * rng.uniform(0.05, 0.95, (n, 3))).astype(np.float32)
b = (lo + (hi - lo) * rng.uniform(0.05, 0.95, (n, 3))).astype(np.float32)
vg, _ = gsc.visibility(a, b); vo, _ = osc.visibility(a, b)
print("visibility agree", (vg == vo).mean(), "visible frac", vo.mean())

tiles = rd.generate_task_list(W, H)
ag, cgc, cntg = gsc.render_round(cam, prm, tiles)
prm_t = wl.params(flags=capi.FLAG_TIME_KERNELS)
t = time.time(); ag2, _, cntt = gsc.render_round(cam, prm_t, tiles); tg = time.time() - t
otiles = O.generate_task_list(W, H)
t = time.time(); ao, coc, cnto = osc.render_round(cam, prm, otiles); to = time.time() - t
ig, io = ag / cgc[..., None], ao / coc[..., None]
print("gpu round s", tg, "Mpaths/s", cntg.paths / tg / 1e6, "| oracle s", to, "Mpaths/s", cnto.paths / to / 1e6)
print("kernel ms trace/shadow/shade/other", cntt.ms_trace, cntt.ms_shadow, cntt.ms_shade, cntt.ms_other, "launches", cntt.n_trace_launches)
print("counters gpu", cntg.path_rays, cntg.shadow_rays, "oracle", cnto.path_rays, cnto.shadow_rays)
print("gpu nodes/ray", cntg.node_visits / max(1, cntg.path_rays), "tris/ray", cntg.tri_tests / max(1, cntg.path_rays),
      "shadow nodes/ray", cntg.shadow_node_visits / max(1, cntg.shadow_rays), cntg.shadow_tri_tests / max(1, cntg.shadow_rays))
dd = np.linalg.norm(ig - io, axis=2); rr = np.linalg.norm(io, axis=2)
print("mean", ig.mean(axis=(0, 1)), io.mean(axis=(0, 1)), "nan", np.isnan(ig).sum(), np.isnan(io).sum())
print("rel L2 image", np.linalg.norm(ig - io) / np.linalg.norm(io), "exact pixels", (dd == 0).mean(),
      "pixels within 1e-3 rel", (dd <= 1e-3 * rr + 1e-6).mean(), "max abs", dd.max())
print("repeatable", np.array_equal(ag, ag2))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"check_{name}.npz"), gpu=ig, oracle=io)
