#!/usr/bin/env python3
"""Node visits per ray and traversal time of the device-built tree against the Morton bits per axis of its sort keys
(RGK_LBVH_MORTON_BITS; 10 = round 2's keys, default = what the 64-bit key leaves beside the index), beside the host-built tree:
200 k random rays + rounds at a quarter of the resolution, on the Sponza proxy and the 1.05 M-triangle dragon scene."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rgk_amd import capi, render_driver as rd
from rgk_amd.workloads import Workload
from conftest import make_rays
for name, wl in (("sponza", Workload("sponza-1080p", scale=0.25, spp=8)), ("dragon", Workload("dragon-sponza-1080p", scale=0.25, spp=4))):
    sb = wl.builder
    variants = [("host-sah", capi.BUILD_HOST_SAH, {})]
    variants += [(f"device, Karras, {b or 'all'} bits per axis", capi.BUILD_DEVICE, {"RGK_LBVH_PLOC": "0", "RGK_LBVH_MORTON_BITS": str(b or 21)}) for b in (10, 0)]
    variants += [(f"device, PLOC radius {r}, {rot} rotation passes", capi.BUILD_DEVICE, {"RGK_LBVH_PLOC": str(r), "RGK_LBVH_ROTATE": str(rot)}) for r in (1, 2, 3, 4, 6, 8, 12) for rot in (0, 4)]
    variants += [(f"device, PLOC radius 6, leaves of {ml}", capi.BUILD_DEVICE, {"RGK_LBVH_PLOC": "6", "RGK_BVH_MAXLEAF_DEV": str(ml)}) for ml in (1, 3)]
    for tag, flags, env in variants:
        sb.build_flags = flags
        os.environ.update(env)
        t0 = time.time(); g = rd.Scene(sb.to_desc()); dt = time.time() - t0
        for k_ in env: os.environ.pop(k_, None)
        i = g.info()
        lo, hi = np.array(list(i.bbox_min)), np.array(list(i.bbox_max))
        rng = np.random.default_rng(41)
        o = (lo + (hi - lo) * rng.uniform(0.02, 0.98, (200000, 3))).astype(np.float32)
        d = rng.normal(size=(200000, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
        h, c = g.trace_closest(make_rays(o, d), count=True)
        prm = wl.params(flags=capi.FLAG_COUNT_TRAVERSAL)
        _, _, k = g.render_round(wl.camera, prm, rd.generate_task_list(wl.xres, wl.yres))
        kt = [g.render_round(wl.camera, wl.params(flags=capi.FLAG_TIME_KERNELS), rd.generate_task_list(wl.xres, wl.yres))[2] for _ in range(3)][-1]
        print(f"{name:7s} {tag:44s} scene {dt:6.3f} s  nodes {i.n_nodes:7d} levels {i.max_depth:3d}  random rays: {c.node_visits / 200000:6.2f} nodes {c.tri_tests / 200000:5.2f} tris per ray | round: {k.node_visits / k.path_rays:6.2f} nodes {k.tri_tests / k.path_rays:5.2f} tris per path ray, {k.shadow_node_visits / max(1, k.shadow_rays):6.2f} nodes per shadow ray | trace {kt.ms_trace:7.2f} ms shadow {kt.ms_shadow:6.2f} ms", flush=True)
        g.close()
    sb.build_flags = capi.BUILD_HOST_SAH
