#!/usr/bin/env python3
"""Node visits per ray of the device-built tree against the number of rotation passes of its refit (RGK_LBVH_ROTATE), and of the
host-built tree: 200 k random rays + one camera round, on the Sponza proxy and the 1.05 M-triangle dragon scene."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rgk_amd import capi, render_driver as rd
from rgk_amd.workloads import Workload
from conftest import make_rays
for name, wl in (("sponza", Workload("sponza-1080p", scale=0.25, spp=8)), ("dragon", Workload("dragon-sponza-1080p", scale=0.25, spp=4))):
    sb = wl.builder
    res = []
    for tag, flags, env, ml in [("host-sah", capi.BUILD_HOST_SAH, None, None)] + [(f"device rot {r} leaf {ml}", capi.BUILD_DEVICE, str(r), str(ml)) for ml in (1, 2, 3, 4) for r in (0, 1, 4)]:
        sb.build_flags = flags
        if env is not None: os.environ["RGK_LBVH_ROTATE"] = env
        if ml is not None: os.environ["RGK_BVH_MAXLEAF_DEV"] = ml
        t0 = time.time(); g = rd.Scene(sb.to_desc()); dt = time.time() - t0
        os.environ.pop("RGK_LBVH_ROTATE", None); os.environ.pop("RGK_BVH_MAXLEAF_DEV", None)
        i = g.info()
        lo, hi = np.array(list(i.bbox_min)), np.array(list(i.bbox_max))
        rng = np.random.default_rng(41)
        o = (lo + (hi - lo) * rng.uniform(0.02, 0.98, (200000, 3))).astype(np.float32)
        d = rng.normal(size=(200000, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
        h, c = g.trace_closest(make_rays(o, d), count=True)
        prm = wl.params(flags=capi.FLAG_COUNT_TRAVERSAL)
        _, _, k = g.render_round(wl.camera, prm, rd.generate_task_list(wl.xres, wl.yres))
        kt = [g.render_round(wl.camera, wl.params(flags=capi.FLAG_TIME_KERNELS), rd.generate_task_list(wl.xres, wl.yres))[2] for _ in range(3)][-1]
        print(f"{name:7s} {tag:22s} scene {dt:6.3f} s  nodes {i.n_nodes:7d} levels {i.max_depth:3d}  random rays: {c.node_visits / 200000:6.2f} nodes {c.tri_tests / 200000:5.2f} tris per ray | round: {k.node_visits / k.path_rays:6.2f} nodes {k.tri_tests / k.path_rays:5.2f} tris per path ray, {k.shadow_node_visits / max(1, k.shadow_rays):6.2f} nodes per shadow ray | trace {kt.ms_trace:7.2f} ms shadow {kt.ms_shadow:6.2f} ms")
        g.close()
    sb.build_flags = capi.BUILD_HOST_SAH
