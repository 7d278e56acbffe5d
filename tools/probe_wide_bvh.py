#!/usr/bin/env python3
"""Would an 8-wide BVH pay?  Dumps the Sponza proxy's triangles and 480x270 camera rays, builds tools/probe_wide_bvh.cpp (CPU,
its own small binned-SAH builder) and prints node visits and triangle tests per ray for: 4-wide with distance-sorted children
(what the product walks), 8-wide distance-sorted, 8-wide in octant order (slot ^ ~octant: no sort, one stack entry per node)
without and with octant-aware slot assignment.  Camera rays and one diffuse bounce off their hits.  No GPU needed."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from rgk_amd.workloads import Workload
wl = Workload("sponza-1080p")
sb = wl.builder; sb.finalize()
d = tempfile.mkdtemp()
sb.V.astype(np.float32)[sb.F].reshape(-1, 9).astype(np.float32).tofile(os.path.join(d, "tris.bin"))
cam = wl.camera
W, H = 480, 270
ys, xs = np.mgrid[0:H, 0:W]
vs, vx, vy, org = (np.array(list(getattr(cam, n)), np.float32) for n in ("viewscreen", "viewscreen_x", "viewscreen_y", "origin"))
p = vs[None] + ((xs.ravel() + 0.5) / W)[:, None].astype(np.float32) * vx[None] + ((ys.ravel() + 0.5) / H)[:, None].astype(np.float32) * vy[None]
dr = p - org[None]; dr /= np.linalg.norm(dr, axis=1, keepdims=True)
np.concatenate([np.tile(org, (len(dr), 1)), dr], 1).astype(np.float32).tofile(os.path.join(d, "rays.bin"))
exe = os.path.join(d, "probe")
subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tools", "probe_wide_bvh.cpp"), "-o", exe])
subprocess.check_call([exe, os.path.join(d, "tris.bin"), os.path.join(d, "rays.bin")])
