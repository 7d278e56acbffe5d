#!/usr/bin/env python3
"""Where do the idle lanes of the closest-hit walker sit?  Camera rays and diffuse bounce rays off the Sponza proxy, traced by the
product kernel's counting variant with RGK_DEBUG_UTIL=1: lane-visits / (64 x wave iterations) for the node loop and the
triangle loop, outer iterations and refills (printed by rgk_trace_closest on stderr)."""
import os, sys
import numpy as np
os.environ["RGK_DEBUG_UTIL"] = "1"
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from rgk_amd import render_driver as rd
from rgk_amd.workloads import Workload

wl = Workload("sponza-1080p")
sb = wl.builder; sb.finalize()
g = rd.Scene(sb.to_desc())
W, H = wl.xres, wl.yres
cam = wl.camera
ys, xs = np.mgrid[0:H, 0:W]
key = ((ys // 32) * (W // 32 + 1) + (xs // 32)) * 1024 + ((ys % 32) // 8 * 4 + (xs % 32) // 8) * 64 + (ys % 8) * 8 + (xs % 8)
order = np.argsort(key.ravel(), kind="stable")
px, py = xs.ravel()[order], ys.ravel()[order]
vs, vx, vy, org = (np.array(list(getattr(cam, n)), np.float32) for n in ("viewscreen", "viewscreen_x", "viewscreen_y", "origin"))
rng = np.random.default_rng(0)
reps = 4
jx, jy = rng.random((2, reps, len(px))).astype(np.float32)
p = vs[None, None] + ((px[None] + jx) / W)[..., None] * vx + ((py[None] + jy) / H)[..., None] * vy
d = (p - org).reshape(-1, 3).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
n = len(d)
rays = np.concatenate([np.tile(org, (n, 1)), d, np.zeros((n, 1), np.float32), np.full((n, 1), 1e4, np.float32)], 1).astype(np.float32)
print("camera rays", n, flush=True)
hits, c = g.trace_closest(rays, count=True)
print(f"  {c.ms_trace:.3f} ms  {n / c.ms_trace / 1e6:.2f} G rays/s  nodes/ray {c.node_visits / n:.2f}  tris/ray {c.tri_tests / n:.2f}", flush=True)
ok = hits["tri"] >= 0
V, F = sb.V, sb.F
tri = hits["tri"][ok]
e1, e2 = V[F[tri, 1]] - V[F[tri, 0]], V[F[tri, 2]] - V[F[tri, 0]]
nrm = np.cross(e1, e2); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
dd = d[ok]
nrm = np.where((np.sum(nrm * dd, 1) > 0)[:, None], -nrm, nrm)
O = org[None] + dd * hits["t"][ok][:, None] + nrm * 1e-2
u = rng.random((len(O), 2)).astype(np.float32)
r, a = np.sqrt(u[:, 0]), 2 * np.pi * u[:, 1]
t1 = np.cross(nrm, np.where(np.abs(nrm[:, :1]) > 0.9, [[0, 1, 0]], [[1, 0, 0]])); t1 /= np.linalg.norm(t1, axis=1, keepdims=True)
t2 = np.cross(nrm, t1)
D = (t1 * (r * np.cos(a))[:, None] + t2 * (r * np.sin(a))[:, None] + nrm * np.sqrt(np.maximum(0, 1 - r * r))[:, None]).astype(np.float32)
D /= np.linalg.norm(D, axis=1, keepdims=True)
R = np.concatenate([O, D, np.zeros((len(O), 1), np.float32), np.full((len(O), 1), 1e4, np.float32)], 1).astype(np.float32)
print("bounce rays", len(R), flush=True)
h2, c2 = g.trace_closest(R, ignore=tri.astype(np.int32), count=True)
print(f"  {c2.ms_trace:.3f} ms  {len(R) / c2.ms_trace / 1e6:.2f} G rays/s  nodes/ray {c2.node_visits / len(R):.2f}  tris/ray {c2.tri_tests / len(R):.2f}", flush=True)
