#!/usr/bin/env python3
"""What would binning the bounce-1 ray queue buy (VERDICT r1 #5a)?  Diffuse bounce rays off the Sponza proxy's camera hits,
traced by the product's closest-hit kernel in four orders: slot order (what the pipeline has), stable bin by direction octant,
by octant x major axis (24 bins), and fully sorted by (octant, Morton code of the origin) -- an upper bound no cheap binning reaches.
Prints kernel ms, rays/s and node visits per ray for each."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from rgk_amd import render_driver as rd
from rgk_amd.workloads import Workload

wl = Workload("sponza-1080p")
sb = wl.builder; sb.finalize()
g = rd.Scene(sb.to_desc())
W, H = wl.xres, wl.yres
cam = wl.camera
# camera rays through pixel centres, pixels in 8x8-block order inside 32x32 tiles like the product's slots
ys, xs = np.mgrid[0:H, 0:W]
key = ((ys // 32) * (W // 32 + 1) + (xs // 32)) * 1024 + ((ys % 32) // 8 * 4 + (xs % 32) // 8) * 64 + (ys % 8) * 8 + (xs % 8)
order = np.argsort(key.ravel(), kind="stable")
px, py = xs.ravel()[order], ys.ravel()[order]
vs, vx, vy, org = (np.array(list(getattr(cam, n)), np.float32) for n in ("viewscreen", "viewscreen_x", "viewscreen_y", "origin"))
p = vs[None] + ((px + 0.5) / W)[:, None].astype(np.float32) * vx[None] + ((py + 0.5) / H)[:, None].astype(np.float32) * vy[None]
d = p - org[None]; d /= np.linalg.norm(d, axis=1, keepdims=True)
n = len(d)
rays = np.concatenate([np.tile(org, (n, 1)), d, np.zeros((n, 1), np.float32), np.full((n, 1), 1e4, np.float32)], 1).astype(np.float32)
hits, _ = g.trace_closest(rays)
ok = hits["tri"] >= 0
V, F = sb.V, sb.F
tri = hits["tri"][ok]
e1, e2 = V[F[tri, 1]] - V[F[tri, 0]], V[F[tri, 2]] - V[F[tri, 0]]
nrm = np.cross(e1, e2); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
dd = d[ok]
nrm = np.where((np.sum(nrm * dd, 1) > 0)[:, None], -nrm, nrm)
pos = org[None] + dd * hits["t"][ok][:, None] + nrm * 1e-2
rng = np.random.default_rng(0)
reps = 6
O = np.repeat(pos, reps, axis=0)   # consecutive samples of a pixel sit in different passes in the product; here side by side: origin coherence is an upper bound
Nn = np.repeat(nrm, reps, axis=0)
u = rng.random((len(O), 2)).astype(np.float32)
r, a = np.sqrt(u[:, 0]), 2 * np.pi * u[:, 1]
t1 = np.cross(Nn, np.where(np.abs(Nn[:, :1]) > 0.9, [[0, 1, 0]], [[1, 0, 0]])); t1 /= np.linalg.norm(t1, axis=1, keepdims=True)
t2 = np.cross(Nn, t1)
D = (t1 * (r * np.cos(a))[:, None] + t2 * (r * np.sin(a))[:, None] + Nn * np.sqrt(np.maximum(0, 1 - r * r))[:, None]).astype(np.float32)
D /= np.linalg.norm(D, axis=1, keepdims=True)
# product order: sample-major (all pixels of sample 0, then sample 1, ...): emulate by interleaving
idx = np.arange(len(O)).reshape(-1, reps).T.ravel()
O, D = O[idx], D[idx]
R = np.concatenate([O, D, np.zeros((len(O), 1), np.float32), np.full((len(O), 1), 1e4, np.float32)], 1).astype(np.float32)
octant = (D[:, 0] < 0) * 1 + (D[:, 1] < 0) * 2 + (D[:, 2] < 0) * 4
major = np.argmax(np.abs(D), axis=1)
lo, hi = O.min(0), O.max(0)
q = np.clip(((O - lo) / (hi - lo) * 1023).astype(np.uint64), 0, 1023)
def spread(v):
    v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249; return v
morton = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
orders = {"slot order": np.arange(len(R)), "octant bins (stable)": np.argsort(octant, kind="stable"),
          "octant x major axis (24 bins)": np.argsort(octant * 3 + major, kind="stable"),
          "octant, then Morton(origin)": np.lexsort((morton, octant))}
print("bounce rays", len(R))
if os.environ.get("ONLY_SLOT"):
    orders = {"slot order": orders["slot order"]}
for name, o in orders.items():
    best = None
    for _ in range(3):
        _, c = g.trace_closest(R[o], count=True)
        best = c.ms_trace if best is None else min(best, c.ms_trace)
    print(f"{name:34s} {best:8.3f} ms  {len(R) / best / 1e6:7.2f} G rays/s  nodes/ray {c.node_visits / len(R):.2f} tris/ray {c.tri_tests / len(R):.2f}")
