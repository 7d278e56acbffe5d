#!/usr/bin/env python3
"""8-bit textures (bytes + byte->float table) against the same texels handed over as floats: identical images expected."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from rgk_amd import capi, render_driver as rd
from rgk_amd.workloads import SceneFixture
from oracle import rgk_oracle as O

def as_float(sb):
    for t in sb.textures:
        if t["kind"] == capi.TEX_RGB8:
            t["data"] = np.ascontiguousarray(t["lut"][t["data"]], dtype=np.float32)
            t["kind"], t["lut"] = capi.TEX_RGB32F, None
    return sb

for name in ("rubiks-bump",):
    path = os.path.join(ROOT, "tests", "golden", f"scene_{name}.npz")
    a, b = SceneFixture(path, scale=0.1, spp=4, depth=4), SceneFixture(path, scale=0.1, spp=4, depth=4)
    as_float(b.builder)
    prm = a.params()
    res = {}
    for tag, wl in (("u8", a), ("f32", b)):
        desc = wl.builder.to_desc()
        g, o = rd.Scene(desc), O.OracleScene(desc)
        res["gpu_" + tag] = g.render_round(wl.camera, prm, rd.generate_task_list(prm.xres, prm.yres))[0]
        res["orc_" + tag] = o.render_round(wl.camera, prm, O.generate_task_list(prm.xres, prm.yres))[0]
    for x, y in (("gpu_u8", "gpu_f32"), ("orc_u8", "orc_f32"), ("gpu_u8", "orc_u8"), ("gpu_f32", "orc_f32")):
        d = np.abs(res[x] - res[y])
        print(name, x, "vs", y, "equal", np.array_equal(res[x], res[y]), "max abs", float(d.max()), "differing pixels", int((d.max(axis=2) > 0).sum()), "of", d.shape[0] * d.shape[1])
    bad = np.argwhere((np.abs(res["gpu_u8"] - res["gpu_f32"]).max(axis=2) > 0))
    print("first differing pixels (y, x):", bad[:10].tolist())
