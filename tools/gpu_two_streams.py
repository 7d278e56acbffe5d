#!/usr/bin/env python3
"""Do the traversal launches (bound by the VALU) and the shading launches (waves waiting on memory half of their time) overlap
when two rounds run side by side?  Two rgk_scene objects in ONE process (each has its own stream and workspace) render the even
and the odd tiles of the benchmark frame from two host threads; compared with one scene rendering all tiles.  (Two PROCESSES
time-slice the card and gained 2.8 % in round 1; streams of one process can share the CUs.)"""
import os, sys, threading, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("RGK_WORKSPACE_GB", "48")
import numpy as np
import torch
from rgk_amd import capi, render_driver as rd
from rgk_amd.workloads import Workload

wl = Workload("sponza-1080p")
desc = wl.builder.to_desc()
tiles = rd.generate_task_list(wl.xres, wl.yres)
prm = wl.params()
dev = torch.device("cuda", 0)
def acc():
    return torch.zeros((wl.yres, wl.xres, 3), dtype=torch.float32, device=dev), torch.zeros((wl.yres, wl.xres), dtype=torch.int32, device=dev)
scenes = [rd.Scene(desc), rd.Scene(desc)]
halves = [(capi.Tile * ((len(tiles) + 1) // 2))(*tiles[0::2]), (capi.Tile * (len(tiles) // 2))(*tiles[1::2])]
bufs = [acc(), acc()]
def run(k, tl, rounds, delay=0.0):
    time.sleep(delay)
    for _ in range(rounds):
        scenes[k].render_round_device(wl.camera, prm, tl, bufs[k][0].data_ptr(), bufs[k][1].data_ptr())
R = 8
run(0, tiles, 1); run(1, halves[1], 1); torch.cuda.synchronize()
t0 = time.perf_counter(); run(0, tiles, R); torch.cuda.synchronize(); t_one = (time.perf_counter() - t0) / R
t0 = time.perf_counter(); run(0, halves[0], R); run(1, halves[1], R); torch.cuda.synchronize(); t_seq = (time.perf_counter() - t0) / R
res = {}
for delay in (0.0, 0.010, 0.020, 0.030):
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(k, halves[k], R, delay * k)) for k in (0, 1)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); res[delay] = (time.perf_counter() - t0 - delay) / R
print(f"RGK_TRACE_PER_CU={os.environ.get('RGK_TRACE_PER_CU', '8')}: one scene, all tiles: {t_one * 1e3:.1f} ms per round; two halves one after the other: {t_seq * 1e3:.1f}; side by side on two streams, second one started 0 / 10 / 20 / 30 ms later: " + " / ".join(f"{v * 1e3:.1f}" for v in res.values()))
