#!/usr/bin/env python3
"""bench.py -- Mpaths/s of the path-tracing hot path on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload sponza-1080p]
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one RenderRound of the workload: every pixel of the frame gets `multisample`
paths (raygen -> [trace -> shade -> shadow]*depth -> resolve on the GPU), followed, for
N > 1, by the RCCL sum-reduce of the per-GPU accumulators to rank 0.  Inputs (scene, BVH,
textures, tile list) are resident in HBM before the timed region; the accumulator stays
on the device.

N = 1 workload = the configuration BASELINE.json's metric is quoted on: Sponza
1920x1080x256spp (scenes/sponza.json + overrides), on the labelled PROXY geometry because
sponza.obj is absent from the reference checkout.  Scaling is "strong" by
default, as the metric reads ("Sponza 1920x1080x256spp at 1/2/4/8 MI355X"): the frame and its
sample count stay fixed and the centre-out tile list is dealt round-robin, so with N GPUs each
traces 1/N of the tiles (`--scaling weak` takes N x the samples per pixel instead: fixed work
per GPU).

The JSON line also carries `roofline` -- the binding ceiling of the dominant kernel with
frac <= 1 (traversal: VALU issue; shading: HBM), per-kernel records under `kernels`, every
profile-derived number with its source -- and, at N = 1, `cpu_baseline` (the CPU oracle
timed on the host cores on a bounded tile sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def host_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
        except (OSError, ValueError, IndexError):
            pass
    cap = os.environ.get("RGK_CPU_THREADS")
    return int(cap) if cap else n


def child_environments(n, port, base=None):
    """The environment of each of the n ranks `python bench.py --gpus n` starts when it was not itself started by
    torch.distributed.run: what that launcher would have set (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_*)."""
    envs = []
    for r in range(n):
        e = dict(os.environ if base is None else base)
        e.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                  "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": e.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        envs.append(e)
    return envs


def self_launch(n, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start one child process per GPU (fresh interpreters,
    started BEFORE this process touches the GPU; nothing is exec'ed), pass rank 0's stdout through, the other ranks' stdout to
    stderr, and return non-zero if any rank does."""
    import socket
    import subprocess
    import threading
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r, env in enumerate(child_environments(n, port)):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE, stderr=None, text=True))

    def relay(p, dst):
        for line in p.stdout:
            # stdout carries the ONE JSON line and nothing else: whatever a library prints on a rank's stdout (gloo's connection
            # notes, for one) goes to stderr
            out = dst if (dst is sys.stdout and line.lstrip().startswith("{")) else sys.stderr
            out.write(line)
            out.flush()
    threads = [threading.Thread(target=relay, args=(p, sys.stdout if r == 0 else sys.stderr), daemon=True) for r, p in enumerate(procs)]
    for t in threads:
        t.start()
    try:
        while any(p.poll() is None for p in procs):
            if any(p.poll() not in (None, 0) for p in procs):  # one rank died: the others would wait in a collective for ever
                for q in procs:
                    if q.poll() is None:
                        q.terminate()
                break
            time.sleep(0.05)
        rcs = [p.wait(timeout=30) for p in procs]
    except subprocess.TimeoutExpired:
        rcs = [p.poll() if p.poll() is not None else -9 for p in procs]
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
    for t in threads:
        t.join(timeout=5)
    bad = [(r, c) for r, c in enumerate(rcs) if c != 0]
    if bad:
        print(f"bench.py: rank(s) failed: {bad}", file=sys.stderr)
        return 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="sponza-1080p")
    ap.add_argument("--scale", type=float, default=1.0, help="resolution scale (testing only; invalidates the number)")
    ap.add_argument("--spp", type=int, default=None, help="override samples per pixel (testing only)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong")
    ap.add_argument("--emulate-shard", type=int, default=0, help="diagnostic: ONE rank's share of a K-GPU strong-scaled round (tiles i %% K == 0) on this GPU, no reduce; not a benchmark number")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--textures", choices=["u8", "f32"], default="u8", help="f32: hand every image texture over as floats, as a binding to the "
                    "reference does (FileTexture keeps only decoded floats); the core then stores those with <= 256 distinct values as bytes itself")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; 'gloo' + RGK_FORCE_DEVICE=0 rehearses N ranks on one GPU")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started as plain `python bench.py --gpus N` (how the driver starts it): become the launcher -- nothing below has
        # touched the GPU yet -- and let N fresh children be the ranks
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    if os.environ.get("RGK_BENCH_ECHO_ENV"):  # launcher test (tests/test_host_cpu.py): what a rank was given, no GPU needed
        print(json.dumps({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")} | {"argv": sys.argv[1:]}))
        raise SystemExit(int(os.environ.get("RGK_BENCH_ECHO_FAIL_RANK", "-1")) == int(os.environ.get("RANK", "0")))

    import numpy as np
    import torch
    from rgk_amd import capi
    from rgk_amd import render_driver as rd
    from rgk_amd.workloads import Workload

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:  # under a launcher the launcher's rank count is the truth
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if os.environ.get("RGK_FORCE_DEVICE") is not None:  # rehearsal: several ranks share one card
        local_rank = int(os.environ["RGK_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=args.backend)

    wl = Workload(args.workload, scale=args.scale, spp=args.spp)
    base_spp = wl.multisample
    if args.scaling == "weak" and world > 1:
        wl.multisample = base_spp * world
    if args.textures == "f32":
        for t in wl.builder.textures:
            if t["kind"] == capi.TEX_RGB8:
                t["data"] = np.ascontiguousarray(t["lut"][t["data"]], dtype=np.float32)
                t["kind"], t["lut"] = capi.TEX_RGB32F, None
    scene = rd.Scene(wl.builder.to_desc(), device=local_rank)
    info = scene.info()

    class Cfg:  # what RenderDriver needs from Config
        xres, yres, render_rounds, render_minutes = wl.xres, wl.yres, 1, None

        @staticmethod
        def get_params(sampler=capi.SAMPLER_HALTON, flags=0):
            return wl.params(sampler, flags)

    drv = rd.RenderDriver(scene, Cfg, wl.camera, rank=rank, world_size=world, device=device, flags=capi.FLAG_TIME_KERNELS,
                          host_reduce=(args.backend != "nccl"))

    if args.emulate_shard > 1:  # what one rank of a K-GPU run computes per round (its fixed per-round costs included)
        drv.rank, drv.world_size = 0, args.emulate_shard
        import types

        def _round(self, reduce=True):
            tiles = rd.generate_task_list(self.cfg.xres, self.cfg.yres, rd.SEEDSTART, self.seedcount)
            self.seedcount += len(tiles)
            mine = rd.shard_tiles(tiles, 0, args.emulate_shard)
            torch.cuda.current_stream(self.device).synchronize()
            return self.scene.render_round_device(self.camera, self.params, mine, self.total_ob.data.data_ptr(), self.total_ob.count.data_ptr())
        drv.render_round = types.MethodType(_round, drv)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        drv.render_round()
    barrier()
    t0 = time.perf_counter()
    cnts = [drv.render_round() for _ in range(args.steps)]
    barrier()
    elapsed = time.perf_counter() - t0
    cdev = device if args.backend == "nccl" else torch.device("cpu")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # whole-job unit counts (all ranks)
    local = np.array([sum(c.paths for c in cnts), sum(c.path_rays for c in cnts), sum(c.shadow_rays for c in cnts)], dtype=np.float64)
    if world > 1:
        t = torch.tensor(local, dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        tot = t.cpu().numpy()
    else:
        tot = local
    paths, path_rays, shadow_rays = tot

    # ---- a frame's FIRST round (BASELINE configs[2] is `rounds: 1`): the timed steps above are later rounds of one frame -- camera-ray
    # entry lists capped behind round 1's first hits, per-frame lists cached.  Resetting the per-frame state (what a new camera
    # or tile list does) makes the next round pay for all of it: uncapped lists, the distance ranges, the list builds.
    first_round_ms = None
    if world == 1 and not args.emulate_shard:
        fr = []
        for _ in range(3):
            scene.set_tuning(entry_points=1)   # (same value: the call drops the per-frame lists)
            barrier()
            t1 = time.perf_counter()
            drv.render_round()
            barrier()
            fr.append((time.perf_counter() - t1) * 1e3)
        first_round_ms = round(sorted(fr)[1], 3)

    # ---- rooflines.  LIVE in this run: HIP-event time, launches and units (rays / vertices) PER KERNEL inside the timed steps
    # (rgk_counters.kernel[], events on the stream the kernels run on), and each kernel's node visits / triangle tests from one
    # extra untimed counting round (deterministic per seed) -> SURVEY 8(d)'s algorithmic bytes per launch.  FROM THE COMMITTED
    # PROFILE of this same command (profiles/<tag>_<workload>_roofline.json: tools/profile_round.sh + tools/summarize_prof.py): the
    # VALU instruction mix, lane utilisation and the physical HBM bytes per launch (rocprofv3 --pmc in separate passes; bench.py
    # cannot run the profiler on itself) -- attached only when workload and N match the profiled run, always with their source.
    KID = capi.KERNEL_IDS
    P_l, R_l, S_l = (sum(c.paths for c in cnts), sum(c.path_rays for c in cnts), sum(c.shadow_rays for c in cnts))  # this rank
    drv_c = rd.RenderDriver(scene, Cfg, wl.camera, rank=rank, world_size=world, device=device, flags=capi.FLAG_COUNT_TRAVERSAL,
                            host_reduce=(args.backend != "nccl"))
    cc = drv_c.render_round(reduce=False)
    nodes_per_ray = cc.node_visits / max(1, cc.path_rays)
    tris_per_ray = cc.tri_tests / max(1, cc.path_rays)
    bytes_per_ray = 32 + 4 + 16 + nodes_per_ray * info.node_bytes + tris_per_ray * info.tri_bytes
    # SURVEY 8(d) per shaded vertex: 56 path state + 96 vertex attributes + 32 material (+ 4*12*n_maps texels, n_maps = 2, + 3*12 bump
    # + 4*20 LTC entries for a textured, bump-mapped LTC material: the Sponza kind) -- 184 B for Cornell's solid diffuse, 396 B otherwise
    shade_bytes = 184.0 if args.workload.startswith("cornell") else 396.0
    prof, prof_name = None, None
    try:
        if args.scale == 1.0 and args.spp is None and world == 1 and not args.emulate_shard:
            cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith(f"_{args.workload}_roofline.json"))
            if cands:
                prof_name = cands[-1]
                prof = json.load(open(os.path.join(ROOT, "profiles", prof_name)))["kernels"]
    except Exception:
        prof = None

    def prof_rec(kid):
        """The profiled record of one kernel: the non-counting instantiation whose name starts with the kernel's prefix."""
        pre = capi.KERNEL_NAMES.get(kid)
        if not prof or not pre:
            return None
        recs = [r for k, r in prof.items() if k.startswith(pre) and "<true, 256" not in k and "<true, 32" not in k]
        if not recs:
            return None
        n = sum(r["launches_profiled"] for r in recs)
        w = lambda key: sum((r.get(key) or 0.0) * r["launches_profiled"] for r in recs) / n
        return {"avg_ms": w("avg_ms"), "valu_slow_pipe_cycles": w("valu_slow_pipe_cycles"), "valu_issue_cycles": w("valu_issue_cycles"),
                "lane_util": w("lane_util"), "hbm_bytes_per_launch": w("hbm_bytes_per_launch") or None, "launches_profiled": n}

    kernels, step_hbm_bytes, step_hbm_covered_ms = [], 0.0, 0.0
    for i, kid in enumerate(KID):
        ms_k = sum(c.kernel[i].ms for c in cnts)
        n_k = sum(c.kernel[i].launches for c in cnts)
        u_k = sum(c.kernel[i].units for c in cnts)
        if n_k == 0:
            continue
        avg_s = ms_k / n_k * 1e-3
        rec = {"kernel": kid, "ms_per_step": round(ms_k / args.steps, 3), "launches_per_step": round(n_k / args.steps, 2),
               "avg_launch_ms": round(ms_k / n_k, 4), "units_per_step": int(u_k / args.steps)}
        ck = cc.kernel[i]
        alg = None   # SURVEY 8(d) algorithmic bytes per unit
        if kid in ("trace_camera", "trace_closest", "light_trace") and ck.units:
            rec["nodes_per_ray"], rec["tris_per_ray"] = round(ck.node_visits / ck.units, 2), round(ck.tri_tests / ck.units, 2)
            alg = 32 + 4 + 16 + ck.node_visits / ck.units * info.node_bytes + ck.tri_tests / ck.units * info.tri_bytes
        elif kid in ("shadow_first", "shadow", "light_splat") and ck.units:
            rec["nodes_per_ray"], rec["tris_per_ray"] = round(ck.node_visits / ck.units, 2), round(ck.tri_tests / ck.units, 2)
            alg = 32 + 4 + ck.node_visits / ck.units * info.node_bytes + ck.tri_tests / ck.units * info.tri_bytes
        elif kid in ("shade_first", "shade"):
            alg = shade_bytes
        elif kid == "resolve":
            alg = 16.0
        if alg is not None and u_k:
            per_launch = alg * u_k / n_k
            rec.update({"algorithmic_bytes_per_unit": round(alg, 1), "algorithmic_bytes_per_launch": round(per_launch),
                        "algorithmic_GBps": round(per_launch / avg_s / 1e9, 1), "algorithmic_frac_of_hbm_peak": round(per_launch / avg_s / 8.0e12, 4)})
        pr = prof_rec(kid)
        if pr:
            rec.update({"source": f"profiles/{prof_name}", "profile_avg_launch_ms": round(pr["avg_ms"], 4),
                        # VALU: the non-fp32 ("slow") pipe's busy cycles from the profiled instruction mix (tools/summarize_prof.py) against
                        # 1024 SIMDs x 2.4 GHz x the LIVE launch time
                        "valu_slow_pipe_frac": round(pr["valu_slow_pipe_cycles"] / (1024 * 2.4e9 * avg_s), 4),
                        "valu_issue_frac": round(pr["valu_issue_cycles"] / (1024 * 2.4e9 * avg_s), 4), "lane_util": round(pr["lane_util"], 4)})
            if pr.get("hbm_bytes_per_launch"):
                # physical bytes: 128 B per TCC_EA0_RDREQ (= FETCH_SIZE x 2: every read request of these kernels, gathers included, is a
                # 128-byte line fill -- profiles/r03_fetch_calibration.txt) + WRITE_SIZE
                rec.update({"hbm_bytes_per_launch": pr["hbm_bytes_per_launch"], "hbm_frac": round(pr["hbm_bytes_per_launch"] / avg_s / 8.0e12, 4)})
                step_hbm_bytes += pr["hbm_bytes_per_launch"] * n_k / args.steps
                step_hbm_covered_ms += ms_k / args.steps
        kernels.append(rec)
    ms_step = elapsed / args.steps * 1e3
    dom = max((r for r in kernels if r["kernel"] not in ("other",)), key=lambda r: r["ms_per_step"])
    traversal = dom["kernel"] in ("trace_camera", "trace_closest", "shadow_first", "shadow", "shadow_jobs", "light_trace", "light_splat")
    if traversal:   # bounded by the VALU's non-fp32 pipe, not by HBM: its bytes are cache-served
        bound, unit, peak = "valu", "G SIMD-cycles/s of the non-fp32 VALU pipe", 1024 * 2.4
        frac = dom.get("valu_slow_pipe_frac")
        achieved = frac * peak if frac is not None else None
    else:           # shading / resolve: gathers + queue records, bounded by the memory system
        bound, unit, peak = "hbm", "GB/s", 8000.0
        achieved = (dom["hbm_bytes_per_launch"] / (dom["avg_launch_ms"] * 1e-3) / 1e9) if dom.get("hbm_bytes_per_launch") else dom.get("algorithmic_GBps")
        frac = achieved / peak if achieved else None
    roofline = {"bound": bound, "kernel": capi.KERNEL_NAMES.get(dom["kernel"], dom["kernel"]), "kernel_id": dom["kernel"],
                "avg_launch_ms": dom["avg_launch_ms"], "achieved": round(achieved, 1) if achieved else None, "peak": peak, "unit": unit,
                "frac": round(frac, 4) if frac is not None else None,
                "traffic": dom.get("hbm_bytes_per_launch"),
                "traffic_source": f"profiles/{prof_name}: rocprofv3 --pmc, 128 B x TCC_EA0_RDREQ (= FETCH_SIZE x 2, calibrated for gathers too: profiles/r03_fetch_calibration.txt) + WRITE_SIZE" if dom.get("hbm_bytes_per_launch") else None,
                "hbm_frac": dom.get("hbm_frac"), "lane_util": dom.get("lane_util"),
                # SURVEY 8(d)'s figure for the same kernel: algorithmic bytes per launch over the live launch time.  Node and triangle
                # bytes are served by L1 / L2 / Infinity Cache, so for traversal this exceeds what HBM could deliver: it is the
                # algorithmic rate, NOT an HBM fraction (the physical one is hbm_frac)
                "algorithmic": {k: dom.get(k) for k in ("algorithmic_bytes_per_unit", "algorithmic_bytes_per_launch", "algorithmic_GBps", "algorithmic_frac_of_hbm_peak", "nodes_per_ray", "tris_per_ray")},
                # the north-star target (>= 40 % of the HBM roofline on Sponza 1080p x 256) is judged on THIS: physical bytes of all
                # kernels of a step over the step's wall time
                "step_hbm_frac": round(step_hbm_bytes / (ms_step * 1e-3) / 8.0e12, 4) if step_hbm_bytes else None,
                "step_hbm_GB": round(step_hbm_bytes / 1e9, 2) if step_hbm_bytes else None,
                "step_algorithmic_GB": round(sum(r.get("algorithmic_bytes_per_launch", 0) * r["launches_per_step"] for r in kernels) / 1e9, 1),
                "first_round_ms": first_round_ms,
                "whole_round": {"nodes_per_path_ray": round(nodes_per_ray, 2), "tris_per_path_ray": round(tris_per_ray, 2), "bytes_per_path_ray": round(bytes_per_ray, 1),
                                "node_bytes": info.node_bytes, "tri_bytes": info.tri_bytes},
                "kernels": kernels}

    metric_name = {"sponza-1080p": "Sponza 1920x1080x256spp", "cornell-1024": "Cornell box 1024x1024x256spp", "cornell-256": "Cornell box 256x256x16spp",
                   "dragon-sponza-1080p": "Dragon-Sponza 1920x1080x512spp reverse 3", "sponza4-2160p": "Sponza4 3840x2160x1024spp"}[args.workload]
    out = {
        "metric": f"Mpaths/s, {metric_name} (path-tracing hot path)",
        "value": round(paths / elapsed / 1e6, 2), "unit": "Mpaths/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload} {wl.xres}x{wl.yres}x{wl.multisample}spp depth {wl.depth} russian {wl.russian:.2f}, {args.scaling} scaling"
                               + ("" if args.scale == 1.0 and args.spp is None else " (REDUCED SIZE: not the benchmark number)"),
                   "geometry": wl.geometry, "triangles": int(len(wl.builder.F)), "sampler": "halton-cp",
                   "textures": f"handed over as {args.textures}; {info.n_palettized_textures} of {info.n_float_textures} float textures stored as bytes + value table",
                   "tiles": drv.n_tasks, "parallelism": f"tiles round-robin over {world} GPU(s) + one RCCL reduce per round"},
        "mrays_per_s_path": round(path_rays / elapsed / 1e6, 2),
        "mrays_per_s_all": round((path_rays + shadow_rays) / elapsed / 1e6, 2),
        "roofline": roofline,
    }

    # ---- CPU baseline: the oracle on the host cores, rank 0 at N = 1 only, bounded sample
    if world == 1 and not args.no_cpu_baseline:
        from oracle import rgk_oracle as O
        osc = O.OracleScene(wl.builder.to_desc())
        threads = max(1, host_cores() - 1)  # hardware_concurrency() - 1, render_driver.cpp:205-206
        otiles = O.generate_task_list(wl.xres, wl.yres)
        prm = wl.params()

        def run(n_tiles, sampler):
            p = wl.params(sampler)
            sub = (capi.Tile * n_tiles)(*[otiles[i] for i in range(n_tiles)])
            t = time.perf_counter()
            _, _, c = osc.render_round(wl.camera, p, sub, n_threads=threads)
            return c, time.perf_counter() - t
        n_probe = min(len(otiles), threads)
        c, dt = run(n_probe, capi.SAMPLER_HALTON)
        n_tiles = int(max(n_probe, min(len(otiles), n_probe * args.cpu_seconds / max(dt, 1e-3))))
        n_tiles = max(threads, (n_tiles // threads) * threads)
        n_tiles = min(n_tiles, len(otiles))
        c, dt = run(n_tiles, capi.SAMPLER_HALTON)
        cs, dts = run(max(threads, n_tiles // 4), capi.SAMPLER_STRATIFIED)
        cpu_model, sockets = "unknown", None
        try:
            ci = open("/proc/cpuinfo").read()
            cpu_model = next(ln.split(":", 1)[1].strip() for ln in ci.splitlines() if ln.startswith("model name"))
            sockets = len({ln.split(":", 1)[1].strip() for ln in ci.splitlines() if ln.startswith("physical id")}) or None
        except (OSError, StopIteration):
            pass
        out["cpu_baseline"] = {
            "value": round(c.paths / dt / 1e6, 3), "unit": "Mpaths/s", "cores": threads, "kind": "port",
            "cpu_model": cpu_model, "sockets": sockets, "logical_cpus_visible": os.cpu_count(),
            "sample": f"first {n_tiles} of {len(otiles)} centre-out 32x32 tiles at {wl.multisample} spp, {dt:.1f} s, shared Halton sampler",
            "mrays_per_s_path": round(c.path_rays / dt / 1e6, 3),
            "faithful_sampler_value": round(cs.paths / dts / 1e6, 3),
            "faithful_sampler_note": "same tiles/4 with the reference's per-pixel mt19937 StratifiedSampler table (src/sampler.cpp:85-116)",
        }
        del prm
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
