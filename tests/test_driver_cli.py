"""the reference's command line and frame monitor over the GPU core (python -m rgk_amd; src/main.cpp:20-260,
src/render_driver.cpp:49-139): progress fed from the device, preview, rotate + --no-overwrite frame claiming, forced timed
mode, single-pixel debug, checkpointed frames."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

SCENE = '''{ // a small scene for the command-line tests
    "output-file": "cli.exr", "output-width": 96, "output-height": 64, "multisample": 8, "rounds": 3, "recursion-max": 4, "clamp": 20,
    "camera": {"position": [0,1.2,5], "lookat": [0,0.8,0], "fov": 35},
    "materials": [{"name": "m", "brdf": "diffuse", "diffuse255": [255, 128, 0]},
                  {"name": "g", "brdf": "ltc_ggx_diffuse", "exponent": 200, "specular": [0.3,0.3,0.3], "diffuse": [0.4,0.4,0.5]},
                  {"name": "l", "brdf": "diffuse", "diffuse": [0.5,0.5,0.5], "emission": [9,9,8]}],
    "scene": [{"primitive": "cube", "material": "m", "translate": [0,0.5,0]},
              {"primitive": "plane", "material": "g", "scale": [4,1,4]},
              {"primitive": "plane", "material": "l", "scale": [0.5,1,0.5], "translate": [0,3,0], "rotate": [180, 0, 0]}],
    "sky": {"color": [0.3, 0.4, 0.6], "intensity": 0.5}
}'''


def test_monitor_formatting_follows_the_reference():
    from rgk_amd.monitor import LowPass, format_int5, format_int_thousands, format_percent, format_time
    assert [format_time(x) for x in (0.2, 59.4, 59.6, 125.0, 3725.4, 86400.0)] == ["0s", "59s", "1m 0s", "2m 5s", "1h 2m", "24h 0m"]
    assert format_percent(12.345) == "12.3%" and format_int5(0) == "00000" and format_int5(42) == "00042"
    assert format_int_thousands(1234567) == "1'234'567" and format_int_thousands(999) == "999"
    lp = LowPass(3)
    assert [lp.add(x) for x in (3.0, 6.0, 9.0, 12.0)] == [3.0, 4.5, 6.0, 9.0]
    assert lp.add(float("nan")) != lp.add(float("nan")) and lp.add(12.0) == 11.0     # NaNs are not stored


def run_cli(args, cwd):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0")
    return subprocess.run([sys.executable, "-m", "rgk_amd"] + args, cwd=cwd, env=env, capture_output=True, text=True, timeout=600)


@pytest.mark.gpu
def test_cli_renders_writes_resumes_and_claims_frames(tmp_path):
    from rgk_amd import render_driver as rd
    cfg = tmp_path / "s.json"
    cfg.write_text(SCENE)
    out = str(tmp_path / "cli.exr")
    # rounds mode with the monitor and a checkpoint: 3 rounds, image rewritten after each
    r = run_cli([str(cfg), "-D", str(tmp_path), "--checkpoint", str(tmp_path / "f.ck")], str(tmp_path))
    assert r.returncode == 0, r.stderr + r.stdout
    assert "Rendered 18432/18432 pixels, round 3/3" in r.stderr and "] 100.0%" in r.stderr and "Total frame rendering time" in r.stderr
    img = rd.read_exr(out)
    assert img.shape == (64, 96, 4) and img[..., :3].max() == 1.0 and (img[..., 3] == 1).all()
    # resumed from the finished checkpoint: nothing left to render, the same image again
    r2 = run_cli([str(cfg), "-D", str(tmp_path), "--checkpoint", str(tmp_path / "f.ck"), "-q"], str(tmp_path))
    assert r2.returncode == 0 and "Resumed from" not in r2.stdout           # -q: verbosity 1
    assert np.array_equal(rd.read_exr(out), img)
    # preview: quarter resolution, half the samples, .preview suffix; -s fixes the scale
    r3 = run_cli([str(cfg), "-D", str(tmp_path), "-p", "-s", "0.25", "-q", "-q"], str(tmp_path))
    assert r3.returncode == 0, r3.stderr
    pv = rd.read_exr(str(tmp_path / "cli.preview.exr"))
    assert pv.shape == (16, 24, 4)
    # single-pixel debug: the pixel alone, with the seed it has inside its tile -> the same value as in a whole first round
    r4 = run_cli([str(cfg), "-d", "40", "30"], str(tmp_path))
    assert r4.returncode == 0 and r4.stdout.startswith("pixel (40, 30): radiance") and "WARNING" not in r4.stdout, r4.stdout + r4.stderr
    vals = [float(x) for x in r4.stdout.split("radiance")[1].split("samples")[0].split()]
    from rgk_amd.config import load_config
    c = load_config(str(cfg))
    sc = rd.Scene(c.build_scene().to_desc())
    acc, cnt, _ = sc.render_round(c.get_camera(), c.get_params(), rd.generate_task_list(c.xres, c.yres))
    assert np.allclose(vals, acc[30, 40] / cnt[30, 40], rtol=1e-6, atol=0)
    # ... and the same pixel replayed on the CPU oracle ("-d X Y single-pixel trace replayed on rgk_cpu")
    from oracle import rgk_oracle
    from rgk_amd import capi
    home = next(tl for tl in rd.generate_task_list(c.xres, c.yres) if tl.x0 <= 40 < tl.x1 and tl.y0 <= 30 < tl.y1)
    one = (capi.Tile * 1)()
    one[0].x0, one[0].x1, one[0].y0, one[0].y1 = 40, 41, 30, 31
    one[0].seed = (home.seed + ((30 - home.y0) * (home.x1 - home.x0) + (40 - home.x0)) * 0x42424242) & 0xFFFFFFFF
    o = rgk_oracle.OracleScene(c.build_scene().to_desc())
    ao, co, _ = o.render_round(c.get_camera(), c.get_params(), one)
    assert np.allclose(vals, ao[30, 40] / co[30, 40], rtol=1e-6, atol=0)
    # rotate: frames are claimed by creating the file; an existing frame is skipped (--no-overwrite is implied)
    open(str(tmp_path / "cli.00001.exr"), "wb").close()
    r5 = run_cli([str(cfg), "-D", str(tmp_path), "-r", "--frames", "3", "-q"], str(tmp_path))
    assert r5.returncode == 0, r5.stderr
    assert "cli.00001.exr` exists, not overwriting." in r5.stdout
    f0, f2 = rd.read_exr(str(tmp_path / "cli.00000.exr")), rd.read_exr(str(tmp_path / "cli.00002.exr"))
    assert f0.shape == f2.shape == (64, 96, 4) and not np.array_equal(f0, f2)      # the camera moved
    assert os.path.getsize(str(tmp_path / "cli.00001.exr")) == 0
    # config errors are reported, not raised
    bad = tmp_path / "bad.json"
    bad.write_text('{"output-file": "x.exr"}')
    r6 = run_cli([str(bad)], str(tmp_path))
    assert r6.returncode == 1 and "Failed to load config file" in r6.stdout


@pytest.mark.gpu
def test_cli_compare_suffix_and_per_frame_checkpoints(tmp_path):
    """-c (src/main.cpp:129-131,196) and the checkpoint rules of ADVICE r2: with -r every frame has its own checkpoint (a
    finished frame's checkpoint says "all rounds done", so sharing one file left every later frame empty), and a checkpoint
    written for another camera / scene / parameter set is refused instead of being continued."""
    from rgk_amd import render_driver as rd
    cfg = tmp_path / "s.json"
    cfg.write_text(SCENE)
    r = run_cli([str(cfg), "-D", str(tmp_path), "-c", "-p", "-q"], str(tmp_path))
    assert r.returncode == 0, r.stderr + r.stdout
    assert os.path.exists(str(tmp_path / "cli.preview.cmp.exr")) and not os.path.exists(str(tmp_path / "cli.exr"))
    ck = str(tmp_path / "rot.ck")
    r = run_cli([str(cfg), "-D", str(tmp_path), "-r", "--frames", "2", "--checkpoint", ck, "-q"], str(tmp_path))
    assert r.returncode == 0, r.stderr + r.stdout
    f0, f1 = rd.read_exr(str(tmp_path / "cli.00000.exr")), rd.read_exr(str(tmp_path / "cli.00001.exr"))
    assert f0[..., :3].max() == 1.0 and f1[..., :3].max() == 1.0 and not np.array_equal(f0, f1)   # BOTH frames were rendered
    assert os.path.exists(str(tmp_path / "rot.00000.ck")) and os.path.exists(str(tmp_path / "rot.00001.ck")) and not os.path.exists(ck)
    # frame 1's checkpoint offered to frame 0's camera: refused, nothing rendered on top of it
    os.replace(str(tmp_path / "rot.00001.ck"), str(tmp_path / "one.ck"))
    r = run_cli([str(cfg), "-D", str(tmp_path), "--checkpoint", str(tmp_path / "one.ck")], str(tmp_path))
    assert r.returncode == 1 and "cannot resume" in r.stdout and "different scene, camera or parameter set" in r.stdout, r.stdout + r.stderr
    # ... while frame 0's own is accepted by the same camera
    os.replace(str(tmp_path / "rot.00000.ck"), str(tmp_path / "zero.ck"))
    r = run_cli([str(cfg), "-D", str(tmp_path), "--checkpoint", str(tmp_path / "zero.ck")], str(tmp_path))
    assert r.returncode == 0 and "Resumed from" in r.stdout and "3 rounds done" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_progress_is_fed_from_the_device(tmp_path):
    """rgk_scene_get_progress from a second thread while a round runs: stages only grow, end at stages == total, rounds count."""
    import threading
    import ctypes as C
    from rgk_amd import capi, render_driver as rd
    from rgk_amd.workloads import Workload
    wl = Workload("cornell-1024", spp=64)
    g = rd.Scene(wl.builder.to_desc())
    lib = capi.load_product()
    seen, stop = [], threading.Event()

    def poll():
        p = capi.Progress()
        while not stop.is_set():
            lib.rgk_scene_get_progress(g.h, C.byref(p))
            seen.append((p.busy, p.stage, p.stages, p.rounds))
    th = threading.Thread(target=poll)
    th.start()
    tiles = rd.generate_task_list(wl.xres, wl.yres)
    import torch
    acc = torch.zeros((wl.yres, wl.xres, 3), dtype=torch.float32, device="cuda:0"); cnt = torch.zeros((wl.yres, wl.xres), dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    g.render_round_device(wl.camera, wl.params(), tiles, acc.data_ptr(), cnt.data_ptr())
    stop.set(); th.join()
    p = capi.Progress()
    lib.rgk_scene_get_progress(g.h, C.byref(p))
    assert (p.busy, p.stage, p.stages, p.rounds, p.round_pixels, p.round_paths) == (0, 10, 10, 1, 1024 * 1024, 1024 * 1024 * 64)
    busy = [s for s in seen if s[0] == 1]
    stages = [s[1] for s in busy]
    assert stages == sorted(stages) and len(set(stages)) >= 3, sorted(set(stages))      # intermediate stages were observed, monotonically
