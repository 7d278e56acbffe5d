"""python -m rgk_amd [OPTIONS]... FILE  -- the reference's command line (src/main.cpp:20-260) over the MI355X core.

  -p, --preview     quarter resolution, half the samples (PREVIEW_DIMENTIONS_RATIO 4, PREVIEW_RAYS_RATIO 2); output gets a
                    `.preview` suffix
  -r, --rotate      renders 501 frames rotating the camera around the look-at point (10 s at 50 fps); implies --no-overwrite
  -t, --timed MIN   forces timed mode
  --no-overwrite    skips frames whose output file exists (several machines / processes sharing a directory claim frames this way)
  -s, --scale V     output brightness scale (default: auto, the brightest channel becomes 1)
  -D, --dir DIR     output directory
  -d, --debug X Y   renders only pixel (X, Y) and prints its radiance, sample count and ray counters
  -c, --compare     output gets a `.cmp` suffix (src/main.cpp:129-131,196: an image to put beside an earlier one)
  -v / -q           verbosity up / down (default 2)
  --checkpoint F    raw accumulator checkpoint: resumed from F if it exists, rewritten after every round.  With -r every
                    frame has its own (F gets the frame number as a suffix); a checkpoint written for another scene, camera
                    or parameter set is refused
  --device N        GPU to use (default 0)
FILE is a .json or .rtc scene config.  One process drives one GPU; several GPUs: python -m torch.distributed.run ... bench.py.
"""
import argparse
import os
import sys

from . import capi
from . import render_driver as rd
from .config import ConfigFileException, load_config
from .monitor import FrameMonitor, format_int5, format_percent

PREVIEW_DIMENSIONS_RATIO, PREVIEW_RAYS_RATIO = 4, 2  # src/global_config.hpp:10-11


def insert_file_suffix(path, suffix):
    """Utils::InsertFileSuffix (src/utils.cpp:79-82)."""
    name, ext = os.path.splitext(path)
    return f"{name}.{suffix}{ext}" if ext else f"{path}.{suffix}."


def frame_digest(sb, camera, params):
    """64-bit digest of what a frame is rendered from: geometry, materials, textures, lights, sky, camera and the PathTracer
    parameters.  Rides in the checkpoint header (rgk_accum_set_tag): resuming after any of them changed would silently mix
    two different images in one accumulator."""
    import hashlib
    h = hashlib.blake2b(digest_size=8)
    sb.finalize()
    for a in (sb.V, sb.N, sb.T, sb.UV, sb.F, sb.FM):
        if a is not None:
            h.update(a.tobytes())
    h.update(repr([(sorted(m.items())) for m in sb.materials]).encode())
    for t in sb.textures:
        h.update(repr((t["kind"], t.get("color"))).encode())
        for key in ("data", "lut"):
            if t.get(key) is not None:
                h.update(t[key].tobytes())
    h.update(repr((sb.pointlights, [list(a) for a in sb.areal], sorted((k, str(v)) for k, v in sb.sky.items()))).encode())
    h.update(bytes(camera))  # ctypes structures: their bytes
    h.update(bytes(params))
    return int.from_bytes(h.digest(), "little") or 1


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m rgk_amd", add_help=True, description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("file")
    ap.add_argument("-p", "--preview", action="store_true")
    ap.add_argument("-r", "--rotate", action="store_true")
    ap.add_argument("-t", "--timed", type=int, default=None)
    ap.add_argument("--no-overwrite", action="store_true")
    ap.add_argument("-s", "--scale", type=float, default=None)
    ap.add_argument("-D", "--dir", default="")
    ap.add_argument("-d", "--debug", nargs=2, type=int, metavar=("X", "Y"))
    ap.add_argument("-c", "--compare", action="store_true")
    ap.add_argument("-v", action="count", default=0)
    ap.add_argument("-q", action="count", default=0)
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--frames", type=int, default=None, help="with -r: stop after this many frames (default: all 501)")
    args = ap.parse_args(argv)
    verbosity = max(0, 2 + args.v - args.q)
    say = lambda lvl, *a: print(*a) if verbosity >= lvl else None
    try:
        cfg = load_config(args.file)
    except ConfigFileException as e:
        print("Failed to load config file:", e)
        return 1
    if args.timed is not None:
        if args.timed <= 0:
            print("ERROR: Invalid argument for -t (--time).")
            return 1
        cfg.render_minutes, cfg.render_rounds = args.timed, 1
    if args.scale is not None:
        cfg.output_scale = args.scale
    output_file = os.path.join(args.dir, cfg.output_file) if args.dir else cfg.output_file
    if args.preview:
        output_file = insert_file_suffix(output_file, "preview")
    if args.compare:
        output_file = insert_file_suffix(output_file, "cmp")  # main.cpp:196 (after the preview suffix, before the frame number)
    if args.preview:
        cfg.xres //= PREVIEW_DIMENSIONS_RATIO
        cfg.yres //= PREVIEW_DIMENSIONS_RATIO
        cfg.multisample = max(1, cfg.multisample // PREVIEW_RAYS_RATIO)
    try:
        sb = cfg.build_scene(asset_dir=os.environ.get("RGK_ASSET_DIR"))
    except (ConfigFileException, RuntimeError) as e:
        print("Failed to load data from config file:", e)
        return 1
    cfg.get_camera(0.0)  # main.cpp:221: the camera is read before the post-check, so its keys count as used
    for w in cfg.perform_post_check():
        say(1, w if w.startswith("WARNING") else f'WARNING: Unused key "{w}" in the config file.')
    import torch
    scene = rd.Scene(sb.to_desc(), device=args.device)
    device = torch.device("cuda", args.device)
    no_overwrite = args.no_overwrite or args.rotate
    fps, time_length = 50.0, (10.0 if args.rotate else 0.0)
    n_frames = int(time_length * fps) + 1
    for frame_no in range(n_frames if args.frames is None else min(n_frames, args.frames)):
        t = frame_no / fps
        out = insert_file_suffix(output_file, format_int5(frame_no)) if args.rotate else output_file
        if no_overwrite and os.path.exists(out):
            say(1, f"File `{out}` exists, not overwriting.")
            continue
        if args.rotate:
            open(out, "ab").close()  # claim the frame before rendering it (processes sharing the directory skip it)
            say(1, f"Rendering frame #{frame_no} of ~{int(time_length * fps)} ({format_percent(t / time_length * 100.0)})")
        camera = cfg.get_camera(t / time_length if args.rotate else 0.0)
        drv = rd.RenderDriver(scene, cfg, camera, device=device)
        if args.debug:
            x, y = args.debug
            if not (0 <= x < cfg.xres and 0 <= y < cfg.yres):
                print("ERROR: debug pixel outside the frame")
                return 1
            tile = (capi.Tile * 1)()
            # the pixel as its own 1 x 1 task with the seed RenderPixel would give it inside its 32 x 32 tile of round 0
            tiles = rd.generate_task_list(cfg.xres, cfg.yres, rd.SEEDSTART, 0)
            home = next(tl for tl in tiles if tl.x0 <= x < tl.x1 and tl.y0 <= y < tl.y1)
            k = (y - home.y0) * (home.x1 - home.x0) + (x - home.x0)
            tile[0].x0, tile[0].x1, tile[0].y0, tile[0].y1 = x, x + 1, y, y + 1
            tile[0].seed = (home.seed + k * 0x42424242) & 0xFFFFFFFF
            acc, cnt, c = scene.render_round(camera, drv.params, tile)
            px = acc[y, x] / max(1, int(cnt[y, x]))
            print(f"pixel ({x}, {y}): radiance {px[0]:.9g} {px[1]:.9g} {px[2]:.9g}  samples {int(cnt[y, x])}  path rays {c.path_rays}  shadow rays {c.shadow_rays}")
            return 0
        # one checkpoint per frame (a finished frame's checkpoint says "all rounds done": the next frame, seen from another
        # camera, would render nothing on top of it), tagged with what the frame is rendered from
        ckpt = (insert_file_suffix(args.checkpoint, format_int5(frame_no)) if args.rotate else args.checkpoint) if args.checkpoint else None
        drv.checkpoint_tag = frame_digest(sb, camera, drv.params)
        if ckpt and os.path.exists(ckpt):
            try:
                drv.load_checkpoint(ckpt)
            except RuntimeError as e:
                print(f"ERROR: cannot resume from `{ckpt}`: {e}")
                return 1
            say(2, f"Resumed from `{ckpt}`: {drv.rounds_done} rounds done.")
        say(2, f"Writing to file {out}")
        timed = cfg.render_minutes is not None
        with FrameMonitor(scene, timed, cfg.render_rounds, cfg.render_minutes or 0, cfg.xres * cfg.yres, verbosity=verbosity) as mon:
            drv.render_frame(rounds=max(0, cfg.render_rounds - drv.rounds_done) if not timed else None, output_file=out, checkpoint=ckpt)
            mon.rays_done = sum(c.path_rays for c in drv.counters)
    return 0


if __name__ == "__main__":
    sys.exit(main())
