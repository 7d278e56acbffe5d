"""Host-side mirror of the reference's render driver, over the C ABI.

Mirrors (names, argument meaning, behaviour) reference src/render_driver.cpp:
  GenerateTaskList :30-46, RenderDriver::RenderRound :144-190, RenderDriver::RenderFrame
  :192-253 (rounds mode and timed mode), and EXRTexture's accumulator half
  (src/texture.cpp:342-354,376-412: AddPixel / GetPixel / Normalize / Accumulate).

The reference runs tiles on a CPU thread pool inside one process.  Here the per-task
body runs on the GPU behind `rgk_render_round*`; with world_size > 1 (one process per
GPU, torch.distributed) tile i of the centre-out list goes to rank i % world_size and
the per-GPU private accumulators are summed with one reduce per round
(EXRTexture::Accumulate under the mutex, render_driver.cpp:179-182, becomes the RCCL
reduce).  Seeds depend only on (round, tile index, pixel-in-tile): the image does not
depend on the number of GPUs.
"""
import ctypes as C
import time

import numpy as np

from . import capi

TILE_SIZE = 32  # src/global_config.hpp:8
SEEDSTART = 42  # src/render_driver.cpp:222


def generate_task_list(xres, yres, seedstart=SEEDSTART, seedcount_base=0, tile_size=TILE_SIZE, mid=None):
    """GenerateTaskList + the `seedstart + c` each task's PathTracer gets."""
    lib = capi.load_product()
    mid = mid or (xres / 2.0, yres / 2.0)
    n = C.c_uint32(0)
    capi.check(lib, lib.rgk_generate_task_list(tile_size, xres, yres, mid[0], mid[1], seedstart, seedcount_base, None, C.byref(n)))
    tiles = (capi.Tile * n.value)()
    capi.check(lib, lib.rgk_generate_task_list(tile_size, xres, yres, mid[0], mid[1], seedstart, seedcount_base, tiles, C.byref(n)))
    return tiles


def shard_tiles(tiles, rank, world_size):
    """Round-robin deal of the centre-out list: tile i -> rank i % world_size (rgk_shard_tiles)."""
    lib = capi.load_product()
    n = C.c_uint32(0)
    capi.check(lib, lib.rgk_shard_tiles(tiles, len(tiles), rank, world_size, None, C.byref(n)))
    out = (capi.Tile * n.value)()
    capi.check(lib, lib.rgk_shard_tiles(tiles, len(tiles), rank, world_size, out, C.byref(n)))
    return out


class Scene:
    """Device-resident committed scene (rgk_scene_create)."""

    def __init__(self, builder_or_desc, device=0):
        self.lib = capi.load_product()
        self._builder = builder_or_desc if hasattr(builder_or_desc, "to_desc") else None
        desc = builder_or_desc.to_desc() if self._builder is not None else builder_or_desc
        h = C.c_void_p()
        capi.check(self.lib, self.lib.rgk_scene_create(C.byref(desc), device, C.byref(h)))
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.rgk_scene_destroy(self.h)
            self.h = None

    __del__ = close

    def set_tuning(self, **kw):
        """rgk_scene_set_tuning: per-scene tuning switches (entry_points, entry_cap, light_entry, sample_group, batch_paths,
        workspace_gb); none changes a result."""
        for k, v in kw.items():
            capi.check(self.lib, self.lib.rgk_scene_set_tuning(self.h, k.encode(), float(v)))
        return self

    def refit(self, vertices, normals=None, tangents=None):
        """rgk_scene_refit: moved vertices, same triangles -- records, epsilon, box, light tables recomputed, the tree refit."""
        arr = [None if a is None else np.ascontiguousarray(a, dtype=np.float32) for a in (vertices, normals, tangents)]
        capi.check(self.lib, self.lib.rgk_scene_refit(self.h, *[None if a is None else a.ctypes.data for a in arr]))
        return self

    def info(self):
        i = capi.SceneInfo()
        capi.check(self.lib, self.lib.rgk_scene_get_info(self.h, C.byref(i)))
        return i

    def trace_closest(self, rays, ignore=None, count=False):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = len(rays)
        ig = None if ignore is None else np.ascontiguousarray(ignore, dtype=np.int32)
        hits = np.zeros(n, dtype=[("t", "f4"), ("tri", "i4"), ("a", "f4"), ("b", "f4"), ("c", "f4")])
        cnt = capi.Counters()
        capi.check(self.lib, self.lib.rgk_trace_closest(self.h, n, rays.ctypes.data, None if ig is None else ig.ctypes.data,
                                                        hits.ctypes.data, C.byref(cnt) if count else None))
        return hits, cnt

    def visibility(self, a, b, count=False):
        a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 3)
        b = np.ascontiguousarray(b, dtype=np.float32).reshape(-1, 3)
        vis = np.zeros(len(a), dtype=np.uint8)
        cnt = capi.Counters()
        capi.check(self.lib, self.lib.rgk_trace_visibility(self.h, len(a), a.ctypes.data, b.ctypes.data, vis.ctypes.data,
                                                           C.byref(cnt) if count else None))
        return vis, cnt

    def render_round(self, camera, params, tiles, accum=None, count=None):
        """Host-buffer entry point: accum (yres, xres, 3) float32 +=, count (yres, xres) uint32 +=."""
        if accum is None:
            accum = np.zeros((params.yres, params.xres, 3), dtype=np.float32)
            count = np.zeros((params.yres, params.xres), dtype=np.uint32)
        cnt = capi.Counters()
        capi.check(self.lib, self.lib.rgk_render_round(self.h, C.byref(camera), C.byref(params), tiles, len(tiles),
                                                       accum.ctypes.data, count.ctypes.data, C.byref(cnt)))
        return accum, count, cnt

    def render_round_device(self, camera, params, tiles, d_accum_ptr, d_count_ptr):
        cnt = capi.Counters()
        capi.check(self.lib, self.lib.rgk_render_round_device(self.h, C.byref(camera), C.byref(params), tiles, len(tiles),
                                                              d_accum_ptr, d_count_ptr, C.byref(cnt)))
        return cnt


def sampler_eval(seed, index, dim, is2d):
    lib = capi.load_product()
    seed = np.ascontiguousarray(seed, dtype=np.uint32)
    index = np.ascontiguousarray(index, dtype=np.uint32)
    dim = np.ascontiguousarray(dim, dtype=np.uint32)
    out = np.zeros((len(seed), 2), dtype=np.float32)
    capi.check(lib, lib.rgk_sampler_eval(len(seed), seed.ctypes.data, index.ctypes.data, dim.ctypes.data, int(is2d), out.ctypes.data))
    return out


def read_exr(path):
    """Minimal reader for the files rgk_output_write_exr writes (uncompressed scan-line half RGBA): (h, w, 4) float32
    in R, G, B, A order.  Test infrastructure for the round trip, not a general OpenEXR reader."""
    import struct
    b = open(path, "rb").read()
    assert struct.unpack_from("<ii", b, 0) == (20000630, 2)
    pos, attrs = 8, {}
    while b[pos] != 0:
        e = b.index(b"\0", pos); name = b[pos:e].decode(); pos = e + 1
        e = b.index(b"\0", pos); typ = b[pos:e].decode(); pos = e + 1
        (size,) = struct.unpack_from("<i", b, pos); pos += 4
        attrs[name] = (typ, b[pos:pos + size]); pos += size
    pos += 1
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    assert attrs["compression"][1] == b"\0"
    chans, cp, cl = [], 0, attrs["channels"][1]
    while cl[cp] != 0:
        e = cl.index(b"\0", cp); chans.append(cl[cp:e].decode()); cp = e + 1 + 16
    offs = struct.unpack_from("<%dQ" % h, b, pos)
    img = np.zeros((h, w, 4), np.float32)
    for y in range(h):
        yy, n = struct.unpack_from("<ii", b, offs[y])
        line = np.frombuffer(b, dtype=np.float16, count=w * len(chans), offset=offs[y] + 8).reshape(len(chans), w)
        for k, c in enumerate(chans):
            img[yy - y0, :, "RGBA".index(c)] = line[k]
    return img


class EXRTexture:
    """The Radiance accumulator (reference src/texture.hpp:83-118); device-resident torch tensors."""

    def __init__(self, xsize, ysize, device):
        import torch
        self.xsize, self.ysize = xsize, ysize
        self.data = torch.zeros((ysize, xsize, 3), dtype=torch.float32, device=device)
        self.count = torch.zeros((ysize, xsize), dtype=torch.int32, device=device)  # bit pattern of uint32

    def get_pixels(self):
        """EXRTexture::GetPixel for every pixel: data / count, 0 where count == 0 (texture.cpp:349-354)."""
        import torch
        c = self.count.to(torch.float32).unsqueeze(-1)
        return torch.where(c > 0, self.data / c.clamp(min=1), torch.zeros_like(self.data))

    def normalize(self, val):
        """EXRTexture::Normalize (texture.cpp:376-400): val <= 0 -> scale so the brightest channel is 1 (Q17)."""
        px = self.get_pixels()
        if val <= 0.0:
            val = 1.0 / float(px.max())
        return px * val

    def write(self, path, output_scale=-1.0):
        """total_ob.Normalize(cfg->output_scale).Write(output_file) (render_driver.cpp:233,245) through the C ABI:
        rgk_output_normalize + rgk_output_write_exr on host copies of the accumulator.  Returns the scale used."""
        lib = capi.load_product()
        acc = np.ascontiguousarray(self.data.cpu().numpy(), dtype=np.float32)
        cnt = np.ascontiguousarray(self.count.cpu().numpy()).view(np.uint32)
        out = np.empty_like(acc)
        val = C.c_float(0.0)
        capi.check(lib, lib.rgk_output_normalize(acc.ctypes.data, cnt.ctypes.data, self.xsize, self.ysize, float(output_scale),
                                                 out.ctypes.data, C.byref(val)))
        capi.check(lib, lib.rgk_output_write_exr(str(path).encode(), self.xsize, self.ysize, out.ctypes.data))
        return val.value


class RenderDriver:
    """RenderDriver::RenderFrame / RenderRound for one process per GPU."""

    def __init__(self, scene, cfg, camera, rank=0, world_size=1, device=None, sampler=capi.SAMPLER_HALTON, flags=0,
                 host_reduce=False):
        import torch
        self.scene, self.cfg, self.camera = scene, cfg, camera
        self.rank, self.world_size = rank, world_size
        self.device = device if device is not None else torch.device("cuda", scene.device)
        self.params = cfg.get_params(sampler=sampler, flags=flags)
        self.tasks = generate_task_list(cfg.xres, cfg.yres, SEEDSTART, 0)
        self.n_tasks = len(self.tasks)
        self.seedcount = 0
        self.total_ob = EXRTexture(cfg.xres, cfg.yres, self.device)
        self.rounds_done = 0
        self.counters = []
        self.host_reduce = host_reduce  # gloo rehearsal: reduce through host copies instead of RCCL
        self.round_ob = None
        self.clock = time.time
        self.checkpoint_tag = 0  # digest of scene + camera + parameters (rgk_accum_set_tag); 0: checkpoints are not compared

    def render_round(self, reduce=True):
        """One RenderRound: every rank renders its tiles into its private accumulator, then ONE sum-reduce of the RGB
        accumulator to rank 0 (no data-path collective inside the round).  Sample counts are not exchanged: every tile
        list covers the frame and every pixel of it gains `multisample` samples per round, splats add none
        (tracer.cpp:18,25), so rank 0 adds that constant itself."""
        import torch
        tiles = generate_task_list(self.cfg.xres, self.cfg.yres, SEEDSTART, self.seedcount)
        self.seedcount += len(tiles)  # `c = seedcount++` per task, render_driver.cpp:160
        if self.world_size == 1:
            torch.cuda.current_stream(self.device).synchronize()
            cnt = self.scene.render_round_device(self.camera, self.params, tiles, self.total_ob.data.data_ptr(), self.total_ob.count.data_ptr())
        else:
            mine = shard_tiles(tiles, self.rank, self.world_size)
            if self.round_ob is None:  # one private accumulator per rank for the whole frame, cleared per round
                self.round_ob = EXRTexture(self.cfg.xres, self.cfg.yres, self.device)
            else:
                self.round_ob.data.zero_()
                self.round_ob.count.zero_()
            ob = self.round_ob
            torch.cuda.current_stream(self.device).synchronize()
            cnt = self.scene.render_round_device(self.camera, self.params, mine, ob.data.data_ptr(), ob.count.data_ptr())
            if reduce:
                import torch.distributed as dist
                if self.host_reduce:
                    hd = ob.data.cpu()
                    dist.reduce(hd, dst=0, op=dist.ReduceOp.SUM)
                    if self.rank == 0:
                        self.total_ob.data += hd.to(self.total_ob.data.device)
                else:
                    dist.reduce(ob.data, dst=0, op=dist.ReduceOp.SUM)
                    if self.rank == 0:
                        self.total_ob.data += ob.data
                if self.rank == 0:
                    self.total_ob.count += int(self.params.multisample)
        self.rounds_done += 1
        self.counters.append(cnt)
        return cnt

    def save_checkpoint(self, path):
        """Raw-accumulator checkpoint (rgk_accum_save): accumulator + rounds done + the running task counter."""
        lib = capi.load_product()
        acc = C.c_void_p()
        capi.check(lib, lib.rgk_accum_create(self.cfg.xres, self.cfg.yres, self.scene.device, C.byref(acc)))
        try:
            a = np.ascontiguousarray(self.total_ob.data.cpu().numpy(), dtype=np.float32)
            c = np.ascontiguousarray(self.total_ob.count.cpu().numpy()).view(np.uint32)
            capi.check(lib, lib.rgk_accum_upload(acc, a.ctypes.data, c.ctypes.data))
            capi.check(lib, lib.rgk_accum_set_tag(acc, self.checkpoint_tag))
            capi.check(lib, lib.rgk_accum_save(acc, str(path).encode(), self.rounds_done, self.seedcount))
        finally:
            lib.rgk_accum_destroy(acc)

    def load_checkpoint(self, path):
        """Resume: the next render_round continues the seed sequence where the saved run stopped."""
        import torch
        lib = capi.load_product()
        acc = C.c_void_p()
        capi.check(lib, lib.rgk_accum_create(self.cfg.xres, self.cfg.yres, self.scene.device, C.byref(acc)))
        try:
            rd_, sc_ = C.c_uint32(0), C.c_uint32(0)
            capi.check(lib, lib.rgk_accum_set_tag(acc, self.checkpoint_tag))
            capi.check(lib, lib.rgk_accum_load(acc, str(path).encode(), C.byref(rd_), C.byref(sc_)))
            a = np.empty((self.cfg.yres, self.cfg.xres, 3), np.float32)
            c = np.empty((self.cfg.yres, self.cfg.xres), np.uint32)
            capi.check(lib, lib.rgk_accum_download(acc, a.ctypes.data, c.ctypes.data))
        finally:
            lib.rgk_accum_destroy(acc)
        self.total_ob.data.copy_(torch.from_numpy(a))
        self.total_ob.count.copy_(torch.from_numpy(c.view(np.int32)))
        self.rounds_done, self.seedcount = rd_.value, sc_.value

    def _continue_timed(self, t0, minutes):
        go = (self.clock() - t0) / 60.0 < minutes
        if self.world_size > 1:
            import torch
            import torch.distributed as dist
            flag = torch.tensor([1 if go else 0], dtype=torch.int32, device="cpu" if self.host_reduce else self.device)
            dist.broadcast(flag, src=0)
            go = bool(flag.item())
        return go

    def render_frame(self, rounds=None, minutes=None, output_file=None, checkpoint=None):
        """RenderFrame: Rounds mode (render_driver.cpp:229-235) or Timed mode (:237-247); with `output_file` the
        normalised image is rewritten after every round, as the reference does (rank 0 only)."""
        rounds = self.cfg.render_rounds if rounds is None else rounds
        minutes = self.cfg.render_minutes if minutes is None else minutes
        t0 = self.clock()

        def one():
            self.render_round()
            if output_file and self.rank == 0:
                self.total_ob.write(output_file, getattr(self.cfg, "output_scale", -1.0))
            if checkpoint and self.rank == 0:
                self.save_checkpoint(checkpoint)
        if minutes is None:
            for _ in range(rounds):
                one()
        else:
            # Timed mode (render_driver.cpp:237-247).  With several ranks the decision to start another round is rank 0's,
            # broadcast to all: ranks reading their own clocks could disagree and leave a reduce unmatched.
            while self._continue_timed(t0, minutes):
                one()
        return self.total_ob
