"""The frame monitor of the reference's render driver (FrameMonitorThread, src/render_driver.cpp:49-139) for a GPU round:
a thread that reads the scene's device-fed progress (rgk_scene_get_progress: a host callback queued behind every bounce of every
pass) every 100 ms and prints the two-line progress bar with elapsed time and a low-pass-filtered ETA.  Formatting helpers follow
src/utils.cpp:94-160 (FormatTime, FormatPercent, FormatInt5, FormatIntThousands, LowPass)."""
import collections
import ctypes as C
import sys
import threading
import time

from . import capi

BARSIZE = 75  # src/global_config.hpp:14


def format_time(s):
    seconds = int(s + 0.5)
    minutes, hours = seconds // 60, seconds // 3600
    minutes %= 60
    seconds %= 60
    out = ""
    if hours > 0:
        out += f"{hours}h "
    if hours > 0 or minutes > 0:
        out += f"{minutes}m "
    if hours == 0:
        out += f"{seconds}s "
    return out.rstrip(" ")


def format_percent(f):
    return f"{f:.1f}%"


def format_int5(i):
    return "%05d" % i


def format_int_thousands(v):
    return f"{int(v):,}".replace(",", "'")


class LowPass:
    def __init__(self, size):
        self.data = collections.deque(maxlen=size)

    def add(self, value):
        if value != value:  # do not store NaNs
            return value
        self.data.append(value)
        return sum(self.data) / len(self.data)


class FrameMonitor:
    """with FrameMonitor(scene, mode, limit_rounds, limit_minutes, pixels_per_round): drv.render_frame(...)"""

    def __init__(self, scene, timed, limit_rounds, limit_minutes, pixels_per_round, out=sys.stderr, verbosity=2, period=0.1):
        self.scene, self.timed = scene, timed
        self.limit_rounds, self.limit_minutes, self.ppr = limit_rounds, limit_minutes, pixels_per_round
        self.out, self.verbosity, self.period = out, verbosity, period
        self.rounds_base = self._progress().rounds
        self.eta_lp = LowPass(40)
        self.rays_done = 0
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)
        self.lines = []  # what was printed last (tests read it)

    def _progress(self):
        p = capi.Progress()
        lib = capi.load_product()
        capi.check(lib, lib.rgk_scene_get_progress(self.scene.h, C.byref(p)))
        return p

    def pixels_done(self):
        p = self._progress()
        rounds = p.rounds - self.rounds_base
        part = (self.ppr * p.stage // p.stages) if (p.busy and p.stages) else 0
        return rounds * self.ppr + part, rounds

    def _print(self, final=False):
        elapsed = time.time() - self.t0
        pixels_done, rounds_done = self.pixels_done()
        mask_eta = False
        if not self.timed:
            total = self.ppr * self.limit_rounds
            fraction = pixels_done / float(total)
            eta = self.eta_lp.add((1.0 - fraction) * elapsed / fraction) if fraction > 0 else float("nan")
            pixels_text = f"Rendered {pixels_done:>{len(str(total))}}/{total} pixels"
            rounds_text = f"round {min(rounds_done + 1, self.limit_rounds)}/{self.limit_rounds}"
            mask_eta = (fraction < 0.03 and elapsed < 20.0) or eta != eta
        else:
            fraction = min(1.0, elapsed / 60.0 / self.limit_minutes)
            eta = max(0.0, self.limit_minutes * 60 - elapsed)
            pixels_text = f"Rendered {pixels_done} pixels"
            rounds_text = f"round {rounds_done + (0 if final else 1)}"
        percent = int(fraction * 1000.0 + 0.5) / 10.0
        if final:
            eta = 0.0
            if not self.timed:
                fraction, percent = 1.0, 100.0
        fill = int(fraction * BARSIZE)
        l1 = "[" + "#" * fill + "-" * (BARSIZE - fill) + "] " + format_percent(percent)
        l2 = f"{pixels_text}, {rounds_text}, time elapsed: {format_time(elapsed)}, ETA: {'???' if mask_eta else format_time(eta)}"
        self.lines = [l1, l2]
        if self.verbosity >= 1:
            self.out.write("\033[1A\33[2K\r" + l1 + "\n\33[2K\r" + l2)
            self.out.flush()

    def _run(self):
        while not self._stop.wait(self.period):
            self._print()

    def __enter__(self):
        self.t0 = time.time()
        if self.verbosity >= 1:
            self.out.write("\n\n")
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._thread.join()
        self._print(final=True)
        total = time.time() - self.t0
        pixels_done, _ = self.pixels_done()
        if self.verbosity >= 1:
            self.out.write("\n")
        if self.verbosity >= 2:
            self.out.write(f"Total frame rendering time: {format_time(total)}\n")
            self.out.write(f"Average pixels per second: {format_int_thousands(pixels_done / max(total, 1e-9))}.\n")
        if self.verbosity >= 3:
            self.out.write(f"Total rays: {self.rays_done}\nAverage rays per second: {format_int_thousands(self.rays_done / max(total, 1e-9))}\n")
        return False
