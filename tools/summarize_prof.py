#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof_<tag>_*, written by tools/profile_round.sh) into the small files kept
under profiles/ -- the evidence bench.py's `roofline` object points at.

  python tools/summarize_prof.py <round-tag>

  profiles/<tag>_kernel_stats.csv   the --kernel-trace --stats table of the default bench command (kernel names shortened)
  profiles/<tag>_hbm_traffic.csv    per kernel: FETCH_SIZE / WRITE_SIZE (separate --pmc passes) and bytes per launch with
                                    the gfx950 correction FETCH x 2 of MI355X_MICROARCH.md (HBM section)
  profiles/<tag>_roofline.json      per kernel: average launch duration, the VALU instruction mix (SQ_INSTS_VALU_* classes),
                                    the VALU ISSUE time it implies under the measured issue costs below, the shader cycles the
                                    kernel had (GRBM_GUI_ACTIVE / 8 XCDs, same pass), lane utilisation, HBM bytes per launch

VALU model (tools/micro/valu_peak.hip on MI355X, profiles/<tag>_valu_issue_costs.txt; wall time x measured clock, 8 waves per
SIMD, independent registers): a wave64 instruction costs one SIMD
    ~2.4 cycles  fp32 add / sub / mul / fma / fmac (modifiers included), v_mov_b32              -- the SIMD-32 rate
    ~3.2 cycles  32-bit integer add / sub, and / or / xor / not, right shifts
    ~4.3 cycles  everything else measured: conversions, min / max / max3 / med3, compares, selects, bfe / bfi / perm, left
                 shifts, lshl_add / lshl_or / and_or, integer multiplies, floor / trunc / fract, ldexp, div_scale / fmas / fixup,
                 dpp moves, fp64 add / mul / fma, packed fp32
    ~9.2 cycles  transcendentals (rcp, sqrt, ...);  ~16.4 v_rcp_f64
and fp32 arithmetic OVERLAPS with the slower class (4 fma + 4 cvt interleaved: 2.5 cycles per instruction, not 3.4): a SIMD
has an issue port (2 cycles per instruction) and a slower, 16-lane-wide path that the non-fp32 instructions occupy for 4.
So two ceilings per kernel, both from the same counter pass and both <= 1 by construction of the model (nominal costs 2 / 3 / 4 / 8):
    valu_issue_frac     = 2 x SQ_INSTS_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
    valu_slow_pipe_frac = (4 x (other + fp64) + 3 x int32/64 + 8 x trans) / (1024 x GRBM_GUI_ACTIVE / 8),
                          other = SQ_INSTS_VALU - fp32 add/mul/fma - trans - fp64 - int   (conversions, compares, selects, min/max, ...)
    lane_util           = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU-weighted instructions) ~ SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU)
The traversal kernels sit at 0.92-0.96 of the slow-pipe ceiling: they are VALU-bound on the instructions that are NOT fp32
arithmetic (byte -> float conversions of the quantised planes, min / max of the slab test, the selects and compares of the
child sort), with 54-78 % of the lanes doing useful work.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
N_SIMD, N_XCD, HBM_PEAK = 1024, 8, 8.0e12
COST = {"issue_per_instruction": 2, "slow_other_and_fp64": 4, "slow_int": 3, "slow_trans": 8}


def short(name):
    return name.replace("void ", "").split("(")[0]


def find(d, suffix):
    f = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return max(f, key=os.path.getmtime) if f else None  # gpurun MERGES a run's files into gpurun_out/: older runs' files stay beside the new ones


def counters(d):
    """{kernel: {counter: (sum, launches)}} of one --pmc pass directory."""
    out = collections.OrderedDict()
    f = find(d, "_counter_sums.csv")  # per-kernel sums made on the GPU box by tools/profile_round.sh (the per-dispatch table is tens of MB)
    if f:
        for r in csv.DictReader(open(f)):
            c = out.setdefault(short(r["Kernel_Name"]), {})
            v, n = c.get(r["Counter_Name"], (0.0, 0))
            c[r["Counter_Name"]] = (v + float(r["Counter_Value"]), n + int(r["Dispatches"]))
        return out
    f = find(d, "_counter_collection.csv")
    if not f:
        return out
    disp = collections.defaultdict(lambda: collections.defaultdict(set))
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        c = out.setdefault(k, collections.defaultdict(float))
        c[r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
    return collections.OrderedDict((k, {c: (v, len(disp[k][c])) for c, v in cs.items()}) for k, cs in out.items())


def main():
    tag = sys.argv[1]
    base = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_")
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    stats = {}
    ks = find(base + "stats", "_kernel_stats.csv")
    with open(ks) as f, open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w", newline="") as g:
        w = csv.writer(g)
        for i, row in enumerate(csv.reader(f)):
            if i:
                row[0] = short(row[0])
                stats[row[0]] = {"calls": int(row[1]), "avg_ms": float(row[3]) / 1e6, "pct": float(row[4])}
            w.writerow(row)
    merged = collections.OrderedDict()
    for d in sorted(x for x in glob.glob(base + "pmc*") if os.path.isdir(x)):
        for k, cs in counters(d).items():
            m = merged.setdefault(k, {})
            has_classes = "SQ_INSTS_VALU_FMA_F32" in cs  # the pass whose shader cycles the issue fraction is taken against
            for c, vn in cs.items():
                if c == "GRBM_GUI_ACTIVE" and has_classes:
                    m[c] = vn
                else:
                    m.setdefault(c, vn)
    with open(os.path.join(out, f"{tag}_hbm_traffic.csv"), "w", newline="") as g:
        w = csv.writer(g)
        # read bytes: every memory-side read request of these kernels is a 128-byte line fill, sparse gathers included
        # (profiles/r03_fetch_calibration.txt: TCC_EA0_RDREQ_32B = 0, a second touch of a gathered line hits in L2), while FETCH_SIZE
        # tallies a request at 64 bytes -- so read bytes = 128 x TCC_EA0_RDREQ = 2 x FETCH_SIZE.  Both are listed when both were collected.
        w.writerow(["kernel", "launches", "FETCH_SIZE_KB_sum", "WRITE_SIZE_KB_sum", "read_MB_per_launch_x2corrected", "write_MB_per_launch", "hbm_MB_per_launch",
                    "TCC_EA0_RDREQ_per_launch", "TCC_EA0_RDREQ_32B_per_launch", "read_MB_per_launch_128B_x_RDREQ", "L2_hit_rate"])
        for k, m in merged.items():
            if "FETCH_SIZE" not in m or "WRITE_SIZE" not in m:
                continue
            (fv, fn), (wv, wn) = m["FETCH_SIZE"], m["WRITE_SIZE"]
            rd, wr = 2.0 * fv * 1024 / max(fn, 1) / 1e6, wv * 1024 / max(wn, 1) / 1e6
            rq = m.get("TCC_EA0_RDREQ_sum"); r32 = m.get("TCC_EA0_RDREQ_32B_sum"); hit = m.get("TCC_HIT_sum"); miss = m.get("TCC_MISS_sum")
            rqpl = rq[0] / max(rq[1], 1) if rq else None
            w.writerow([k, max(fn, wn), round(fv, 1), round(wv, 1), round(rd, 2), round(wr, 2), round(rd + wr, 2),
                        round(rqpl, 1) if rq else "", round(r32[0] / max(r32[1], 1), 1) if r32 else "",
                        round((128.0 * (rqpl - r32[0] / max(r32[1], 1)) + 32.0 * r32[0] / max(r32[1], 1)) / 1e6, 2) if rq and r32 else "",
                        round(hit[0] / max(hit[0] + miss[0], 1), 4) if hit and miss else ""])
    roof = {"tag": tag, "command": "python bench.py --steps 1 --warmup 0 --no-cpu-baseline (one rocprofv3 --pmc pass per counter group); "
                                   "durations from the --kernel-trace --stats pass of bench.py --steps 3 --warmup 1",
            "valu_issue_cost_cycles": COST, "n_simd": N_SIMD, "hbm_peak_GBps": HBM_PEAK / 1e9, "kernels": {}}
    for k, m in merged.items():
        if not k.startswith("k_") or k not in stats or "SQ_INSTS_VALU" not in m:
            continue
        per = lambda c: (m[c][0] / max(m[c][1], 1)) if c in m else 0.0
        insts = per("SQ_INSTS_VALU")
        fast = per("SQ_INSTS_VALU_ADD_F32") + per("SQ_INSTS_VALU_MUL_F32") + per("SQ_INSTS_VALU_FMA_F32")
        trans = per("SQ_INSTS_VALU_TRANS_F32")
        f64 = per("SQ_INSTS_VALU_ADD_F64") + per("SQ_INSTS_VALU_MUL_F64") + per("SQ_INSTS_VALU_FMA_F64")
        ints = per("SQ_INSTS_VALU_INT32") + per("SQ_INSTS_VALU_INT64")
        other = max(insts - fast - trans - f64 - ints, 0.0)
        issue = COST["issue_per_instruction"] * insts
        slow = COST["slow_other_and_fp64"] * (other + f64) + COST["slow_int"] * ints + COST["slow_trans"] * trans
        cyc = per("GRBM_GUI_ACTIVE") / N_XCD
        lane = per("SQ_THREAD_CYCLES_VALU") / (64.0 * insts) if insts else 0.0
        rec = {"launches_profiled": m["SQ_INSTS_VALU"][1], "avg_ms": round(stats[k]["avg_ms"], 4), "pct_of_gpu_time": stats[k]["pct"],
               "insts_valu": insts, "insts_f32_add_mul_fma": fast, "insts_trans_f32": trans, "insts_f64": f64, "insts_int": ints, "insts_other": other,
               "valu_issue_cycles": issue, "valu_slow_pipe_cycles": slow, "shader_cycles": cyc,
               "valu_issue_frac": round(issue / (N_SIMD * cyc), 4) if cyc else None,
               "valu_slow_pipe_frac": round(slow / (N_SIMD * cyc), 4) if cyc else None, "lane_util": round(min(lane, 1.0), 4),
               "valu_lane_frac": round(slow / (N_SIMD * cyc) * min(lane, 1.0), 4) if cyc else None}
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            b = 2.0 * per("FETCH_SIZE") * 1024 + per("WRITE_SIZE") * 1024
            rec["hbm_bytes_per_launch"] = b
            if "TCC_EA0_RDREQ_sum" in m:  # the same read bytes from the raw request counters: 128 B per request (+ 32 B ones, if any)
                r32 = per("TCC_EA0_RDREQ_32B_sum")
                rec["read_bytes_per_launch_from_rdreq"] = 128.0 * (per("TCC_EA0_RDREQ_sum") - r32) + 32.0 * r32
                rec["rdreq_32B_share"] = round(r32 / max(per("TCC_EA0_RDREQ_sum"), 1.0), 6)
                if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m:
                    rec["l2_hit_rate"] = round(per("TCC_HIT_sum") / max(per("TCC_HIT_sum") + per("TCC_MISS_sum"), 1.0), 4)
            rec["hbm_GBps"] = round(b / (stats[k]["avg_ms"] * 1e-3) / 1e9, 1)
            rec["hbm_frac"] = round(b / (stats[k]["avg_ms"] * 1e-3) / HBM_PEAK, 4)
        roof["kernels"][k] = rec
    json.dump(roof, open(os.path.join(out, f"{tag}_roofline.json"), "w"), indent=1)
    for k, r in roof["kernels"].items():
        print(f"{k:34s} {r['avg_ms']:9.3f} ms  valu issue {r['valu_issue_frac']} slow pipe {r['valu_slow_pipe_frac']}  lane {r['lane_util']}  hbm {r.get('hbm_frac')}")
    micro = os.path.join(ROOT, "gpurun_out", "r2_valu_peak.txt")
    if os.path.exists(micro):
        open(os.path.join(out, f"{tag}_valu_issue_costs.txt"), "w").write(open(micro).read())


if __name__ == "__main__":
    main()
