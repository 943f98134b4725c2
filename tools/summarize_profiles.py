#!/usr/bin/env python3
"""Reduce the rocprofv3 outputs of tools/collect_profiles.sh (gpurun_out/<tag>/) to the summaries kept under profiles/:

    <tag>_bench_kernel_stats.csv   per-kernel totals of `bench.py` (rocprofv3 --kernel-trace --stats)
    <tag>_conv_hbm_traffic.json    HBM bytes of the FCN kernels from the FETCH_SIZE / WRITE_SIZE passes at 8 / 24 / 72 frames
    <tag>_conv_mfma_pmc.json       matrix-core busy fraction, issue / wait split and clock per convolution kernel
    <tag>_vote_pmc.json            VALU issue fraction, HBM bytes and clock per voting kernel (read by bench.py)

    python tools/summarize_profiles.py gpurun_out/r02p1 r02
Units (guide /opt/skills/guides/MI355X_MICROARCH.md): FETCH_SIZE / WRITE_SIZE are KB, and gfx950 tallies 64 of every 128 B of wide
coalesced reads, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024; SQ_*_CYCLES other than SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES
count quad-cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs (kernel cycles = GRBM_GUI_ACTIVE / 8)."""
import glob
import json
import os
import re
import shutil
import sys

import pandas as pd

SRC, TAG = sys.argv[1], sys.argv[2]
DT = sys.argv[3] if len(sys.argv) > 3 else "f16x3"   # FCN arithmetic of the traffic / matrix-core passes
FB = int(sys.argv[4]) if len(sys.argv) > 4 else 53   # their frames per step
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "profiles")
N_SIMD = 256 * 4


def short(n):
    mm = re.search(r'k_convI(\w+?)Li(\d+)ELi(\d+)', n)
    if mm:
        return 'k_conv<%s,%s,%s>' % ("16-bit" if ("16b" in mm.group(1) or "DF16" in mm.group(1)) else "f32", mm.group(2), mm.group(3))
    mc2 = re.search(r'k_chain2_x3<(\d+)', n) or re.search(r'k_chain2_x3ILi(\d+)', n)
    if mc2:
        return 'k_chain2_x3<%s>' % mc2.group(1)
    mc = re.search(r'k_chain_x3<(\d+)>', n) or re.search(r'k_chain_x3ILi(\d+)', n)
    if mc:
        return 'k_chain_x3<%s>' % mc.group(1)
    mb = re.search(r'k_conv_block<(\d+), (\w+), (\w+)>', n)
    if mb:
        return 'k_conv_block<%s,%s,%s>' % mb.groups()
    mh = re.search(r'k_conv_halo<(\d+), (\d+)(?:, (\w+))?(?:, (\w+))?>', n)
    if mh:
        return 'k_conv_halo<%s,%s>%s' % (mh.group(1), mh.group(2), ' tail' if mh.group(4) == 'true' else ' q6' if mh.group(3) == 'true' else '')
    ml = re.search(r'k_bb_level<(\w+), (\w+), (\d+)>', n)
    if ml:
        return 'k_bb_level<%s,%s>' % ml.groups()[:2]
    m = re.search(r'(k_[a-z_0-9]+)', n)
    return m.group(1) if m else n[:40]


def counters(name):
    f = glob.glob(os.path.join(SRC, name, "*", "*counter_collection.csv"))
    if not f:
        return None
    d = pd.read_csv(f[0])
    d["k"] = d.Kernel_Name.map(short)
    d["dur_ns"] = d.End_Timestamp - d.Start_Timestamp
    return d.pivot_table(index=["Dispatch_Id", "k", "dur_ns"], columns="Counter_Name", values="Counter_Value").reset_index()


def log_tail(name):
    try:
        lines = [ln for ln in open(os.path.join(SRC, name + ".log")).read().splitlines() if not re.match(r"^[EWI]20\d\d", ln)]
        return lines[-1] if lines else ""
    except OSError:
        return ""


# ---- 1. kernel stats of the bench command
st = glob.glob(os.path.join(SRC, "stats", "*", "*kernel_stats.csv"))
if st:
    shutil.copy(st[0], os.path.join(OUT, TAG + "_bench_kernel_stats.csv"))

# ---- 2. HBM traffic of the FCN kernels
traffic = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/fcn_bench.py --dtype %s --batch B --iters 1 "
                     "(3 forwards of 3 networks each: 2 warm-up + 1 timed)" % DT, "dtype": DT,
           "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 64 of every 128 B of wide coalesced reads; both counters in KB)",
           "batches": {}}
for B in (12, FB):
    f, w = counters("fetch_b%d" % B), counters("write_b%d" % B)
    if f is None or w is None:
        continue
    fk, wk = f.groupby("k").FETCH_SIZE.sum(), w.groupby("k").WRITE_SIZE.sum()
    n_fw = int(f.k.isin(["k_tail", "k_tail_x3"]).sum())
    conv = [k for k in fk.index if k.startswith("k_conv") or k.startswith("k_chain") or k.startswith("k_tail")]
    allk = [k for k in fk.index if k.startswith("k_")]
    tot = lambda ks: float((2 * fk[ks].sum() + wk.reindex(ks).fillna(0).sum()) * 1024)
    traffic["batches"][str(B)] = {
        "forwards_profiled": n_fw, "networks": 3,
        "conv_family_hbm_bytes_per_network_forward": tot(conv) / n_fw / (3 * B),
        "all_fcn_kernels_hbm_bytes_per_network_forward": tot(allk) / n_fw / (3 * B),
        "per_kernel_bytes_per_network_forward": {k: float((2 * fk[k] + wk.get(k, 0)) * 1024) / n_fw / (3 * B) for k in allk},
    }
if str(FB) in traffic["batches"]:
    traffic["conv_family"] = {"frames_per_step": FB,
                              "hbm_bytes_per_network_forward": traffic["batches"][str(FB)]["conv_family_hbm_bytes_per_network_forward"]}
    json.dump(traffic, open(os.path.join(OUT, TAG + "_conv_hbm_traffic_%s.json" % DT), "w"), indent=1)

# ---- 3. matrix-core utilisation of the convolution kernels
a, b = counters("mfma"), counters("mfma2")
if a is not None:
    rows = {}
    ga = a.groupby("k").mean(numeric_only=True)
    gb = b.groupby("k").mean(numeric_only=True) if b is not None else None
    for k, r in ga.iterrows():
        if not (k.startswith("k_conv") or k.startswith("k_chain") or k in ("k_tail", "k_tail_x3", "k_stem_mfma")):
            continue
        cyc = r.GRBM_GUI_ACTIVE / 8.0
        e = {"dispatches": int((a.k == k).sum()), "avg_us": r.dur_ns / 1e3, "clock_ghz": cyc / r.dur_ns,
             "mfma_busy_frac": r.SQ_VALU_MFMA_BUSY_CYCLES / (cyc * N_SIMD),
             "wave_cycles_per_simd_cycle": 4 * r.SQ_WAVE_CYCLES / (cyc * N_SIMD)}
        if gb is not None and k in gb.index:
            q = gb.loc[k]
            e.update({"mfma_instructions": q.SQ_INSTS_MFMA, "active_frac_of_wave_time": q.SQ_ACTIVE_INST_ANY / r.SQ_WAVE_CYCLES,
                      "parked_frac_of_wave_time (s_waitcnt / barrier)": q.SQ_WAIT_ANY / r.SQ_WAVE_CYCLES,
                      "issue_stall_frac_of_wave_time (MFMA pipe / dependency)": q.SQ_WAIT_INST_ANY / r.SQ_WAVE_CYCLES,
                      "lds_issue_stall_frac_of_wave_time": q.SQ_WAIT_INST_LDS / r.SQ_WAVE_CYCLES})
        rows[k] = e
    json.dump({"source": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE (pass 1) and "
                         "--pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS (pass 2) "
                         "-- python3 tools/fcn_bench.py --dtype %s --batch %d --iters 1; averages per dispatch" % (DT, FB), "dtype": DT,
               "definitions": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs); clock = GRBM_GUI_ACTIVE/8 / duration",
               "tool_output": log_tail("mfma"), "kernels": rows}, open(os.path.join(OUT, TAG + "_conv_mfma_pmc.json"), "w"), indent=1)

# ---- 4. the voting kernels
v = counters("valu")
if v is not None:
    gv = v.groupby("k").mean(numeric_only=True)
    fz, wz = counters("vote_fetch"), counters("vote_write")
    n_runs = int((v.k == "k_setup").sum())
    rows = {}
    for k, r in gv.iterrows():
        if not k.startswith("k_"):
            continue
        cyc = r.GRBM_GUI_ACTIVE / 8.0
        n = int((v.k == k).sum())
        e = {"launches_per_batch": n / n_runs, "avg_us": r.dur_ns / 1e3, "ms_per_batch": r.dur_ns * n / n_runs / 1e6, "clock_ghz": cyc / r.dur_ns,
             "valu_instructions_per_launch": r.SQ_INSTS_VALU,
             "valu_issue_frac": 4 * r.SQ_INSTS_VALU / (cyc * N_SIMD)}
        if fz is not None and wz is not None:
            fb = float(fz[fz.k == k].FETCH_SIZE.sum()) if k in set(fz.k) else 0.0
            wb = float(wz[wz.k == k].WRITE_SIZE.sum()) if k in set(wz.k) else 0.0
            e["hbm_bytes_per_batch"] = (2 * fb + wb) * 1024 / n_runs
        rows[k] = e
    tot_ms = sum(e["ms_per_batch"] for e in rows.values())
    tot_bytes = sum(e.get("hbm_bytes_per_batch", 0.0) for e in rows.values())
    json.dump({"source": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE, and the "
                         "FETCH_SIZE / WRITE_SIZE passes -- python3 tools/vote_batch_bench.py --frames %d --iters 2 (seam + voting of %d (frame, keypoint) items)" % (FB, 3 * FB),
               "definitions": "valu_issue_frac = 4 * SQ_INSTS_VALU / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs): a wave64 VALU instruction holds its SIMD's 16 lanes for 4 cycles, "
                              "so this is lane-instructions / (256 CU x 4 SIMD x 16 lanes/clk x cycles)",
               "tool_output": log_tail("valu"), "frames_per_batch": FB, "kernel_ms_per_batch_sum": tot_ms, "hbm_bytes_per_batch_sum": tot_bytes,
               "dominant_kernel": max(rows, key=lambda k: rows[k]["ms_per_batch"]), "kernels": rows},
              open(os.path.join(OUT, TAG + "_vote_pmc.json"), "w"), indent=1)
print("wrote", [f for f in sorted(os.listdir(OUT)) if f.startswith(TAG)])
