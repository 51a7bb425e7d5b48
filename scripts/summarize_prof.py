"""Summarise rocprofv3 output (kernel-trace stats + PMC passes) into profiles/<tag>_summary.{json,md}.

usage: python scripts/summarize_prof.py <tag> <kernel_trace_dir> <pmc_dir> [bench.json]
Corrections applied exactly as MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE (KiB) is doubled for wide
coalesced reads on gfx950 (it tallies 128-B requests at 64 B); WRITE_SIZE (KiB) is taken as is.

Derived columns and the counters they come from (rocprofv3 sums every counter over the chip; SQ_* "cycles" are quad-cycles,
GRBM_GUI_ACTIVE is the sum of the 8 XCDs' busy cycles; gfx950 has no derived-metric section, these are the gfx94x forms):
  cycles            = GRBM_GUI_ACTIVE / 8                         shader cycles of the launch
  VALU busy %       = min(100, raw) with raw = 100 * 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * cycles).  SQ_ACTIVE_INST_VALU
                      charges every wave 4 cycles per VALU instruction, while fma/mul/add/and/mov-class instructions of
                      DIFFERENT waves issue 2.2-3 cycles apart on a SIMD (rt_oprate, profiles/r01g_peaks.md), so the raw
                      sum can pass 100: raw >= 100 means the SIMDs' VALU issue ports were never idle.  raw is kept beside it.
  VALU inst / SIMD-cycle = SQ_INSTS_VALU / (1024 * cycles)        ceiling 0.25 for the 4-cycle classes (min/max, cvt, cmp,
                      cndmask, shifts, packed fma), ~0.4 for the 2.4-cycle classes
  lane util %       = 100 * SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)
  waves/CU          = 4 * SQ_WAVE_CYCLES / cycles / 256
VGPRs, scratch and the compiler's occupancy bound come from `make -C raytracer_amd/csrc resources`
(profiles/<tag>_kernel_resources.txt), not from rocprofv3: its VGPR_Count column reads 40 for a kernel that allocates 80,
and its LDS_Block_Size column shows only static LDS (0 here: the traversal stacks are dynamic LDS, a launch argument).
Dynamic LDS per 256-thread block is taken from the bench line's `launch` object when a bench.json is given.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

tag, kdir, pdir = sys.argv[1], sys.argv[2], sys.argv[3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
launch = {}
if len(sys.argv) > 4 and os.path.exists(sys.argv[4]):
    try:
        launch = json.loads([l for l in open(sys.argv[4]).read().splitlines() if l.startswith("{")][-1]).get("launch", {})
    except Exception:
        launch = {}


def dyn_lds(kernel):
    for key, val in launch.get("dynamic_lds_bytes_per_block", {}).items():
        if kernel.startswith(key):
            return val
    return None



def short(name):
    for k in ("k_trace", "k_primary", "k_bounce", "k_shadow", "k_shade", "k_resolve", "k_bin"):
        if k in name:
            t = name[name.index(k):]
            return t[: t.index("(")] if "(" in t else t
    return name[:40]


kern = {}
for f in glob.glob(os.path.join(kdir, "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        kern[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                  "total_ms": float(r["TotalDurationNs"]) / 1e6, "pct": float(r["Percentage"])}

pmc = defaultdict(lambda: defaultdict(list))      # kernel -> counter -> per-dispatch values
meta = {}
for f in glob.glob(os.path.join(pdir, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        pmc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = {"vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"]), "lds": int(r["LDS_Block_Size"]),
                   "grid": int(r["Grid_Size"]), "wg": int(r["Workgroup_Size"])}

out = {"tag": tag, "kernels": {}}
for k, ctrs in pmc.items():
    if k.startswith("__amd") or "Cijk" in k or "at::" in k or "elementwise" in k:
        continue
    m = {c: sum(v) / len(v) for c, v in ctrs.items()}     # mean per launch
    d = dict(meta.get(k, {}))
    d.update(kern.get(k, {}))
    d["counters_per_launch"] = {c: round(v, 2) for c, v in sorted(m.items())}
    der = {}
    if "FETCH_SIZE" in m:
        der["hbm_read_bytes"] = m["FETCH_SIZE"] * 1024 * 2          # gfx950 correction (x2)
    if "WRITE_SIZE" in m:
        der["hbm_write_bytes"] = m["WRITE_SIZE"] * 1024
    if "hbm_read_bytes" in der and "hbm_write_bytes" in der and "avg_us" in d:
        der["hbm_bytes"] = der["hbm_read_bytes"] + der["hbm_write_bytes"]
        der["hbm_GBps"] = der["hbm_bytes"] / (d["avg_us"] * 1e-6) / 1e9
        der["hbm_pct_of_8TBps"] = 100 * der["hbm_GBps"] / 8000
    if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m and m["TCC_HIT_sum"] + m["TCC_MISS_sum"] > 0:
        der["l2_hit_pct"] = 100 * m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    if "SQ_ACTIVE_INST_VALU" in m and m.get("GRBM_GUI_ACTIVE", 0) > 0:
        cycles = m["GRBM_GUI_ACTIVE"] / 8
        raw = 100 * 4 * m["SQ_ACTIVE_INST_VALU"] / (1024 * cycles)
        der["valu_busy_raw_pct"] = raw
        der["valu_busy_pct"] = min(100.0, raw)
        if "SQ_INSTS_VALU" in m:
            der["valu_inst_per_simd_cycle"] = m["SQ_INSTS_VALU"] / (1024 * cycles)
        if "avg_us" in d:
            der["effective_clock_GHz"] = cycles / (d["avg_us"] * 1e3)
    if "SQ_THREAD_CYCLES_VALU" in m and "SQ_ACTIVE_INST_VALU" in m and m["SQ_ACTIVE_INST_VALU"] > 0:
        der["valu_lane_utilisation_pct"] = 100 * m["SQ_THREAD_CYCLES_VALU"] / (m["SQ_ACTIVE_INST_VALU"] * 64)
    if "SQ_LDS_BANK_CONFLICT" in m and "SQ_LDS_IDX_ACTIVE" in m and m["SQ_LDS_IDX_ACTIVE"] > 0:
        der["lds_bank_conflict_pct_of_lds_cycles"] = 100 * m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]
    if "SQ_WAVE_CYCLES" in m and "GRBM_GUI_ACTIVE" in m and m["GRBM_GUI_ACTIVE"] > 0:
        # SQ_WAVE_CYCLES counts quad-cycles summed over waves; GRBM_GUI_ACTIVE is summed over the 8 XCDs
        der["mean_waves_per_cu"] = m["SQ_WAVE_CYCLES"] * 4 / (m["GRBM_GUI_ACTIVE"] / 8) / 256
        der["occupancy_pct_of_32_waves"] = 100 * der["mean_waves_per_cu"] / 32
    if "SQ_WAIT_ANY" in m and "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"] > 0:
        der["wave_wait_any_pct"] = 100 * m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]
        if "SQ_ACTIVE_INST_ANY" in m:
            der["wave_issuing_pct"] = 100 * m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"]
    if "SQ_INSTS_VALU" in m and "SQ_WAVES" in m and m["SQ_WAVES"] > 0:
        der["valu_insts_per_wave"] = m["SQ_INSTS_VALU"] / m["SQ_WAVES"]
    d["dynamic_lds_bytes_per_block"] = dyn_lds(k)
    d["derived"] = {a: (round(b, 3) if isinstance(b, float) else b) for a, b in der.items()}
    out["kernels"][k] = d
for k, v in kern.items():
    if k not in out["kernels"] and not k.startswith("__amd"):
        out["kernels"][k] = v

os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_summary.json"), "w"), indent=1)
with open(os.path.join(ROOT, "profiles", f"{tag}_summary.md"), "w") as f:
    f.write(f"# rocprofv3 summary `{tag}`\n\nkernel-trace: `{kdir}`; PMC passes: `{pdir}` (one pass per run, scripts/pmc_passes.sh)\n\n")
    f.write("Formulas and counters: header of `scripts/summarize_prof.py`.  VGPRs / scratch: `profiles/%s_kernel_resources.txt`.\n\n" % tag)
    f.write("| kernel | calls | avg us | dyn. LDS B/block | HBM GB/s (% of 8 TB/s) | L2 hit % | VALU busy % (raw) | VALU inst/SIMD-cycle | lane util % | waves/CU (occ %) | LDS conflict % | wait-any % |\n|---|---|---|---|---|---|---|---|---|---|---|---|\n")
    for k, d in out["kernels"].items():
        e = d.get("derived", {})
        g = lambda x, fmt="{:.1f}": fmt.format(e[x]) if e.get(x) is not None else "-"
        lds = d.get("dynamic_lds_bytes_per_block")
        f.write(f"| `{k}` | {d.get('calls','-')} | {d.get('avg_us',0):.1f} | {lds if lds is not None else '-'} | {g('hbm_GBps')} ({g('hbm_pct_of_8TBps')}) | {g('l2_hit_pct')} | {g('valu_busy_pct')} ({g('valu_busy_raw_pct')}) | {g('valu_inst_per_simd_cycle','{:.3f}')} | {g('valu_lane_utilisation_pct')} | {g('mean_waves_per_cu')} ({g('occupancy_pct_of_32_waves')}) | {g('lds_bank_conflict_pct_of_lds_cycles','{:.2f}')} | {g('wave_wait_any_pct')} |\n")
# what bench.py reads for roofline.traffic
cands = [d for k, d in out["kernels"].items() if k.startswith("k_trace") and "hbm_bytes" in d.get("derived", {})]
kt = max(cands, key=lambda d: d.get("total_ms", 0.0)) if cands else None     # the bounce closest-hit launch
if kt and "hbm_bytes" in kt.get("derived", {}):
    import hashlib
    ksha = hashlib.sha256(b"".join(open(os.path.join(ROOT, "raytracer_amd", "csrc", f), "rb").read() for f in ("kernels.hip", "kernels.hpp", "Makefile"))).hexdigest()
    json.dump({"tag": tag, "kernel_sources_sha256": ksha, "k_trace_hbm_bytes_per_launch": kt["derived"]["hbm_bytes"],
               "k_trace_valu_busy_pct": kt["derived"].get("valu_busy_pct"),
               "k_trace_valu_inst_per_simd_cycle": kt["derived"].get("valu_inst_per_simd_cycle"),
               "k_trace_lane_utilisation_pct": kt["derived"].get("valu_lane_utilisation_pct"),
               "k_trace_l2_hit_pct": kt["derived"].get("l2_hit_pct")},
              open(os.path.join(ROOT, "profiles", "pmc_latest.json"), "w"))
print(open(os.path.join(ROOT, "profiles", f"{tag}_summary.md")).read())
