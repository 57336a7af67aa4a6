#!/usr/bin/env python3
"""Turns one tools/profile.sh output directory into the committed summaries under profiles/:
  <tag>_kernel_stats.csv     rocprofv3's own --stats table, unchanged
  <tag>_by_label.md/.json    per bench label: launches, average duration, HBM bytes from the PMC
                             counters, VALU activity
  <tag>_pmc_summary.json     label -> hbm_bytes_per_launch (read by bench.py for roofline.traffic)

Dispatches are mapped to bench.py's labels by kernel name, grid size and order (see BY_GRID).
HBM bytes follow the gfx950 rule: FETCH_SIZE and WRITE_SIZE are in KiB, FETCH_SIZE counts half the
bytes of wide coalesced reads, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import OrderedDict, defaultdict

# Labels of one round-2+3 step.  The grand product and the Z transforms run on a side stream, so
# dispatch order interleaves between steps; launches are told apart by kernel name and grid size
# (side-stream transforms carry 1 column, main-stream ones 17; an extension has 3 cosets per column)
# and, for the two passes of one transform, by their order within that (name, grid) class.
BY_GRID = {
    "gp_local_z_kernel": [["gp_local_z_kernel"]],
    "gp_ratio_scan_kernel": [["gp_ratio_scan_kernel"]],
    "gp_apply_kernel": [["gp_apply_kernel"]],
    "gp_apply_fused_kernel": [["gp_apply_fused_kernel"]],
    # (a 2^20-point transform takes two passes: one ntt_pass launch and the final one; with three passes each
    # list here held p0 and p1)
    "ntt_pass_kernel": [["ntt_pass_kernel/intt20/side/p0"],
                        ["ntt_pass_kernel/lde20to22/side/p0"],
                        ["ntt_pass_kernel/intt20/p0"],
                        ["ntt_pass_kernel/lde20to22/p0"]],
    "ntt_final_kernel": [["ntt_final_kernel/intt20/side"], ["ntt_final_kernel/lde20to22/side"],
                         ["ntt_final_kernel/intt20"], ["ntt_final_kernel/lde20to22"]],
    "gate_constraint_coset_kernel": [["gate_constraint_coset_kernel"]],
    "perm_constraint_coset_kernel": [["perm_constraint_coset_kernel"]],
    "round23_constraints_coset_kernel": [["round23_constraints_coset_kernel"]],
}
STEP = [l for groups in BY_GRID.values() for g in groups for l in g]


def short(name):
    """'void plk::gp_apply_kernel<4>(unsigned long*, ...)' -> 'gp_apply_kernel'"""
    base = re.sub(r"plk(_field\d+)?::", "", name.split("(")[0])       # (the namespace carries the field's number)
    if base.startswith("void "):
        base = base[5:]
    return base.split("<")[0]


def grid_of(r):
    if "Grid_Size" in r:  # counter-collection rows carry the flattened size
        return int(r["Grid_Size"])
    return int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])


def labelled(rows, name_key):
    """yield (label, row) for the dispatches that belong to round-2+3 steps."""
    rows = [r for r in rows if re.search(r"plk(_field\d+)?::", r[name_key].split("(")[0]) and "table_kernel" not in r[name_key]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    grids = defaultdict(set)
    for r in rows:
        grids[short(r[name_key])].add(grid_of(r))
    seen = defaultdict(int)
    for r in rows:
        name = short(r[name_key])
        groups = BY_GRID[name]
        order = sorted(grids[name])
        assert len(order) == len(groups), (name, order)
        labels = groups[order.index(grid_of(r))]
        k = (name, grid_of(r))
        yield labels[seen[k] % len(labels)], r
        seen[k] += 1


def main():
    src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    os.makedirs(dst, exist_ok=True)
    stats = glob.glob(os.path.join(src, "trace", "**", "*_kernel_stats.csv"), recursive=True)[0]
    shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
    trace = glob.glob(os.path.join(src, "trace", "**", "*_kernel_trace.csv"), recursive=True)[0]
    agg = OrderedDict((l, defaultdict(float)) for l in STEP)
    info = {}
    for label, r in labelled(list(csv.DictReader(open(trace))), "Kernel_Name"):
        a = agg[label]
        a["launches"] += 1
        a["dur_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        info[label] = {"grid": f'{r["Grid_Size_X"]}x{r["Grid_Size_Y"]}', "wg": r["Workgroup_Size_X"], "vgpr": r["VGPR_Count"],
                       "sgpr": r["SGPR_Count"], "lds": r["LDS_Block_Size"], "scratch": r["Scratch_Size"]}
    counters = defaultdict(lambda: defaultdict(float))
    ncount = defaultdict(lambda: defaultdict(int))
    newest = {}   # one file per pass directory: merged output of repeated runs may hold older ones
    for path in glob.glob(os.path.join(src, "pmc_*", "**", "*_counter_collection.csv"), recursive=True):
        d = os.path.dirname(path)
        if d not in newest or os.path.getmtime(path) > os.path.getmtime(newest[d]):
            newest[d] = path
    for path in newest.values():
        rows = list(csv.DictReader(open(path)))
        names = sorted(set(r["Counter_Name"] for r in rows))
        for cname in names:
            sub = [r for r in rows if r["Counter_Name"] == cname]
            for label, r in labelled(sub, "Kernel_Name"):
                counters[label][cname] += float(r["Counter_Value"])
                ncount[label][cname] += 1
    out, pmc = [], {}
    for label in STEP:
        a = agg[label]
        if not a["launches"]:
            continue
        c = {k: v / ncount[label][k] for k, v in counters[label].items()}
        rec = {"label": label, "launches": int(a["launches"]), "avg_ms": a["dur_ns"] / a["launches"] / 1e6, **info[label]}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            rec["hbm_read_bytes"] = 2 * c["FETCH_SIZE"] * 1024
            rec["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024
            rec["hbm_bytes_per_launch"] = rec["hbm_read_bytes"] + rec["hbm_write_bytes"]
            rec["hbm_GBps"] = rec["hbm_bytes_per_launch"] / (rec["avg_ms"] * 1e-3) / 1e9
            pmc[label] = {"hbm_bytes_per_launch": rec["hbm_bytes_per_launch"], "fetch_kib_raw": c["FETCH_SIZE"],
                          "write_kib_raw": c["WRITE_SIZE"]}
        if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c:
            rec["valu_insts_per_launch"] = c["SQ_INSTS_VALU"]          # wave-level instructions, all SIMDs
            rec["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / max(c["SQ_WAVES"], 1)
            rec["waves"] = c["SQ_WAVES"]
            # SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles summed over waves: ratios only
            rec["wait_any_over_wave_cycles"] = c.get("SQ_WAIT_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 0), 1)
            if "GRBM_GUI_ACTIVE" in c:
                # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs issue the launch's VALU instructions
                rec["cycles_per_valu_inst_per_simd"] = (c["GRBM_GUI_ACTIVE"] / 8) * 1024 / max(c["SQ_INSTS_VALU"], 1)
            if label in pmc:
                pmc[label]["valu_insts_per_launch"] = rec["valu_insts_per_launch"]
                pmc[label]["cycles_per_valu_inst_per_simd"] = rec.get("cycles_per_valu_inst_per_simd")
        for k in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "GRBM_GUI_ACTIVE", "SQ_ACTIVE_INST_LDS"):
            if k in c:
                rec[k] = c[k]
        if "GRBM_GUI_ACTIVE" in c:
            rec["eff_clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8 / (rec["avg_ms"] * 1e6)
        rec["raw_counters"] = c
        out.append(rec)
    json.dump(out, open(os.path.join(dst, f"{tag}_by_label.json"), "w"), indent=1)
    json.dump(pmc, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1)
    with open(os.path.join(dst, f"{tag}_by_label.md"), "w") as f:
        f.write(f"# rocprofv3 summary `{tag}` — `python3 bench.py --steps 3 --warmup 1` (n = 2^20, N = 2^22, 18 columns)\n\n")
        f.write("| label | launches | avg ms | grid | VGPR | LDS B | HBM GB/launch (PMC) | HBM GB/s | VALU wave-instr / launch | cycles per VALU instr per SIMD | wait-any / wave-cycles | eff. clock GHz |\n|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for r in out:
            f.write(f'| {r["label"]} | {r["launches"]} | {r["avg_ms"]:.3f} | {r["grid"]} | {r["vgpr"]} | {r["lds"]} | '
                    f'{r.get("hbm_bytes_per_launch", 0) / 1e9:.3f} | {r.get("hbm_GBps", 0):.0f} | '
                    f'{r.get("valu_insts_per_launch", 0):.3g} | {r.get("cycles_per_valu_inst_per_simd", 0):.2f} | '
                    f'{r.get("wait_any_over_wave_cycles", 0):.3f} | {r.get("eff_clock_GHz", 0):.2f} |\n')
        f.write("\nHBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE counts half of a wide coalesced read).\n"
                "Cycles per VALU instruction per SIMD = (GRBM_GUI_ACTIVE / 8 XCDs) x 1024 SIMDs / SQ_INSTS_VALU: the launch's duration in\n"
                "shader cycles divided by the wave-level vector instructions each SIMD issued.  The transform kernels' instruction mix\n"
                "(about two thirds v_mad_u64_u32 at ~5 cycles, the rest at ~2.6; tools/ubench_valu.hip) costs ~4.2 cycles per instruction,\n"
                "so a value near 4.3 means the vector ALUs are issuing back to back: those kernels are bound by integer multiply-add issue,\n"
                "not by HBM.  Side-stream launches (`/side`, grand product) overlap main-stream kernels, so their own figures are diluted.\n")
    print(open(os.path.join(dst, f"{tag}_by_label.md")).read())


if __name__ == "__main__":
    main()
