#!/usr/bin/env python3
"""Condenses a tools/collect_profiles.sh output directory into the files that get committed under profiles/:
kernel_stats.csv (rocprofv3 --stats), pmc_summary.json (per-launch means of the nmi_grid_kernel counters) and
pmc_traffic.json (HBM-side bytes per launch with the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md)."""
import csv, glob, json, os, shutil, statistics as st, sys

out = sys.argv[1]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # kernel_source_id(): the stamp that ties a counter profile to the kernel sources it was taken on
SRC_ID = bench.kernel_source_id()
summary = {}
for name in ("pmc_fetch", "pmc_write", "pmc_lds"):
    files = glob.glob(os.path.join(out, name, "*", "*_counter_collection.csv"))
    if not files:
        continue
    rows = [r for r in csv.DictReader(open(files[0])) if "nmi_grid_kernel" in r["Kernel_Name"]]
    by = {}
    for r in rows:
        by.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in by.items():
        summary[k] = {"mean": st.mean(v), "min": min(v), "max": max(v), "launches": len(v)}
stats = glob.glob(os.path.join(out, "trace", "*", "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(out, "kernel_stats.csv"))
    for r in csv.DictReader(open(stats[0])):
        if "nmi_grid_kernel" in r["Name"]:
            summary["kernel_trace"] = {"name": r["Name"], "calls": int(r["Calls"]), "average_ns": float(r["AverageNs"]),
                                       "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])}
json.dump(summary, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  On gfx950 FETCH_SIZE counts exactly half of the bytes of a wide
    # (16 B/lane) coalesced read stream -- this kernel's loads -- so it is doubled; WRITE_SIZE is exact (MI355X_MICROARCH.md, HBM).
    fetch = summary["FETCH_SIZE"]["mean"] * 1024 * 2
    write = summary["WRITE_SIZE"]["mean"] * 1024
    json.dump({"hbm_bytes_per_launch": fetch + write, "fetch_bytes_corrected_x2": fetch, "write_bytes": write,
               "note": "L2-fabric side counters (TCC_EA0): Infinity-Cache hits are included, so this is an upper bound on HBM bytes",
               "source": os.path.basename(out), "kernel_source_sha16": SRC_ID}, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
if "SQ_LDS_IDX_ACTIVE" in summary and "SQ_INSTS_LDS" in summary:
    # LDS side of the dominant kernel (what actually bounds it).  SQ counters tick in units of 4 cycles
    # summed over the CUs' SIMDs.  Conflict-free reference: a 64-lane ds_add_u32 is served as 2 groups of 32 lanes, one
    # LDS-array cycle each when no two lanes of a group meet in a bank (MI355X_MICROARCH.md section LDS, ds_write_b32 row).
    insts = summary["SQ_INSTS_LDS"]["mean"]
    active = summary["SQ_LDS_IDX_ACTIVE"]["mean"]
    conflict = summary["SQ_LDS_BANK_CONFLICT"]["mean"]
    json.dump({"lds_wave_instructions_per_launch": insts, "lds_idx_active_cycles_per_launch": active,
               "lds_bank_conflict_cycles_per_launch": conflict, "conflict_ratio": conflict / active if active else None,
               "lds_active_cycles_per_wave_instruction": active / insts if insts else None,
               "conflict_free_cycles_per_wave_instruction": 2.0,
               "note": "SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT / SQ_INSTS_LDS per launch of nmi_grid_kernel (rocprofv3 --pmc, own pass); "
                       "conflict-free rate from MI355X_MICROARCH.md section LDS (2 x 32 lanes, one LDS-array cycle per group)",
               "source": os.path.basename(out), "kernel_source_sha16": SRC_ID}, open(os.path.join(out, "pmc_lds.json"), "w"), indent=1)
for cfg in ("c3", "c4"):
    stats = glob.glob(os.path.join(out, "trace_" + cfg, "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(out, f"kernel_stats_{cfg}.csv"))
print(json.dumps(summary, indent=1)[:1500])
