#!/usr/bin/env python3
"""Condenses a tools/collect_profiles_producers.sh directory into pmc_producers.json: for every kernel of a default-mode level
(prep, front / binning, clip, tiles / resolve, search) per map: mean duration from the kernel trace, per-launch means of the SQ
counters and of the L2-fabric byte counters (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide loads on gfx950 --
an upper bound for narrow ones; WRITE_SIZE exact), and what they say: share of wave-cycles waiting, VALU and LDS instruction
rates, bytes moved against the HBM peak."""
import csv, glob, json, os, statistics as st, sys

out = sys.argv[1]
CUS, SIMDS, HBM = 256, 4, 8.0e12
res = {"units": {"time_us": "mean kernel duration (rocprofv3 --kernel-trace)", "sq_*": "per launch, summed over the chip (SQ counters tick per SIMD "
                 "in units of 4 cycles where the guide says so; ratios between them are what is used here)",
                 "fetch_bytes": "FETCH_SIZE x 1024 x 2 (gfx950 half-count correction for 16-byte-per-lane reads; an upper bound otherwise)",
                 "write_bytes": "WRITE_SIZE x 1024"}}
for m in ("cloud", "60x40", "300x200"):
    kernels = {}
    for f in glob.glob(os.path.join(out, f"trace_{m}", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("nmi::", "")
            kernels.setdefault(name, {"dur": []})["dur"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for tag in ("sq", "fetch", "write"):
        for f in glob.glob(os.path.join(out, f"pmc_{tag}_{m}", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("nmi::", "")
                kernels.setdefault(name, {"dur": []}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    table = {}
    for name, d in kernels.items():
        if not d["dur"] or not name.startswith("nmi_"):
            continue
        # steady state: drop the first fifth of the launches (warm-up, first-touch)
        def mean(v):
            v = v[len(v) // 5:] if len(v) >= 10 else v
            return st.mean(v)
        e = {"launches_traced": len(d["dur"]), "time_us": round(mean(d["dur"]), 2)}
        for k, v in d.items():
            if k != "dur":
                e[k] = mean(v)
        if "SQ_WAVE_CYCLES" in e and e["SQ_WAVE_CYCLES"] > 0:
            wc = e["SQ_WAVE_CYCLES"]
            e["derived"] = {
                "wave_cycles_waiting_frac": round(e.get("SQ_WAIT_ANY", 0.0) / wc, 3),
                "wave_cycles_issuing_valu_frac": round(e.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 3),
                "wave_cycles_stalled_on_lds_issue_frac": round(e.get("SQ_WAIT_INST_LDS", 0.0) / wc, 3),
                "valu_instructions_per_us": round(e.get("SQ_INSTS_VALU", 0.0) / e["time_us"], 1),
                "lds_instructions_per_us": round(e.get("SQ_INSTS_LDS", 0.0) / e["time_us"], 1),
                "lds_bank_conflict_cycles_per_lds_instruction": round(e.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(e.get("SQ_INSTS_LDS", 0.0), 1.0), 2),
                "mean_waves_resident_per_simd": round(wc / max(e.get("SQ_BUSY_CYCLES", 1.0), 1.0), 2),
            }
        if "FETCH_SIZE" in e or "WRITE_SIZE" in e:
            fb, wb = e.get("FETCH_SIZE", 0.0) * 1024 * 2, e.get("WRITE_SIZE", 0.0) * 1024
            e["fetch_bytes"], e["write_bytes"] = fb, wb
            e["fabric_bytes_per_s_frac_of_hbm_peak"] = round((fb + wb) / (e["time_us"] * 1e-6) / HBM, 3)
        table[name] = e
    res[m] = table
json.dump(res, open(os.path.join(out, "pmc_producers.json"), "w"), indent=1)
for m in ("cloud", "60x40", "300x200"):
    print(m)
    for name, e in sorted(res[m].items(), key=lambda kv: -kv[1]["time_us"]):
        d = e.get("derived", {})
        print(f"  {name:34s} {e['time_us']:7.1f} us  waiting {d.get('wave_cycles_waiting_frac', float('nan')):.2f}  valu-issue {d.get('wave_cycles_issuing_valu_frac', float('nan')):.2f}"
              f"  lds-stall {d.get('wave_cycles_stalled_on_lds_issue_frac', float('nan')):.2f}  fetch {e.get('fetch_bytes', 0) / 1e6:7.1f} MB  write {e.get('write_bytes', 0) / 1e6:7.1f} MB"
              f"  ({e.get('fabric_bytes_per_s_frac_of_hbm_peak', float('nan'))} of HBM peak)")
