"""Post-process the rocprofv3 --pmc passes of tools/r06/counters.sh: per-launch averages of the largest-grid RoIAlign kernel + derived fractions."""
import csv, json, collections, glob, sys
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for f in sorted(glob.glob(O + "/g*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "roi_align" in r["Kernel_Name"]:
            g = int(r["Grid_Size"])
            acc[g][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[g][(f, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
g = max(acc, key=lambda k: (len(next(iter(acc[k].values()))), k))      # the launch that was repeated (ties: the larger grid)
avg = {k: sum(v[1:]) / max(1, len(v) - 1) for k, v in acc[g].items()}
avg["grid"] = g
avg["kernel"] = [r["Kernel_Name"][:60] for r in csv.DictReader(open(sorted(glob.glob(O + "/g1_counter_collection.csv"))[0])) if "roi_align" in r["Kernel_Name"] and int(r["Grid_Size"]) == g][0]
d = list(dur[g].values())
avg["launch_ms_under_counters"] = sum(d) / len(d)
if "TCP_TCC_READ_REQ_sum" in avg:
    avg["l1_fill_latency_cycles"] = avg["TCP_TCC_READ_REQ_LATENCY_sum"] / max(1.0, avg["TCP_TCC_READ_REQ_sum"])
    avg["l1_fill_GB (x128 B)"] = avg["TCP_TCC_READ_REQ_sum"] * 128 / 1e9
if "TCC_HIT_sum" in avg:
    avg["l2_read_hit_fraction"] = avg["TCC_HIT_sum"] / max(1.0, avg["TCC_HIT_sum"] + avg["TCC_MISS_sum"])
if "GRBM_GUI_ACTIVE" in avg and "SQ_WAVE_CYCLES" in avg:
    cyc = avg["GRBM_GUI_ACTIVE"] / 8.0          # summed over the 8 XCDs
    simd_q = cyc * 1024 / 4.0                   # SIMD quad-cycles of the launch
    avg["derived"] = {"launch_cycles": cyc, "waves_per_simd": avg["SQ_WAVE_CYCLES"] / simd_q,
                      "wave_issuing_frac": avg["SQ_ACTIVE_INST_ANY"] / avg["SQ_WAVE_CYCLES"], "wave_parked_frac": avg["SQ_WAIT_ANY"] / avg["SQ_WAVE_CYCLES"],
                      "wave_issue_stall_frac": avg["SQ_WAIT_INST_ANY"] / avg["SQ_WAVE_CYCLES"], "valu_busy_frac_of_simd_time": avg["SQ_ACTIVE_INST_VALU"] / simd_q,
                      "lds_array_busy_frac (SQ_LDS_IDX_ACTIVE, conflicts included)": avg["SQ_LDS_IDX_ACTIVE"] / 256.0 / cyc,
                      "lds_conflict_frac_of_lds_active": avg["SQ_LDS_BANK_CONFLICT"] / avg["SQ_LDS_IDX_ACTIVE"]}
if "TD_TD_BUSY_sum" in avg and "GRBM_GUI_ACTIVE" in avg:
    cyc = avg["GRBM_GUI_ACTIVE"] / 8.0
    avg.setdefault("derived", {})
    avg["derived"]["td_busy_frac (TD_TD_BUSY / 256 CUs / cycles)"] = avg["TD_TD_BUSY_sum"] / 256.0 / cyc
    avg["derived"]["ta_busy_frac (TA_TA_BUSY / 256 CUs / cycles)"] = avg["TA_TA_BUSY_sum"] / 256.0 / cyc
json.dump(avg, open(O + "/counters.json", "w"), indent=1)
for k, v in sorted(avg.items()):
    if isinstance(v, dict):
        for kk, vv in v.items(): print("   ", kk, round(vv, 4))
    else: print(k, v if not isinstance(v, float) else round(v, 4))
