"""gpurun_out/r05_* (tools/collect_profiles_r05.sh) -> profiles/r05_*: rocprofv3 kernel tables, the per-env-step counter sums
of the resident family's launches (the small class's launch + the full-size class's: all instantiations of kr_step together),
the traffic and VALU records bench.py quotes, the 16x16 and Jinan records, the hour(s)."""
import collections
import csv
import glob
import json
import os
import shutil

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(REPO, "gpurun_out")
P = os.path.join(REPO, "profiles")
CLOCK_GHZ = 2.4
N_SIMD = 1024


def bench_line(log):
    if not os.path.exists(log):
        return None
    for line in open(log):
        if line.startswith("{"):
            return json.loads(line)
    return None


def kernel_stats(tag):
    f = glob.glob(os.path.join(G, f"r05_stats_{tag}", "**", "*kernel_stats.csv"), recursive=True)
    if f:
        shutil.copy(f[0], os.path.join(P, f"r05_{tag}_kernel_stats.csv"))
    bl = bench_line(os.path.join(G, f"r05_stats_{tag}.log"))
    if bl:
        with open(os.path.join(P, f"r05_{tag}_bench_line_under_rocprof.json"), "w") as o:
            json.dump(bl, o)
    return (list(csv.DictReader(open(f[0]))) if f else []), bl


def short(name):
    return name.replace("void ", "").split("<")[0].split("(")[0]


def per_step(tag, steps, prefixes=None):
    """{counter: mean per ENV STEP over the last `steps` env steps, summed over the kernels of a step}, and the same by
    kernel.  The number of launches of a kernel per env step = its launches / those of the kernel launched once per step."""
    f = glob.glob(os.path.join(G, f"r05_pmc_{tag}", "**", "*counter_collection.csv"), recursive=True)
    if not f:
        return {}, {}
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f[0])):
        name = short(row["Kernel_Name"])
        if name.startswith("at::") or name.startswith("__amd") or "elementwise" in name:
            continue
        if prefixes and not any(name.startswith(p) for p in prefixes):
            continue
        # (kr_step: the small and the full-size launch of a step are instantiations of one template: kept apart by their
        #  thread count, each is launched once per env step)
        if name == "kr_step":
            name = "kr_step<512>" if ", 512, " in row["Kernel_Name"] else ("kr_step<256 pool>" if ", 256, " in row["Kernel_Name"] else "kr_step<768>")
        per[row["Counter_Name"]][name].append(float(row["Counter_Value"]))
    tot, by = {}, {}
    for c, d in per.items():
        once = d.get("k_stats") or d.get("kr_step<768>") or d.get("kr_step<512>") or next(iter(d.values()))
        n_once = max(1, len(once))
        t, b = 0.0, {}
        for k, v in d.items():
            n_per_step = max(1, round(len(v) / n_once))
            take = v[-steps * n_per_step:]
            b[k] = sum(take) / steps
            t += b[k]
        tot[c], by[c] = t, b
    return tot, by


def main():
    os.makedirs(P, exist_ok=True)
    for tag in ("resident", "resident240", "c16", "c16_sets2", "jinan"):
        kernel_stats(tag)
    # ---- resident family, driver command: counters per env step (small launch + full-size launch)
    allc = {}
    for tag in ("fetch", "write", "sq1", "sq2", "tcc"):
        t, _ = per_step(tag, 20, ("kr_step",))
        allc.update(t)
    bl = bench_line(os.path.join(G, "r05_stats_resident.log"))
    if allc:
        with open(os.path.join(P, "r05_resident_pmc_summary.txt"), "w") as o:
            o.write("kr_step (small class's launch + full-size class's: every launch of an env step summed), mean per ENV STEP over the "
                    "last 20 env steps of `bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5` (after the 150-step pre-roll), "
                    "separate rocprofv3 --pmc passes, kernel-trace only\n")
            for c in sorted(allc):
                o.write(f"{c:28s} {allc[c]:18.1f}\n")
            if "SQ_WAVE_CYCLES" in allc:
                w = allc["SQ_WAVE_CYCLES"]
                o.write("\nshares of resident wave-cycles: waiting (s_waitcnt / barrier) %.1f %%, issue stalls %.1f %%, issuing %.1f %%\n" % (
                    100 * allc.get("SQ_WAIT_ANY", 0) / w, 100 * allc.get("SQ_WAIT_INST_ANY", 0) / w, 100 * allc.get("SQ_ACTIVE_INST_ANY", 0) / w))
            if "TCC_HIT_sum" in allc:
                o.write("L2 hit rate %.1f %%\n" % (100 * allc["TCC_HIT_sum"] / (allc["TCC_HIT_sum"] + allc["TCC_MISS_sum"])))
    bl_f = bench_line(os.path.join(G, "r05_pmc_fetch.log"))
    if bl_f and "FETCH_SIZE" in allc and "WRITE_SIZE" in allc:
        R, A = bl_f["config"]["replicas_per_gpu"], bl_f["config"]["agents"]
        rec = {"kernel": "kr_step", "replicas": R, "V_a": bl_f["config"]["V_a"], "command": "bench.py --steps 20 --warmup 5",
               "fetch_size_kb_per_launch": allc["FETCH_SIZE"], "write_size_kb_per_launch": allc["WRITE_SIZE"],
               "bytes_per_launch": (2.0 * allc["FETCH_SIZE"] + allc["WRITE_SIZE"]) * 1024.0,
               "algorithmic_bytes_per_launch": bl_f["roofline"]["bytes_per_agent_env_step"] * R * A,
               "formula": "2 x FETCH_SIZE + WRITE_SIZE (kB x 1024) summed over the launches of an env step (small class, full-size class), "
                          "mean of the last 20 env steps",
               "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate kernel-trace-only passes (tools/collect_profiles_r05.sh); "
                      "FETCH_SIZE doubled (gfx950 tallies 128-byte requests at 64 bytes: MI355X_MICROARCH.md, HBM)"}
        rec["ratio_to_algorithmic"] = rec["bytes_per_launch"] / rec["algorithmic_bytes_per_launch"]
        with open(os.path.join(P, "r05_traffic_resident.json"), "w") as o:
            json.dump([rec], o, indent=1)
        print("resident traffic ratio", rec["ratio_to_algorithmic"])
    bl_sq = bench_line(os.path.join(G, "r05_pmc_sq1.log"))
    if bl_sq and "SQ_INSTS_VALU" in allc and "SQ_INSTS_SALU" in allc:
        dur_us = bl_sq["roofline"]["avg_launch_us"]
        rec = {"kernel": "kr_step", "replicas": bl_sq["config"]["replicas_per_gpu"], "V_a": bl_sq["config"]["V_a"],
               "avg_launch_us_under_pmc": dur_us,
               "valu_busy_frac": allc["SQ_INSTS_VALU"] * 4.0 / (N_SIMD * dur_us * 1e3 * CLOCK_GHZ),
               "active_lane_frac": allc["SQ_THREAD_CYCLES_VALU"] / (64.0 * allc["SQ_ACTIVE_INST_VALU"]),
               "salu_per_valu": allc["SQ_INSTS_SALU"] / allc["SQ_INSTS_VALU"],
               "wait_frac": allc["SQ_WAIT_ANY"] / allc["SQ_WAVE_CYCLES"],
               "lds_bank_conflict_ratio": allc["SQ_LDS_BANK_CONFLICT"] / allc["SQ_ACTIVE_INST_LDS"],
               "waves_per_simd": 4,
               "how": "rocprofv3 --pmc SQ_* passes (kernel-trace only) of `bench.py --steps 20 --warmup 5`, per env step (all launches of "
                      "the step summed), mean of the last 20; valu_busy = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x step duration x "
                      f"{CLOCK_GHZ} GHz); waves per SIMD: two 512-thread workgroups per CU"}
        rec["useful_valu_frac"] = rec["valu_busy_frac"] * rec["active_lane_frac"]
        if bl:
            u_us = bl["roofline"]["avg_launch_us"]
            rec["avg_launch_us_undisturbed"] = u_us
            rec["valu_busy_frac_undisturbed"] = allc["SQ_INSTS_VALU"] * 4.0 / (N_SIMD * u_us * 1e3 * CLOCK_GHZ)
            rec["useful_valu_frac_undisturbed"] = rec["valu_busy_frac_undisturbed"] * rec["active_lane_frac"]
        rec["valu_insts_per_launch"] = allc["SQ_INSTS_VALU"]
        with open(os.path.join(P, "r05_valu_resident.json"), "w") as o:
            json.dump([rec], o, indent=1)
        print(json.dumps(rec))
    # ---- Jinan, 16x16
    tj, _ = per_step("jinan_fetch", 40, ("kr_step",))
    tw, _ = per_step("jinan_write", 40, ("kr_step",))
    blj = bench_line(os.path.join(G, "r05_pmc_jinan_fetch.log"))
    if blj and "FETCH_SIZE" in tj and "WRITE_SIZE" in tw:
        R, A = blj["config"]["replicas_per_gpu"], blj["config"]["agents"]
        rec = {"kernel": "kr_step", "replicas": R, "V_a": blj["config"]["V_a"], "bytes_per_launch": (2.0 * tj["FETCH_SIZE"] + tw["WRITE_SIZE"]) * 1024.0,
               "algorithmic_bytes_per_launch": blj["roofline"]["bytes_per_agent_env_step"] * R * A, "command": "bench.py --scenario jinan --replicas 8192 --steps 40 --warmup 5",
               "formula": "2 x FETCH_SIZE + WRITE_SIZE per env step, last 40"}
        rec["ratio_to_algorithmic"] = rec["bytes_per_launch"] / rec["algorithmic_bytes_per_launch"]
        with open(os.path.join(P, "r05_traffic_jinan.json"), "w") as o:
            json.dump([rec], o, indent=1)
    bl16 = bench_line(os.path.join(G, "r05_pmc_c16_fetch.log"))
    f16, by_f = per_step("c16_fetch", 10)
    w16, by_w = per_step("c16_write", 10)
    if bl16 and "FETCH_SIZE" in f16 and "WRITE_SIZE" in w16:
        R, A = bl16["config"]["replicas_per_gpu"], bl16["config"]["agents"]
        alg = bl16["roofline"]["bytes_per_agent_env_step"] * R * A
        b = (2 * f16["FETCH_SIZE"] + w16["WRITE_SIZE"]) * 1024.0
        rec = {"workload": bl16["config"]["workload"], "V_a": bl16["config"]["V_a"], "value_under_the_counter_pass": bl16["value"],
               "ms_per_step_under_the_counter_pass": bl16["ms_per_step"], "bytes_per_env_step": b, "algorithmic_bytes_per_env_step": alg,
               "ratio_to_algorithmic": b / alg, "fetch_kb_by_kernel": by_f.get("FETCH_SIZE"), "write_kb_by_kernel": by_w.get("WRITE_SIZE"),
               "how": "sum over every kernel of an env step (list family), last 10 env steps; FETCH_SIZE doubled; 64 flow sets: every "
                      "lane of a wavefront reads its own set's spawn queue / path rows"}
        with open(os.path.join(P, "r05_traffic_c16.json"), "w") as o:
            json.dump(rec, o, indent=1)
        print("16x16 traffic ratio", rec["ratio_to_algorithmic"])
    # ---- hours
    for name in ["r05_hour.json"] + [f"r05_hour_rate{r}.json" for r in (700, 900, 1100, 1300, 1500)]:
        src = os.path.join(G, name)
        if os.path.exists(src) and bench_line(src):
            shutil.copy(src, os.path.join(P, name))
    curve = []
    for r in (700, 900, 1100, 1300, 1500):
        d = bench_line(os.path.join(G, f"r05_hour_rate{r}.json"))
        if d:
            curve.append({"rate_veh_per_h_per_entry_road": r, "value": d["value"], "windows_M": [round(w["value"] / 1e6, 1) for w in d["windows"]],
                          "left_resident_family_at_env_step": d["left_resident_family_at_env_step"],
                          "share_of_replica_steps_off_resident": d["share_of_replica_steps_off_resident"],
                          "promotions": d["spill_pool"]["slots"], "first_promotion_at_env_step": d["spill_pool"]["first_promotion_at_env_step"],
                          "small_class_share_at_window_start": d.get("share_of_replicas_in_the_small_class_at_window_start"),
                          "mean_live_vehicles_per_replica_at_the_end": d["mean_live_vehicles_per_replica_at_the_end"],
                          "stalled_replicas": d["stalled_replicas"], "device_errors": d["device_errors"]})
            before = bench_line(os.path.join(P, f"r05_hour_rate{r}_log_replay.json"))  # (kept from before kr_to_list)
            if before:
                curve[-1]["with_log_replay_before_kr_to_list"] = {"value": before["value"], "windows_M": [round(w["value"] / 1e6, 1) for w in before["windows"]]}
    if curve:
        with open(os.path.join(P, "r05_demand_curve.json"), "w") as o:
            json.dump({"workload": "4x4 @ 4096, 256 flow sets, one simulated hour from empty + restart, every replica at the same rate", "curve": curve,
                       "note": "a handle that leaves the resident family writes its replicas' state across (kr_to_list) unless a replica had stalled; "
                               "with_log_replay_before_kr_to_list: the same hour while the hand-over replayed the log"}, o, indent=1)
        print(json.dumps([(c["rate_veh_per_h_per_entry_road"], round(c["value"] / 1e6, 1), c["left_resident_family_at_env_step"]) for c in curve]))


if __name__ == "__main__":
    main()
