"""gpurun_out/r04_* (tools/collect_profiles_r04.sh) -> profiles/r04_*: kernel stats (headline, list family, 16x16, Jinan),
the counter summary of kr_step, the traffic records bench.py quotes (`roofline.traffic`), the VALU record
(`roofline_valu`) and the flow-set sweep."""
import collections
import csv
import glob
import json
import os
import shutil

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(REPO, "gpurun_out")
P = os.path.join(REPO, "profiles")
CLOCK_GHZ = 2.4  # MI355X peak engine clock (MI355X_MICROARCH.md); kernel cycles below = duration x this
N_SIMD = 1024


def bench_line(log):
    if not os.path.exists(log):
        return None
    for line in open(log):
        if line.startswith("{"):
            return json.loads(line)
    return None


def kernel_stats(tag):
    f = glob.glob(os.path.join(G, f"r04_stats_{tag}", "**", "*kernel_stats.csv"), recursive=True)
    if f:
        shutil.copy(f[0], os.path.join(P, f"r04_{tag}_kernel_stats.csv"))
    bl = bench_line(os.path.join(G, f"r04_stats_{tag}.log"))
    if bl:
        with open(os.path.join(P, f"r04_{tag}_bench_line_under_rocprof.json"), "w") as o:
            json.dump(bl, o)
    rows = list(csv.DictReader(open(f[0]))) if f else []
    return rows, bl


def counters(tag, kernel_prefixes=("kr_step",), last=20):
    f = glob.glob(os.path.join(G, f"r04_pmc_{tag}", "**", "*counter_collection.csv"), recursive=True)
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    if not f:
        return {}
    for row in csv.DictReader(open(f[0])):
        name = row["Kernel_Name"].replace("void ", "").split("<")[0].split("(")[0]
        if any(name.startswith(p) for p in kernel_prefixes):
            per[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v[-last:]) / len(v[-last:]) for c, v in d.items()} | {"_launches": len(next(iter(d.values())))}
            for k, d in per.items()}


def per_step_sum(tag, counter, steps):
    """Sum over ALL kernels of one env step (list family: ~27 launches): total of the counter over the last `steps` env
    steps' worth of every kernel's launches / steps."""
    f = glob.glob(os.path.join(G, f"r04_pmc_{tag}", "**", "*counter_collection.csv"), recursive=True)
    if not f:
        return None, {}
    per = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        if row["Counter_Name"] != counter:
            continue
        name = row["Kernel_Name"].replace("void ", "").split("<")[0].split("(")[0]
        if name.startswith("at::") or name.startswith("__amd") or "elementwise" in name:
            continue
        per[name].append(float(row["Counter_Value"]))
    total, by = 0.0, {}
    for k, v in per.items():
        n_per_step = max(1, round(len(v) / max(1, len(per.get("k_stats", per.get("kr_step", v))))))
        take = v[-steps * n_per_step:]
        by[k] = sum(take) / steps
        total += by[k]
    return total, by


def traffic_record(tag_f, tag_w, kernel, last, label):
    bl = bench_line(os.path.join(G, f"r04_pmc_{tag_f}.log"))
    cf = counters(tag_f, (kernel,), last).get(kernel, {})
    cw = counters(tag_w, (kernel,), last).get(kernel, {})
    if not (bl and "FETCH_SIZE" in cf and "WRITE_SIZE" in cw):
        return None
    fetch_b, write_b = cf["FETCH_SIZE"] * 1024.0, cw["WRITE_SIZE"] * 1024.0
    R, A = bl["config"]["replicas_per_gpu"], bl["config"]["agents"]
    rec = {"kernel": kernel, "replicas": R, "V_a": bl["config"]["V_a"], "command": label,
           "fetch_size_kb_per_launch": cf["FETCH_SIZE"], "write_size_kb_per_launch": cw["WRITE_SIZE"],
           "bytes_per_launch": 2.0 * fetch_b + write_b,
           "algorithmic_bytes_per_launch": bl["roofline"]["bytes_per_agent_env_step"] * R * A,
           "formula": f"2 x FETCH_SIZE + WRITE_SIZE (kB x 1024), mean of the last {last} launches",
           "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate kernel-trace-only passes (tools/collect_profiles_r04.sh); "
                  "FETCH_SIZE doubled (gfx950 tallies 128-byte requests at 64 bytes: MI355X_MICROARCH.md, HBM); "
                  "Infinity-Cache hits are included in both counters"}
    rec["ratio_to_algorithmic"] = rec["bytes_per_launch"] / rec["algorithmic_bytes_per_launch"]
    return rec


def main():
    os.makedirs(P, exist_ok=True)
    stats = {tag: kernel_stats(tag) for tag in ("resident", "list", "c16", "jinan")}
    # ---- kr_step counters
    allc = {}
    for tag in ("fetch", "write", "sq1", "sq2", "sq3", "tcc"):
        allc.update(counters(tag).get("kr_step", {}))
    allc.pop("_launches", None)
    krow = next((r for r in stats["resident"][0] if r["Name"].startswith("void kr_step") or r["Name"].startswith("kr_step")), None)
    if allc:
        with open(os.path.join(P, "r04_resident_pmc_summary.txt"), "w") as o:
            o.write("kr_step: mean per launch over the last 20 launches of `bench.py --no-cpu-baseline --no-extras --family resident "
                    "--steps 20 --warmup 5` (after the 150-step pre-roll), separate rocprofv3 --pmc passes, kernel-trace only\n")
            for c in sorted(allc):
                o.write(f"{c:28s} {allc[c]:18.1f}\n")
            if "SQ_WAVE_CYCLES" in allc:
                w = allc["SQ_WAVE_CYCLES"]
                o.write("\nshares of resident wave-cycles: waiting (s_waitcnt / barrier) %.1f %%, issue stalls %.1f %%, issuing %.1f %%\n" % (
                    100 * allc.get("SQ_WAIT_ANY", 0) / w, 100 * allc.get("SQ_WAIT_INST_ANY", 0) / w, 100 * allc.get("SQ_ACTIVE_INST_ANY", 0) / w))
            if "TCC_HIT_sum" in allc:
                o.write("L2 hit rate %.1f %%\n" % (100 * allc["TCC_HIT_sum"] / (allc["TCC_HIT_sum"] + allc["TCC_MISS_sum"])))
    # ---- the VALU record (roofline_valu)
    bl_sq = bench_line(os.path.join(G, "r04_pmc_sq1.log"))
    if bl_sq and "SQ_INSTS_VALU" in allc and "SQ_INSTS_SALU" in allc:
        dur_us = bl_sq["roofline"]["avg_launch_us"]
        cycles = dur_us * 1e3 * CLOCK_GHZ
        rec = {"kernel": "kr_step", "replicas": bl_sq["config"]["replicas_per_gpu"], "V_a": bl_sq["config"]["V_a"],
               "avg_launch_us_under_pmc": dur_us,
               "valu_busy_frac": allc["SQ_INSTS_VALU"] * 4.0 / (N_SIMD * cycles),
               "active_lane_frac": allc["SQ_THREAD_CYCLES_VALU"] / (64.0 * allc["SQ_ACTIVE_INST_VALU"]),
               "salu_per_valu": allc["SQ_INSTS_SALU"] / allc["SQ_INSTS_VALU"],
               "wait_frac": allc["SQ_WAIT_ANY"] / allc["SQ_WAVE_CYCLES"],
               "lds_bank_conflict_ratio": allc["SQ_LDS_BANK_CONFLICT"] / allc["SQ_ACTIVE_INST_LDS"],
               "waves_per_simd": 3,
               "how": "rocprofv3 --pmc SQ_* passes (kernel-trace only) of `bench.py --family resident --steps 20 --warmup 5`, mean of "
                      "the last 20 kr_step launches; valu_busy = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x launch duration x "
                      f"{CLOCK_GHZ} GHz); active lanes = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); waves per SIMD: one "
                      "768-thread workgroup per CU (LDS)"}
        rec["useful_valu_frac"] = rec["valu_busy_frac"] * rec["active_lane_frac"]
        # the counter pass itself slows the launch (its HIP-event duration is what the fractions above divide by); the same
        # instruction counts against the duration of an undisturbed run of the same command (the committed bench line):
        und = os.path.join(P, "r04_bench_line_driver_cmd.json")
        if os.path.exists(und):
            u_us = json.load(open(und))["roofline"]["avg_launch_us"]
            rec["avg_launch_us_undisturbed"] = u_us
            rec["valu_busy_frac_undisturbed"] = allc["SQ_INSTS_VALU"] * 4.0 / (N_SIMD * u_us * 1e3 * CLOCK_GHZ)
            rec["useful_valu_frac_undisturbed"] = rec["valu_busy_frac_undisturbed"] * rec["active_lane_frac"]
        rec["valu_insts_per_launch"] = allc["SQ_INSTS_VALU"]
        with open(os.path.join(P, "r04_valu_resident.json"), "w") as o:
            json.dump([rec], o, indent=1)
        print(json.dumps(rec, indent=1))
    # ---- traffic: headline (two states), Jinan
    recs = [r for r in (traffic_record("fetch", "write", "kr_step", 20, "bench.py --family resident --steps 20 --warmup 5"),
                        traffic_record("fetch240", "write240", "kr_step", 240, "bench.py --family resident --steps 240 --warmup 20")) if r]
    if recs:
        with open(os.path.join(P, "r04_traffic_resident.json"), "w") as o:
            json.dump(recs, o, indent=1)
        print(json.dumps([{k: r[k] for k in ("V_a", "bytes_per_launch", "ratio_to_algorithmic")} for r in recs]))
    rj = traffic_record("jinan_fetch", "jinan_write", "kr_step", 40, "bench.py --scenario jinan --replicas 8192 --steps 40 --warmup 5")
    if rj:
        with open(os.path.join(P, "r04_traffic_jinan.json"), "w") as o:
            json.dump([rj], o, indent=1)
        print("jinan", rj["ratio_to_algorithmic"])
    # ---- 16x16 (list family): all kernels of an env step together
    bl16 = bench_line(os.path.join(G, "r04_pmc_c16_fetch.log"))
    f16, by_f = per_step_sum("c16_fetch", "FETCH_SIZE", 10)
    w16, by_w = per_step_sum("c16_write", "WRITE_SIZE", 10)
    if bl16 and f16 and w16:
        R, A = bl16["config"]["replicas_per_gpu"], bl16["config"]["agents"]
        alg = bl16["roofline"]["bytes_per_agent_env_step"] * R * A
        rec = {"workload": bl16["config"]["workload"], "V_a": bl16["config"]["V_a"], "value": bl16["value"], "ms_per_step": bl16["ms_per_step"],
               "fetch_kb_per_env_step": f16, "write_kb_per_env_step": w16, "bytes_per_env_step": (2 * f16 + w16) * 1024.0,
               "algorithmic_bytes_per_env_step": alg, "ratio_to_algorithmic": (2 * f16 + w16) * 1024.0 / alg,
               "fetch_kb_by_kernel": by_f, "write_kb_by_kernel": by_w,
               "how": "sum over every kernel of an env step (list family, ~27 launches), last 10 env steps; FETCH_SIZE doubled as above"}
        with open(os.path.join(P, "r04_traffic_c16.json"), "w") as o:
            json.dump(rec, o, indent=1)
        print("16x16 traffic ratio", rec["ratio_to_algorithmic"])
    # ---- flow-set sweep
    sweep = []
    for fs in (8, 64, 512, 4096):
        bl = bench_line(os.path.join(G, f"r04_pmc_fs{fs}_fetch.log"))
        c = counters(f"fs{fs}_fetch").get("kr_step", {})
        if bl and "FETCH_SIZE" in c:
            sweep.append({"flow_sets": fs, "value_under_pmc": bl["value"], "ms_per_step": bl["ms_per_step"],
                          "avg_launch_us": bl["roofline"]["avg_launch_us"], "V_a": bl["config"]["V_a"],
                          "fetch_size_kb_per_launch": c["FETCH_SIZE"], "fetch_bytes_per_launch_x2": 2048.0 * c["FETCH_SIZE"]})
    if sweep:
        with open(os.path.join(P, "r04_flow_sets_sweep.json"), "w") as o:
            json.dump(sweep, o, indent=1)
        print(json.dumps(sweep))


if __name__ == "__main__":
    main()
