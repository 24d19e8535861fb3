"""gpurun_out/r02_* (tools/collect_profiles_r02.sh) -> profiles/r02_*: kernel stats of both families, the counter
summary of kr_step and the traffic record bench.py quotes (`roofline.traffic`)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(REPO, "gpurun_out")
P = os.path.join(REPO, "profiles")


def kernel_stats(fam):
    f = glob.glob(os.path.join(G, f"r02_stats_{fam}", "**", "*kernel_stats.csv"), recursive=True)
    if f:
        shutil.copy(f[0], os.path.join(P, f"r02_{fam}_kernel_stats.csv"))
    log = os.path.join(G, f"r02_stats_{fam}.log")
    for line in open(log):
        if line.startswith("{"):
            with open(os.path.join(P, f"r02_{fam}_bench_line_under_rocprof.json"), "w") as o:
                o.write(line)


def counters(tag, last=20):
    f = glob.glob(os.path.join(G, f"r02_pmc_{tag}", "**", "*counter_collection.csv"), recursive=True)
    per = collections.defaultdict(list)
    if not f:
        return {}
    for row in csv.DictReader(open(f[0])):
        if row["Kernel_Name"].startswith("void kr_step") or row["Kernel_Name"].startswith("kr_step"):
            per[row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {c: sum(v[-last:]) / len(v[-last:]) for c, v in per.items()}


def bench_line(tag):
    for line in open(os.path.join(G, f"r02_pmc_{tag}.log")):
        if line.startswith("{"):
            return json.loads(line)
    return None


def main():
    os.makedirs(P, exist_ok=True)
    for fam in ("resident", "list"):
        kernel_stats(fam)
    allc = {}
    for tag in ("fetch", "write", "sq1", "sq2", "tcc"):
        allc.update(counters(tag))
    with open(os.path.join(P, "r02_resident_pmc_summary.txt"), "w") as o:
        o.write("kr_step: mean per launch over the last 20 launches of `bench.py --no-cpu-baseline --family resident --steps 20 "
                "--warmup 5` (after the 150-step pre-roll), separate rocprofv3 --pmc passes, kernel-trace only\n")
        for c in sorted(allc):
            o.write(f"{c:28s} {allc[c]:18.1f}\n")
        if "SQ_WAVE_CYCLES" in allc:
            w = allc["SQ_WAVE_CYCLES"]
            o.write("\nshares of resident wave-cycles: waiting (s_waitcnt / barrier) %.1f %%, issue stalls %.1f %%, issuing %.1f %%\n" % (
                100 * allc.get("SQ_WAIT_ANY", 0) / w, 100 * allc.get("SQ_WAIT_INST_ANY", 0) / w, 100 * allc.get("SQ_ACTIVE_INST_ANY", 0) / w))
        if "TCC_HIT_sum" in allc:
            o.write("L2 hit rate %.1f %%\n" % (100 * allc["TCC_HIT_sum"] / (allc["TCC_HIT_sum"] + allc["TCC_MISS_sum"])))
    recs = []
    for tag_f, tag_w, last in (("fetch", "write", 20), ("fetch240", "write240", 240)):
        bl = bench_line(tag_f) if os.path.exists(os.path.join(G, f"r02_pmc_{tag_f}.log")) else None
        cf, cw = counters(tag_f, last), counters(tag_w, last)
        if not (bl and "FETCH_SIZE" in cf and "WRITE_SIZE" in cw):
            continue
        # FETCH_SIZE / WRITE_SIZE are in kilobytes; gfx950 tallies 128-byte read requests at 64 bytes: doubled
        fetch_b, write_b = cf["FETCH_SIZE"] * 1024.0, cw["WRITE_SIZE"] * 1024.0
        rec = {"kernel": "kr_step", "replicas": bl["config"]["replicas_per_gpu"], "V_a": bl["config"]["V_a"],
               "command": f"bench.py --no-cpu-baseline --family resident --steps {bl['steps']} --warmup {bl['warmup']}",
               "fetch_size_kb_per_launch": cf["FETCH_SIZE"], "write_size_kb_per_launch": cw["WRITE_SIZE"],
               "bytes_per_launch": 2.0 * fetch_b + write_b,
               "algorithmic_bytes_per_launch": bl["roofline"]["bytes_per_agent_env_step"] * bl["config"]["replicas_per_gpu"] * bl["config"]["agents"],
               "formula": f"2 x FETCH_SIZE + WRITE_SIZE (kB x 1024), mean of the last {last} launches",
               "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate kernel-trace-only passes (tools/collect_profiles_r02.sh); "
                      "FETCH_SIZE doubled (gfx950 tallies 128-byte requests at 64 bytes: MI355X_MICROARCH.md, HBM); "
                      "Infinity-Cache hits are included in both counters"}
        rec["ratio_to_algorithmic"] = rec["bytes_per_launch"] / rec["algorithmic_bytes_per_launch"]
        recs.append(rec)
    if recs:
        with open(os.path.join(P, "r02_traffic_resident.json"), "w") as o:
            json.dump(recs, o, indent=1)
        print(json.dumps(recs, indent=1))
    src = os.path.join(G, "r02_bench_line.json")
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, "r02_bench_line.json"))


if __name__ == "__main__":
    main()
