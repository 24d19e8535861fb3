"""Which of the reference's shipped CityFlow scenarios does the HIP path accept, and with which kernel family?

Build-container tool (reads the scenario DATA files under /root/reference/pytsc/scenarios/cityflow; runs no
reference code): compiles every shipped road network with Poisson demand into tables and asks the library's pure-host
check (tsc_check_tables, the same create-time requirements tsc_create applies).  Writes
tests/golden/shipped_scenarios_support.json, which tests/test_shipped_scenarios.py compares against on every run, and
gzip copies of a few small road networks under tests/golden/roadnets/ for the GPU test of the same file.
"""
import ctypes as C
import glob
import gzip
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF = "/root/reference/pytsc/scenarios/cityflow"
OUT = os.path.join(REPO, "tests", "golden", "shipped_scenarios_support.json")
SHIP = {"syn_5x5_oneway": "syn_5x5_roadnet.json", "1x5_cityflow_grid": "1x5_roadnet.json", "syn_1x7": "syn_1x7_roadnet.json",
        "hangzhou_4_4": "roadnet_4_4.json", "syn_1x1": "1x1_roadnet.json"}


def check(net, n_replicas=4, rate=400.0, family="auto"):
    from pytsc_amd import _lib
    from pytsc_amd.config import Config
    from pytsc_amd.flows import poisson_flows
    from pytsc_amd.roadnet import RoadNet
    from pytsc_amd.scenario import build_tables

    rn = RoadNet(net)
    entries = poisson_flows(rn, rate, 600, seed=0)
    cfg = Config("shipped", signal={"action_space": "phase_selection", "round_robin": False},
                 hip={"roadnet_json": net, "flow_entries": [entries], "sim_length": 600, "kernel_family": family})
    tables = build_tables(cfg, n_replicas=n_replicas)
    lib = _lib.load()
    fam, cap = C.c_int32(0), C.c_int32(0)
    rc = lib.tsc_check_tables(C.byref(tables.net), C.byref(tables.flows), C.byref(tables.cfg), C.byref(fam), C.byref(cap))
    return dict(rc=int(rc), family={0: None, 1: "list", 2: "resident"}[fam.value if rc == 0 else 0],
                resident_capacity=int(cap.value), agents=int(tables.n_agents), lanes=int(tables.n_lanes),
                links=int(tables.net.n_links)), tables


def main():
    res = {}
    for d in sorted(os.listdir(REF)):
        nets = sorted(glob.glob(os.path.join(REF, d, "*roadnet*.json")))
        nets = [n for n in nets if "log" not in os.path.basename(n) and "tl_plan" not in os.path.basename(n)]
        for path in nets:
            key = f"{d}/{os.path.basename(path)}"
            try:
                with open(path) as f:
                    net = json.load(f)
                if "intersections" not in net:  # a flow file that carries "roadnet" in its name
                    continue
                res[key], _ = check(net)
            except (KeyError, ValueError) as e:  # a road link names a road / lane the file does not define
                res[key] = dict(rc=None, note=f"inconsistent file: {type(e).__name__}: {e}"[:200])
            print(key, res[key], flush=True)
            if d in SHIP and os.path.basename(path) == SHIP[d] and res[key].get("rc") == 0:
                os.makedirs(os.path.join(REPO, "tests", "golden", "roadnets"), exist_ok=True)
                with gzip.open(os.path.join(REPO, "tests", "golden", "roadnets", d + ".json.gz"), "wt", compresslevel=9) as f:
                    json.dump(net, f, separators=(",", ":"))
    with open(OUT, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
