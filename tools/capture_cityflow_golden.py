"""CityFlow pin kit, part 1: record traces of a REAL CityFlow for the golden cases.

The engine half of the oracle (oracle/tsc_oracle.c, rows E1-E6 of SURVEY.md §8a) restates third-party CityFlow from
its published algorithm; CityFlow itself is in no container of this project, so that half is "parity unpinned".
This script closes the gap wherever `import cityflow` works (any machine with the reference's backend installed):

    python tools/capture_cityflow_golden.py            # every case under tests/golden/
    python tools/capture_cityflow_golden.py syn_2x2_fixed_time jinan_3_4_random

For each case it feeds CityFlow exactly what the fixture fed the reference's Python layer -- the same road network,
the same flow file, the recorded phase of every signal at every env step through set_tl_phase (call sites
pytsc/backends/cityflow/traffic_signal.py:31,58), delta_time x next_step (simulator.py:80-89) -- and dumps what the
reference's retriever reads (retriever.py:35,94-112): per lane the vehicle count and the waiting count every step,
every 6th step each vehicle's lane, distance and speed, plus get_vehicle_count / get_average_travel_time /
get_current_time.  Output: tests/golden/cityflow/<case>.npz.  Commit those files; tests/test_cityflow_pin.py then
replays them against the oracle (and skips while they are absent).

`--engine-module NAME` imports the Engine from another module (the self-test of the kit uses the oracle-backed
stand-in tests/golden/ref_stubs/cityflow.py, which must reproduce the oracle exactly).
"""
import argparse
import gzip
import importlib
import json
import os
import re
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)


def vehicle_key(vehicle_id, entry_of_vid=None):
    """(flow entry, number within the entry) of a vehicle.  CityFlow names vehicles flow_<entry>_<k>; the stand-in
    names them v<creation-order id> and needs the tables' map back to entries."""
    m = re.match(r"^flow_(\d+)_(\d+)$", vehicle_id)
    if m:
        return int(m.group(1)), int(m.group(2))
    m = re.match(r"^v(\d+)$", vehicle_id)
    if m and entry_of_vid is not None:
        return entry_of_vid[int(m.group(1))]
    raise ValueError(f"unexpected vehicle id {vehicle_id!r}")


def capture(case, Engine, out_dir, snap_every=6):
    from pytsc_amd.flows import expand_flows
    from pytsc_amd.grid import generate_grid

    d = os.path.join(GOLDEN, case)
    with gzip.open(os.path.join(d, "inputs.json.gz"), "rt") as f:
        inputs = json.load(f)
    exp = np.load(os.path.join(d, "expected.npz"))
    with open(os.path.join(d, "parser.json")) as f:
        parser = json.load(f)
    net = generate_grid(*inputs["roadnet"]["grid"]) if "grid" in inputs["roadnet"] else inputs["roadnet"]["net"]
    entries = expand_flows(inputs["flows"])
    lanes = parser["lanes"]
    lane_idx = {l: i for i, l in enumerate(lanes)}
    agents = parser["agent_order"]
    phases = [parser["signals"][a]["phases"] for a in agents]  # PyTSC phase index -> raw CityFlow phase id
    root = tempfile.mkdtemp(prefix="cf_capture_")
    with open(os.path.join(root, "roadnet.json"), "w") as f:
        json.dump(net, f)
    with open(os.path.join(root, "flows.json"), "w") as f:
        json.dump(entries, f)
    cfg = {"dir": root + "/", "roadnetFile": "roadnet.json", "flowFile": "flows.json", "interval": 1.0, "seed": 0,
           "rlTrafficLight": True, "laneChange": False, "saveReplay": False,
           "roadnetLogFile": "roadnet_log_file.json", "replayLogFile": "replay_log_file.txt"}  # config.py:88-99
    cfg_path = os.path.join(root, "cityflow_cfg.json")
    with open(cfg_path, "w") as f:
        json.dump(cfg, f)
    eng = Engine(config_file=cfg_path, thread_num=1)
    entry_of_vid = None
    if hasattr(eng, "tables"):  # the stand-in: creation-order ids -> (entry, k)
        fs = eng.tables.fc.sets[0]
        entry_of_vid = list(zip(fs.veh_entry.tolist(), fs.veh_k.tolist()))
    T = int(inputs["steps"])
    delta_time = 5
    lane_n = np.zeros((T, len(lanes)), np.int32)
    lane_q = np.zeros((T, len(lanes)), np.int32)
    veh_count = np.zeros(T, np.int64)
    avg_tt = np.zeros(T, np.float64)
    now = np.zeros(T, np.float64)
    snap_steps, snap_ptr, s_flow, s_k, s_lane, s_dist, s_speed = [], [0], [], [], [], [], []
    for s in range(T):
        for a, agent in enumerate(agents):
            eng.set_tl_phase(agent, int(phases[a][int(exp["phase"][s][a][0])]))
        for _ in range(delta_time):
            eng.next_step()
        lv = eng.get_lane_vehicles()
        lw = eng.get_lane_waiting_vehicle_count()
        for l, i in lane_idx.items():
            lane_n[s, i] = len(lv[l])
            lane_q[s, i] = lw[l]
        veh_count[s] = eng.get_vehicle_count()
        avg_tt[s] = eng.get_average_travel_time()
        now[s] = eng.get_current_time()
        if s % snap_every == snap_every - 1:
            rows = []
            for l, i in lane_idx.items():
                for v in lv[l]:
                    info = eng.get_vehicle_info(v)
                    e, k = vehicle_key(v, entry_of_vid)
                    rows.append((e, k, i, float(info["distance"]), float(info["speed"])))
            rows.sort()
            snap_steps.append(s)
            snap_ptr.append(snap_ptr[-1] + len(rows))
            s_flow += [r[0] for r in rows]
            s_k += [r[1] for r in rows]
            s_lane += [r[2] for r in rows]
            s_dist += [r[3] for r in rows]
            s_speed += [r[4] for r in rows]
        if bool(exp["done"][s]) and bool(exp["terminated"][s]):
            eng.reset()  # the reference restarts the simulated hour (pytsc/__init__.py:164-176)
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, case + ".npz")
    np.savez_compressed(out, lane_n=lane_n, lane_q=lane_q, vehicle_count=veh_count, average_travel_time=avg_tt, time=now,
                        snap_steps=np.asarray(snap_steps, np.int32), snap_ptr=np.asarray(snap_ptr, np.int64),
                        snap_flow=np.asarray(s_flow, np.int32), snap_k=np.asarray(s_k, np.int32),
                        snap_lane=np.asarray(s_lane, np.int32),
                        snap_dist=np.asarray(s_dist, np.float64), snap_speed=np.asarray(s_speed, np.float64),
                        engine=np.asarray(getattr(Engine, "__module__", "?")))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="*")
    ap.add_argument("--engine-module", default="cityflow")
    ap.add_argument("--out", default=os.path.join(GOLDEN, "cityflow"))
    args = ap.parse_args()
    Engine = importlib.import_module(args.engine_module).Engine
    cases = args.cases or sorted(c for c in os.listdir(GOLDEN) if os.path.exists(os.path.join(GOLDEN, c, "expected.npz")))
    for case in cases:
        print(case, "->", capture(case, Engine, args.out), flush=True)


if __name__ == "__main__":
    main()
