"""Engine-compatible object over the CPU oracle (fixture generation only).

Quacks like the 11 `cityflow.Engine` methods the reference's CityFlow backend
calls: pytsc/backends/cityflow/simulator.py:71-95,
retriever.py:35,94-112, traffic_signal.py:31,58.
"""
import json
import os

from oracle.pyoracle import Oracle
from pytsc_amd.config import Config
from pytsc_amd.scenario import build_tables


class Engine:
    def __init__(self, config_file, thread_num=1):
        with open(config_file, "r") as f:
            cfg = json.load(f)
        self.cfg = cfg
        config = Config(
            "shim", scenario_dir=cfg["dir"],
            hip={"roadnet_file": cfg["roadnetFile"], "flow_file": cfg["flowFile"],
                 "flow_rate_type": "constant", "interval": cfg["interval"], "initial_wait_time": 0},
        )
        self.tables = build_tables(config, n_replicas=1)
        self.oracle = Oracle(self.tables)
        self.rn = self.tables.rn
        self._info = None

    def next_step(self):
        self.oracle.next_step(0)
        self._info = None

    def reset(self):
        self.oracle.reset()

    def get_current_time(self):
        return self.oracle.lib.tsco_engine_current_time(self.oracle.h, 0)

    def set_tl_phase(self, intersection_id, phase_id):
        self.oracle.set_tl_phase(0, self.rn.agent_index[intersection_id], int(phase_id))

    def _lanes(self):
        out = {}
        for k, lane in enumerate(self.rn.lane_ids):  # sorted, like std::map keys
            out[lane] = self.oracle.lane_vehicles(0, k)
        return out

    def get_lane_waiting_vehicle_count(self):
        return {l: int((s < 0.1).sum()) for l, (v, d, s) in self._lanes().items()}

    def get_lane_vehicles(self):
        return {l: [f"v{i}" for i in v] for l, (v, d, s) in self._lanes().items()}

    def get_vehicle_speed(self):
        out = {}
        for l, (v, d, s) in self._lanes().items():
            for i, sp in zip(v, s):
                out[f"v{i}"] = float(sp)
        # vehicles on lane-links are running too
        drv, dist, speed, vid = self.oracle.snapshot(0)
        for i, sp in zip(vid, speed):
            out[f"v{i}"] = float(sp)
        return out

    def get_vehicle_info(self, vehicle_id):
        if self._info is None:
            drv, dist, speed, vid = self.oracle.snapshot(0)
            self._info = {int(v): (float(d), float(s)) for v, d, s in zip(vid, dist, speed)}
        d, s = self._info[int(vehicle_id[1:])]
        # std::to_string(double): "%f"
        return {"running": "1", "distance": "%f" % d, "speed": "%f" % s}

    def get_vehicle_count(self):
        return self.oracle.lib.tsco_engine_vehicle_count(self.oracle.h, 0)

    def get_average_travel_time(self):
        return self.oracle.lib.tsco_engine_average_travel_time(self.oracle.h, 0)
