"""Simulator façade (reference: pytsc/common/simulator.py:4-65,
pytsc/backends/cityflow/simulator.py:8-95) over the batched HIP engine."""
import numpy as np
import torch

from pytsc_amd.backends.hip.retriever import Retriever
from pytsc_amd.scenario import build_tables


def default_engine_factory(tables, device):
    from pytsc_amd.vec_env import VecEngine  # raises without a GPU / without the HIP library

    return VecEngine(tables, device=device)


class Simulator:
    def __init__(self, parsed_network, engine_factory=None, replica=0):
        self.parsed_network = parsed_network
        self.config = parsed_network.config
        self.engine = None
        self.engine_factory = engine_factory or default_engine_factory
        self.replica = replica
        self.agent_index = {t: i for i, t in enumerate(parsed_network.rn.agent_ids)}
        self.n_agents = len(self.agent_index)
        self.stepped = False
        self._staged = {}
        self._cache = {}
        self._measurements = None
        self.n_starts = 0

    # ---- lifecycle ---------------------------------------------------------
    def start_simulator(self):
        self.config._set_flow_file()
        sim = self.config.simulator
        n_replicas = int(sim.get("n_replicas", 1))
        if self.engine is None:
            files = list(sim.get("flow_files") or [])
            set_offset = 0
            if sim.get("flow_rate_type", "constant") != "constant" and self.config.flow_file in files:
                set_offset = files.index(self.config.flow_file)
            self.tables = build_tables(self.config, n_replicas=n_replicas,
                                       roadnet=self.parsed_network.rn, set_offset=set_offset)
            # the façade always hands target phase indices to the engine
            # (action_to_phase); masks are evaluated host-side from the phase state
            self.tables.cfg.action_space = 0
            self.engine = self.engine_factory(self.tables, int(sim.get("device", 0)))
        else:
            # restart in place; for `random` / `sequential` demand the flow file the config just
            # chose (reference config.py:63-76) is handed to the engine as the replicas' flow set
            files = list(sim.get("flow_files") or [])
            sets = None
            if sim.get("flow_rate_type", "constant") != "constant" and self.config.flow_file in files:
                sets = np.full(n_replicas, files.index(self.config.flow_file), dtype=np.int32)
            if sets is not None:
                self.engine.reset(sets=torch.as_tensor(sets))
            else:
                self.engine.reset()
        self.cityflow_retriever = self.retriever = Retriever(self)
        self.stepped = False
        self._staged = {}
        self.n_starts += 1
        self._refresh(self.engine.observe())

    def close_simulator(self):
        pass  # the next start_simulator() resets the replicas in place

    def shutdown(self):
        if self.engine is not None:
            self.engine.close()
            self.engine = None

    # ---- stepping ----------------------------------------------------------
    def stage_action(self, agent_index, phase_index):
        self._staged[agent_index] = phase_index

    def simulator_step(self, n_steps=None):
        delta = self.config.simulator["delta_time"]
        if n_steps is None:
            n_steps = delta
        if not n_steps:
            return
        if n_steps != delta:
            raise ValueError(f"the HIP engine advances delta_time={delta} sub-steps per call")
        acts = np.full((self.tables.n_replicas, self.n_agents), -1, dtype=np.int32)
        for a, p in self._staged.items():
            acts[:, a] = p
        self._staged = {}
        dev = getattr(self.engine, "device", "cpu")
        buf = self.engine.step(torch.as_tensor(acts, dtype=torch.int32, device=dev))
        self.stepped = True
        self._refresh(buf)

    def _refresh(self, buf):
        self._buf = buf
        self._cache = {}
        self._measurements = None

    def retrieve_step_measurements(self):
        self._measurements = None

    # ---- host mirror ---------------------------------------------------------
    def host(self, name):
        if name not in self._cache:
            t = self._buf[name][self.replica]
            self._cache[name] = t.cpu().numpy() if hasattr(t, "cpu") else np.asarray(t)
        return self._cache[name]

    def host_scalar_vehicles(self):
        if "n_veh" not in self._cache:
            d, _, _, _ = self.engine.snapshot(self.replica)
            self._cache["n_veh"] = len(d)
        return self._cache["n_veh"]

    @property
    def step_measurements(self):
        if self._measurements is None:
            self._measurements = {
                "lane": self.retriever.retrieve_lane_measurements(),
                "sim": self.retriever.retrieve_sim_measurements(),
            }
        return self._measurements

    # ---- clocks ----------------------------------------------------------------
    @property
    def sim_time(self):
        return float(self.host("stats")[0])

    @property
    def sim_step(self):
        return self.sim_time - self.config.simulator["initial_wait_time"]

    @property
    def is_terminated(self):
        return self.sim_step == self.config.simulator["sim_length"]
