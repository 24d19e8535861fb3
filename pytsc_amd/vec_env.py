"""Batched environment: R replicas of one signalised network stepped together on
one MI355X through the C-ABI (include/pytsc_amd.h).  Outputs are torch tensors
that live in HBM; nothing on the step path touches the host.

API mirrors the reference façade (pytsc/__init__.py:17-182) with a leading
replica axis: step(actions[R, A]) -> (global_reward[R], done[R], stats[R, 8]),
get_observations() -> [R, A, obs], get_state(), get_rewards(), get_action_mask().
"""
import ctypes as C

import numpy as np
import torch

from pytsc_amd import _lib
from pytsc_amd.tables import N_AGENT_STATS, N_STATS, TscStepOut

STAT_KEYS = ("time_step", "average_travel_time", "n_queued", "mean_speed", "mean_delay",
             "density", "pressure", "network_flow")


class VecEngine:
    """Thin owner of a tsc_handle_t plus its caller-owned device output buffers."""

    def __init__(self, tables, device=0, want=("obs", "state", "mask", "local_reward", "global_reward",
                                              "done", "terminated", "stats", "agent_stats", "phase",
                                              "local_reward64", "global_reward64", "valid", "phase_scores"), family="auto"):
        if getattr(tables, "phase_moves", None) is None:  # (a signal with more than 16 lanes on a side: no fused scores)
            want = tuple(k for k in want if k != "phase_scores")
        if not torch.cuda.is_available():
            raise RuntimeError("pytsc_amd needs a ROCm GPU (MI355X); there is no CPU fallback")
        self.lib = _lib.load()
        self.tables = tables
        self.device = torch.device("cuda", device)
        self.R, self.A = tables.n_replicas, tables.n_agents
        # kernel family (include/pytsc_amd.h TSC_FAMILY_*): "auto" keeps what the tables / config ask for
        if family not in ("auto", "list", "resident"):
            raise ValueError(f"family {family!r}: expected auto, list or resident")
        if family != "auto":
            tables.cfg.kernel_family = {"list": 1, "resident": 2}[family]
        h = C.c_void_p()
        _lib.check(self.lib.tsc_create(C.byref(tables.net), C.byref(tables.flows), C.byref(tables.cfg),
                                       device, C.byref(h)), "tsc_create")
        self.h = h
        self.n_actions = self.lib.tsc_n_actions(h)
        # the dominant kernel of the family: what tsc_profile_enable brackets and bench.py prices
        self._kernel_infos = {
            "list": dict(family="list", name="k_exec", substeps_per_launch=1, traffic_file="r04_traffic_list.json", valu_file="none",
                         note="dense 64-item batches of lane / lane-link work items drawn from all replicas; "
                              "one of ~22 launches per env step"),
            "resident": dict(family="resident", name="kr_step", substeps_per_launch=5,
                             traffic_file="r05_traffic_resident.json", valu_file="r05_valu_resident.json",
                             note="one workgroup per replica, state resident in LDS over the 5 sub-steps and the measurement; an env "
                                  "step = the small class's launch (512 threads, two workgroups per CU) and, behind it, the full-size "
                                  "class's (768 threads; it also redoes a step the small class abandoned): timed and counted together"),
        }
        self.obs_size = self.lib.tsc_obs_size(h)
        self.state_size = self.lib.tsc_state_size(h)
        R, A, dev = self.R, self.A, self.device
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        shapes = {
            "obs": ((R, A, self.obs_size), torch.float32),
            "state": ((R, A, self.state_size), torch.float32),
            "mask": ((R, A, self.n_actions), torch.uint8),
            "local_reward": ((R, A), torch.float32),
            "global_reward": ((R,), torch.float32),
            "done": ((R,), torch.uint8),
            "terminated": ((R,), torch.uint8),
            "stats": ((R, N_STATS), torch.float64),
            "agent_stats": ((R, A, N_AGENT_STATS), torch.float64),
            "phase": ((R, A, 3), torch.int32),
            "local_reward64": ((R, A), torch.float64),
            "global_reward64": ((R,), torch.float64),
            # 1: the replica's rows are those of the call just made; 0: it is stalled (include/pytsc_amd.h, valid_dev)
            "valid": ((R,), torch.uint8),
            # per-phase scores of the rule-based controllers (include/pytsc_amd.h, tsc_step_out_t::phase_scores_dev):
            # [..., 0] stopped vehicles, [..., 1] vehicles in the incoming windows, [..., 2] pressure of the phase's movements
            "phase_scores": ((R, A, 20, 3), torch.int16),
        }
        self.buf = {k: z(*shapes[k]) for k in want}
        ptr = lambda k: self.buf[k].data_ptr() if k in self.buf else None
        self.out = TscStepOut(ptr("obs"), ptr("state"), ptr("mask"), ptr("local_reward"),
                              ptr("global_reward"), ptr("done"), ptr("terminated"), ptr("stats"),
                              ptr("agent_stats"), ptr("phase"), ptr("local_reward64"),
                              ptr("global_reward64"), ptr("valid"), ptr("phase_scores"))

    @property
    def family(self):
        """The kernel family the handle runs NOW: an `auto` handle that started resident moves to the list family when a
        replica comes close to its record capacity (include/pytsc_amd.h, TSC_FAMILY_*)."""
        return {1: "list", 2: "resident"}[self.lib.tsc_kernel_family(self.h)]

    @property
    def kernel_variant(self):
        """Instantiation of the resident kernel the handle launches (include/pytsc_amd.h, tsc_kernel_variant)."""
        return {-1: "list family", 0: "generic", 1: "layout", 2: "layout + default vehicle"}[self.lib.tsc_kernel_variant(self.h)]

    @property
    def kernel_info(self):
        return self._kernel_infos[self.family]

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if getattr(self, "h", None):
            self.lib.tsc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, mask=None, sets=None):
        """Restart the replicas selected by `mask` ([R] uint8, None = all).  `sets` ([R] int32)
        names the flow set each restarted replica uses from now on (flow_rate_type: random,
        reference backends/cityflow/config.py:63-76); None / negative entries keep the handle's rule."""
        m = s = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            m = C.c_void_p(mask.data_ptr())
        if sets is not None:
            sets = torch.as_tensor(sets).to(device=self.device, dtype=torch.int32).contiguous()
            if sets.shape != (self.R,):
                raise ValueError(f"sets must be [{self.R}]")
            s = C.c_void_p(sets.data_ptr())
        self._last_reset_args = (mask, sets)  # keep alive until the kernels have run
        _lib.check(self.lib.tsc_reset_sets(self.h, m, s, self._stream()), "tsc_reset_sets")

    def park(self, mask):
        """Replicas with mask[r] != 0 are skipped by every later step / observe (valid = 0) until a reset that selects them
        (include/pytsc_amd.h, tsc_park; both kernel families)."""
        mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
        self._last_park = mask
        _lib.check(self.lib.tsc_park(self.h, C.c_void_p(mask.data_ptr()), self._stream()), "tsc_park")

    def observe(self):
        _lib.check(self.lib.tsc_observe(self.h, C.byref(self.out), self._stream()), "tsc_observe")
        return self.buf

    def step(self, actions):
        if actions.dtype != torch.int32 or not actions.is_contiguous() or actions.device != self.device:
            actions = actions.to(device=self.device, dtype=torch.int32).contiguous()
        if actions.shape != (self.R, self.A):
            raise ValueError(f"actions must be [{self.R}, {self.A}]")
        self._last_actions = actions  # keep alive until the kernels have run
        _lib.check(self.lib.tsc_step(self.h, C.c_void_p(actions.data_ptr()), C.byref(self.out),
                                     self._stream()), "tsc_step")
        return self.buf

    def lane_detail(self, with_posmat=False):
        R, nl, mb = self.R, self.tables.n_lanes, self.tables.max_bins
        counts = torch.zeros((R, nl, 2), dtype=torch.int32, device=self.device)
        speed = torch.zeros((R, nl), dtype=torch.float64, device=self.device)
        pm = torch.zeros((R, nl, mb), dtype=torch.float64, device=self.device) if with_posmat else None
        _lib.check(self.lib.tsc_lane_detail(self.h, C.c_void_p(counts.data_ptr()),
                                            C.c_void_p(speed.data_ptr()),
                                            C.c_void_p(pm.data_ptr()) if with_posmat else None, mb,
                                            self._stream()), "tsc_lane_detail")
        return counts, speed, pm

    def lane_window_counts(self):
        """[R, n_lanes, 3] int32: (# == 0.0, # >= 0 in the incoming window, # >= 0 in the outgoing window)."""
        out = torch.zeros((self.R, self.tables.n_lanes, 3), dtype=torch.int32, device=self.device)
        _lib.check(self.lib.tsc_lane_window_counts(self.h, C.c_void_p(out.data_ptr()), self._stream()),
                   "tsc_lane_window_counts")
        return out

    def snapshot(self, rep=0, cap=65536):
        drv = np.zeros(cap, np.int32)
        dist = np.zeros(cap, np.float64)
        speed = np.zeros(cap, np.float64)
        vid = np.zeros(cap, np.int32)
        torch.cuda.synchronize(self.device)
        n = self.lib.tsc_snapshot(self.h, rep, cap, drv.ctypes.data, dist.ctypes.data,
                                  speed.ctypes.data, vid.ctypes.data)
        if n < 0:
            _lib.check(n, "tsc_snapshot")
        return drv[:n], dist[:n], speed[:n], vid[:n]

    def random_valid_actions(self, mask, seed=0, out=None):
        """[R, A] int32: a uniformly random allowed action per signal, one launch
        (tsc_random_valid_actions; reference RandomController, controllers.py:245-268)."""
        if out is None:
            out = torch.empty(mask.shape[:2], dtype=torch.int32, device=self.device)
        assert mask.dtype == torch.uint8 and mask.is_contiguous() and out.is_contiguous()
        _lib.check(self.lib.tsc_random_valid_actions(self.h, mask.data_ptr(), int(seed), out.data_ptr(),
                                                     torch.cuda.current_stream(self.device).cuda_stream),
                   "tsc_random_valid_actions")
        return out

    CONTROLLERS = {"fixed_time": 0, "greedy": 1, "max_pressure": 2, "sotl": 3, "random": 4}

    def controller_actions(self, kind, buf=None, params=(), seed=0, out=None):
        """[R, A] int32: the reference's rule-based controller `kind` (pytsc/controllers/controllers.py:26-268) for every
        replica and signal in ONE launch (tsc_controller_actions), from the `phase` and `phase_scores` outputs of the step
        just made (the engine must have been created with both in `want`).  params: fixed_time (green_time = 25,);
        sotl (theta = 3, mu = 4, phi_min = 5)."""
        buf = self.buf if buf is None else buf
        k = self.CONTROLLERS[kind]
        p = list(params) + [0, 0, 0]
        if kind == "fixed_time" and not params:
            p[0] = 25
        if kind == "sotl" and not params:
            p[:3] = [3, 4, 5]
        if out is None:
            out = torch.empty((self.R, self.A), dtype=torch.int32, device=self.device)
        sc = buf.get("phase_scores")
        _lib.check(self.lib.tsc_controller_actions(self.h, k, int(p[0]), int(p[1]), int(p[2]), int(seed), buf["phase"].data_ptr(),
                                                   sc.data_ptr() if sc is not None else None, out.data_ptr(),
                                                   torch.cuda.current_stream(self.device).cuda_stream), "tsc_controller_actions")
        return out

    def class_counts(self):
        """(replicas in the small class, in the full-size class) of a two-class handle (tsc_class_counts)."""
        c = (C.c_int32 * 2)()
        _lib.check(self.lib.tsc_class_counts(self.h, c), "tsc_class_counts")
        return int(c[0]), int(c[1])

    def replica_counts(self):
        """[R, 3] int32: vehicles entered so far, running, finished (entered == running + finished)."""
        out = torch.zeros((self.R, 3), dtype=torch.int32, device=self.device)
        _lib.check(self.lib.tsc_replica_counts(self.h, C.c_void_p(out.data_ptr()), self._stream()),
                   "tsc_replica_counts")
        return out

    def errors(self):
        e = C.c_int32(0)
        _lib.check(self.lib.tsc_errors(self.h, C.byref(e)), "tsc_errors")
        return e.value

    PATH_KEYS = ("two_phase", "walk", "serial_scan", "conflict", "dirty", "stalled", "class_redo", "class_change")

    def path_counts(self):
        """dict: how often the resident family took each of its data-dependent paths since create
        (include/pytsc_amd.h, tsc_path_counts); synchronous."""
        c = (C.c_int64 * 8)()
        _lib.check(self.lib.tsc_path_counts(self.h, c), "tsc_path_counts")
        return {k: int(c[i]) for i, k in enumerate(self.PATH_KEYS)}

    def spill_counts(self):
        """dict: the spill pool of an `auto` handle on the resident family (include/pytsc_amd.h, tsc_spill_counts)."""
        c = (C.c_int32 * 4)()
        _lib.check(self.lib.tsc_spill_counts(self.h, c), "tsc_spill_counts")
        return dict(zip(("live", "slots", "capacity", "limit"), (int(x) for x in c)))

    def mean_vehicles(self):
        m = C.c_double(0)
        _lib.check(self.lib.tsc_mean_vehicles(self.h, C.byref(m)), "tsc_mean_vehicles")
        return m.value

    def profile(self, on):
        _lib.check(self.lib.tsc_profile_enable(self.h, int(on)), "tsc_profile_enable")

    def profile_read(self):
        ms, n = C.c_double(0), C.c_int32(0)
        _lib.check(self.lib.tsc_profile_read(self.h, C.byref(ms), C.byref(n)), "tsc_profile_read")
        return ms.value, n.value


class VecTrafficSignalNetwork:
    """Batched façade: the reference's TrafficSignalNetwork API
    (pytsc/__init__.py:17-182) with a leading replica axis and torch tensors in
    HBM instead of Python lists.  Uses the scenario's configured action space
    and its device-side masks."""

    def __init__(self, scenario, simulator_backend="hip", n_replicas=1024, device=0, set_offset=0,
                 outputs=("obs", "state", "mask", "local_reward", "global_reward", "done", "terminated",
                          "stats", "valid"), **kwargs):
        from pytsc_amd.config import Config, DisruptedConfig
        from pytsc_amd.scenario import build_tables

        # domain-randomised demand (reference pytsc/__init__.py:21-30): every file of the mode is a flow set of the tables,
        # each (re)started replica draws its own (domain, disruption value, file) -- DisruptedConfig.draw_sets
        self.disrupted = bool(kwargs.get("disrupted", False))
        self.config = DisruptedConfig(scenario, **kwargs) if self.disrupted else Config(scenario, **kwargs)
        sig = self.config.signal
        self.obs_kind = sig.get("observation_space", "position_matrix")
        if self.obs_kind not in ("position_matrix", "lane_features"):
            raise ValueError(f"observation_space {self.obs_kind!r}: expected position_matrix or lane_features")
        # reference common/observations.py:72-88: one np.random.normal per surviving bin of the position matrix.  The batched
        # path draws on the device, where kr_step assembles the rows (tsc_config_t::obs_noise_std): the same law -- N(0, std)
        # per surviving entry, clip to [0, 1], then the row's integer truncation -- from a counter-based generator instead
        # of numpy's global stream (which R replicas cannot follow at once; the single-env facade does follow it)
        self.obs_noise_std = float(sig.get("obs_noise_std", 0.0) or 0.0)
        if self.obs_noise_std and self.obs_kind != "position_matrix":
            raise ValueError("obs_noise_std applies to observation_space: position_matrix (reference "
                             "common/observations.py:72-88,140-160)")
        self.dropout_prob = float(sig.get("obs_dropout_prob", 0.0) or 0.0)
        if self.dropout_prob and self.obs_kind != "lane_features":
            raise ValueError("obs_dropout_prob applies to observation_space: lane_features (reference "
                             "common/observations.py:245-253)")
        if self.obs_kind == "lane_features" and "state" not in outputs:
            outputs = tuple(outputs) + ("state",)  # the lane-feature observation is assembled with the state rows
        self.tables = build_tables(self.config, n_replicas=n_replicas, set_offset=set_offset)
        self.parsed_network = self.tables.parser
        if self.obs_noise_std:
            self.tables.cfg.obs_noise_std = self.obs_noise_std
            self.tables.cfg.obs_noise_seed = int(self.config.simulator["seed"]) & 0xFFFFFFFFFFFFFFFF
        self.engine = VecEngine(self.tables, device=device, want=tuple(outputs))
        self.n_replicas = n_replicas
        self.n_agents = self.tables.n_agents
        self.agent_ids = list(self.parsed_network.traffic_signals.keys())
        self.buf = self.engine.observe()
        self.episode_count = 0
        self._set_rng = torch.Generator().manual_seed(int(self.config.simulator["seed"]))
        self._drop_rng = torch.Generator(device=self.engine.device).manual_seed(int(self.config.simulator["seed"]) + 1)
        self._keep = None  # [R, A, state_size] 0/1 multiplier of the lane-feature observation (lane dropout)
        self._dropped = None
        self.current_domain_class = None  # [R] int64 (disrupted configs): index into config.domain_classes
        if self.disrupted:
            sets, cls = self.config.draw_sets(n_replicas, self._set_rng)
            self.engine.reset(sets=sets)
            self.buf = self.engine.observe()
            self.current_domain_class = cls.to(self.engine.device)
            self.current_sets = sets.clone()  # [R] int32 (host): the flow set = file of config.simulator["flow_files"] each replica runs
        if self.obs_kind == "lane_features":
            self._lane_slot = self._lane_slots()
            self.reset_dropped_lanes()

    @property
    def episode_limit(self):
        return int(self.config.simulator["episode_limit"] / self.config.simulator["delta_time"])

    def get_observation_size(self):
        return self.engine.state_size if self.obs_kind == "lane_features" else self.engine.obs_size

    # -- observation_space: lane_features (reference common/observations.py:225-382) on the device ------------------
    def _lane_slots(self):
        """[A, 16] int64: network lane index of the k-th incoming lane of agent a (-1: padding)."""
        rn = self.tables.rn
        slots = torch.full((self.n_agents_static(), 16), -1, dtype=torch.int64)
        for a, ts_id in enumerate(rn.agent_ids):
            for k, lane in enumerate(self.parsed_network.traffic_signals[ts_id]["incoming_lanes"]):
                slots[a, k] = rn.lane_index[lane]
        return slots.to(self.engine.device)

    def n_agents_static(self):
        return self.tables.n_agents

    def reset_dropped_lanes(self, replicas=None):
        """Draw the dropped lanes anew (reference LaneFeatures.reset_dropped_lanes: int(p * n_lanes) lanes without
        replacement) -- per replica, from a counter-based device generator (the reference's numpy stream cannot be
        followed by R replicas at once; the single-env facade does follow it).  `replicas`: [R] bool, None = all."""
        if self.obs_kind != "lane_features":
            return
        R, nl = self.n_replicas_static(), self.tables.n_lanes
        n_drop = int(self.dropout_prob * nl)
        dropped = torch.zeros((R, nl + 1), dtype=torch.bool, device=self.engine.device)  # (+1: the padding slot)
        if n_drop:
            order = torch.rand((R, nl), device=self.engine.device, generator=self._drop_rng).argsort(dim=1)
            dropped[:, :nl].scatter_(1, order[:, :n_drop], True)
        if replicas is not None and self._dropped is not None:
            dropped = torch.where(replicas.to(torch.bool)[:, None], dropped, self._dropped)
        self._dropped = dropped
        lane_dropped = dropped[:, self._lane_slot]  # [R, A, 16] (slot -1 reads the padding column: never dropped)
        keep = torch.ones((R, self.tables.n_agents, self.engine.state_size), dtype=torch.float32, device=self.engine.device)
        body = keep[:, :, :16 * 12].view(R, self.tables.n_agents, 16, 12)
        body[:, :, :, 9:12] = (~lane_dropped).to(torch.float32)[..., None]
        self._keep = keep if n_drop else None

    def n_replicas_static(self):
        return self.tables.n_replicas

    def get_state_size(self):
        return self.engine.state_size

    def get_action_size(self):
        return self.engine.n_actions

    def get_observations(self):
        if self.obs_kind == "lane_features":
            # == the state rows with the three measured entries of dropped lanes zeroed (observations.py:305-329)
            st = self.buf["state"]
            return st if self._keep is None else st * self._keep
        return self.buf["obs"]

    def get_state(self):
        return self.buf["state"]

    def get_action_mask(self):
        return self.buf["mask"]

    def get_valid(self):
        """[R] uint8: 1 where the rows of the last step() / restart() are that call's; 0 for a replica that has run out of
        vehicle records on the resident family and waits for the migration (its rows are those of its last completed
        step).  The reference never returns a stale observation (pytsc/__init__.py:178-182): mask these replicas out."""
        return self.buf["valid"]

    def get_reward(self):
        return self.buf["global_reward"]

    def get_rewards(self):
        return self.buf["local_reward"]

    def get_env_info(self):
        """dict of [R] tensors, keys as the reference's get_step_stats (metrics.py:221-237)."""
        st = self.buf["stats"]
        info = {k: st[:, i] for i, k in enumerate(STAT_KEYS)}
        info["episode_limit"] = self.episode_limit
        if self.disrupted:  # (pytsc/__init__.py:153-155, per replica)
            info["n_domains"] = len(self.config.domain_classes)
            info["domain_class"] = self.current_domain_class
        return info

    def step(self, actions):
        """actions [R, A] int32 -> (global_reward [R], episode_over [R], info)."""
        self.buf = self.engine.step(actions)
        return self.buf["global_reward"], self.buf["done"], self.get_env_info()

    def restart(self):
        """Restart the replicas whose simulated hour is over (pytsc/__init__.py:164-176): the
        reference rebuilds the signals and measures again, so masks / observations / phases
        describe the fresh state afterwards.  `flow_rate_type: random` draws a flow set per
        restarted replica (backends/cityflow/config.py:63-76), here from a seeded torch generator."""
        sets = None
        if self.disrupted:
            sets, cls = self.config.draw_sets(self.n_replicas, self._set_rng)
            over = self.buf["terminated"].to(torch.bool)
            self.current_domain_class = torch.where(over, cls.to(over.device), self.current_domain_class)
            self.current_sets = torch.where(over.cpu(), sets, self.current_sets)
        elif self.config.simulator.get("flow_rate_type", "constant") == "random" and self.tables.flows.n_sets > 1:
            sets = torch.randint(0, int(self.tables.flows.n_sets), (self.n_replicas,), generator=self._set_rng,
                                 dtype=torch.int32)
        self.engine.reset(self.buf["terminated"], sets=sets)
        self.buf = self.engine.observe()

    def reset_all(self):
        """Every replica from an empty network.  A disrupted config draws anew for everybody (without the draw the
        replicas fell back to the handle's set rule while `current_sets` / `info["domain_class"]` kept the old draw)."""
        sets = None
        if self.disrupted:
            sets, cls = self.config.draw_sets(self.n_replicas, self._set_rng)
            self.current_domain_class = cls.to(self.engine.device)
            self.current_sets = sets.clone()
        self.engine.reset(sets=sets)
        self.buf = self.engine.observe()

    def random_valid_actions(self, generator=None):
        m = self.buf["mask"]
        u = torch.rand(m.shape, device=m.device, generator=generator).clamp_min_(1e-6)
        return (u * m).argmax(-1).to(torch.int32)

    def close(self):
        self.engine.close()
