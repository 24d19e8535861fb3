"""Flat tables handed across the C-ABI (include/pytsc_amd.h: tsc_net_t,
tsc_flows_t, tsc_config_t).  Field order must match the header exactly."""
import ctypes as C
import math

import numpy as np

TMPL_STRIDE = 12
MAX_AGENT_PHASES = 32
MAX_CTRL_LANES = 16
MAX_PHASES = 20
LANE_STATIC = 9
N_STATS = 8
N_AGENT_STATS = 8

_pd = C.POINTER(C.c_double)
_pi = C.POINTER(C.c_int32)
_pu = C.POINTER(C.c_uint32)
_pl = C.POINTER(C.c_int64)


class TscNet(C.Structure):
    _fields_ = [
        ("n_lanes", C.c_int32), ("n_links", C.c_int32), ("n_agents", C.c_int32),
        ("n_routes", C.c_int32), ("n_tmpl", C.c_int32), ("n_raw_phases_max", C.c_int32),
        ("drv_length", _pd), ("drv_max_speed", _pd), ("drv_cap", _pi),
        ("lane_out_ptr", _pi), ("lane_out_links", _pi),
        ("lane_in_ptr", _pi), ("lane_in_links", _pi),
        ("lane_len_pytsc", _pd), ("lane_bins", _pi), ("lane_static", _pi),
        ("link_start_lane", _pi), ("link_end_lane", _pi), ("link_type", _pi),
        ("link_agent", _pi), ("link_avail_mask", _pu),
        ("link_cross_ptr", _pi), ("cross_other", _pi),
        ("cross_dist_self", _pd), ("cross_dist_other", _pd),
        ("agent_n_phases", _pi), ("agent_phase_raw", _pi), ("agent_phase_green", _pi),
        ("agent_phase_min", _pi), ("agent_phase_max", _pi),
        ("agent_inc_ptr", _pi), ("agent_inc_lanes", _pi),
        ("agent_out_ptr", _pi), ("agent_out_lanes", _pi),
        ("agent_nbr_ptr", _pi), ("agent_nbr", _pi), ("agent_nbr_w", _pd),
        ("route_ptr", _pi), ("route_drv", _pi),
        ("tmpl", _pd), ("agent_phase_moves", _pu),
    ]


class TscFlows(C.Structure):
    _fields_ = [
        ("n_sets", C.c_int32), ("sim_len", C.c_int32),
        ("set_veh_ptr", _pi), ("veh_route", _pi), ("veh_start", _pi), ("veh_tmpl", _pi),
        ("spawn_ptr", _pi), ("spawn_vid", _pi),
        ("created_cnt", _pi), ("created_tsum", _pl), ("veh_priority", _pi),
    ]


class TscConfig(C.Structure):
    _fields_ = [
        ("n_replicas", C.c_int32), ("delta_time", C.c_int32), ("yellow_time", C.c_int32),
        ("action_space", C.c_int32), ("round_robin", C.c_int32), ("visibility", C.c_int32),
        ("sim_length", C.c_int32), ("episode_limit", C.c_int32),
        ("initial_wait_time", C.c_int32), ("set_offset", C.c_int32),
        ("sequential_sets", C.c_int32), ("reward_kind", C.c_int32),
        ("interval", C.c_double), ("veh_size_min_gap", C.c_double),
        ("flickering_coef", C.c_double),
        ("kernel_family", C.c_int32), ("resident_capacity", C.c_int32),
        ("resident_threads", C.c_int32), ("resident_force", C.c_int32),
        ("resident_task_capacity", C.c_int32), ("resident_margin", C.c_int32),
        ("resident_spill_limit", C.c_int32), ("replica_base", C.c_int32),
        ("obs_noise_std", C.c_double), ("obs_noise_seed", C.c_uint64),
    ]


class TscStepOut(C.Structure):
    _fields_ = [
        ("obs_dev", C.c_void_p), ("state_dev", C.c_void_p), ("mask_dev", C.c_void_p),
        ("local_reward_dev", C.c_void_p), ("global_reward_dev", C.c_void_p),
        ("done_dev", C.c_void_p), ("terminated_dev", C.c_void_p),
        ("stats_dev", C.c_void_p), ("agent_stats_dev", C.c_void_p),
        ("phase_dev", C.c_void_p), ("local_reward64_dev", C.c_void_p),
        ("global_reward64_dev", C.c_void_p), ("valid_dev", C.c_void_p),
        ("phase_scores_dev", C.c_void_p),
    ]


def _csr(lists, dtype=np.int32):
    ptr = np.zeros(len(lists) + 1, dtype=np.int32)
    for i, l in enumerate(lists):
        ptr[i + 1] = ptr[i] + len(l)
    flat = np.asarray([x for l in lists for x in l], dtype=dtype)
    if flat.size == 0:
        flat = np.zeros(1, dtype=dtype)
    return ptr, flat


def lane_static_features(parser, lane):
    """Static lane features (reference: common/observations.py:90-117)."""
    one_hot = [0.0] * 6
    one_hot[parser.lane_indices[lane]] = 1.0
    length = float(np.clip(parser.lane_lengths[lane] / 500, 0, 1))
    angle = float(np.clip(parser.lane_angles[lane] / np.pi, -1, 1))
    speed = float(np.clip(parser.lane_max_speeds[lane] / 15.0, 0, 1))
    return [length, angle, speed] + one_hot


class Tables:
    """Owns the numpy arrays and the ctypes structs pointing into them."""

    def __init__(self, parser, flow_compiler, config, n_replicas, set_offset=0):
        rn = parser.rn
        if getattr(rn, "inconsistent", None):
            # (the C-ABI would answer TSC_EINVAL; said here, with the reason, before any table is built)
            raise ValueError("inconsistent road network (TSC_EINVAL): " + "; ".join(rn.inconsistent[:3]) +
                             (f" ... {len(rn.inconsistent)} road links in all" if len(rn.inconsistent) > 3 else ""))
        self.rn = rn
        self.parser = parser
        self.fc = flow_compiler
        self.config = config
        self._keep = []
        sim = config.simulator
        sig = config.signal
        interval = float(sim["interval"])
        self.n_lanes, self.n_links, self.n_agents = rn.n_lanes, rn.n_links, rn.n_agents
        nl, nk = rn.n_lanes, rn.n_links

        min_len = min([t[0] for t in flow_compiler.tmpls] or [5.0])
        drv_length = np.concatenate([rn.lane_length, [l["length"] for l in rn.links]]).astype(np.float64)
        drv_max_speed = np.concatenate([rn.lane_max_speed, np.full(nk, 10000.0)]).astype(np.float64)
        drv_cap = np.asarray([int(math.floor(L / min_len)) + 2 for L in drv_length], dtype=np.int32)
        lane_out_ptr, lane_out = _csr([[nl + k for k in ks] for ks in rn.lane_out_links])
        lane_in_ptr, lane_in = _csr([[nl + k for k in ks] for ks in rn.lane_in_links])
        v_size = float(sim["veh_size_min_gap"])
        lane_bins = np.asarray([int(L / v_size) for L in rn.lane_len_pytsc], dtype=np.int32)
        self.lane_features = {l: lane_static_features(parser, l) for l in rn.lane_ids}
        lane_static = np.asarray(
            [[int(x) for x in self.lane_features[l]] for l in rn.lane_ids], dtype=np.int32
        ).reshape(-1)

        link_start = np.asarray([l["start"] for l in rn.links], dtype=np.int32)
        link_end = np.asarray([l["end"] for l in rn.links], dtype=np.int32)
        link_type = np.asarray([l["type"] for l in rn.links], dtype=np.int32)
        link_agent = np.asarray([l["agent"] for l in rn.links], dtype=np.int32)
        link_mask = np.asarray([l["mask"] for l in rn.links], dtype=np.uint32)
        cross_ptr, cross_other = _csr([[nl + c[2] for c in cs] for cs in rn.link_crosses])
        _, cross_ds = _csr([[c[0] for c in cs] for cs in rn.link_crosses], np.float64)
        _, cross_do = _csr([[c[3] for c in cs] for cs in rn.link_crosses], np.float64)

        A = rn.n_agents
        n_phases = np.zeros(A, dtype=np.int32)
        ph_raw = np.zeros((A, MAX_AGENT_PHASES), dtype=np.int32)
        ph_green = np.zeros((A, MAX_AGENT_PHASES), dtype=np.int32)
        ph_min = np.zeros((A, MAX_AGENT_PHASES), dtype=np.int32)
        ph_max = np.zeros((A, MAX_AGENT_PHASES), dtype=np.int32)
        inc_l, out_l, nbr_l, nbr_w = [], [], [], []
        gamma = config.misc["reward_gamma"]
        for a, ts_id in enumerate(rn.agent_ids):
            ts = parser.traffic_signals[ts_id]
            n = ts["n_phases"]
            if n > MAX_AGENT_PHASES:
                raise ValueError("too many phases for one signal")
            n_phases[a] = n
            for k, raw in enumerate(ts["phases"]):
                ph_raw[a, k] = raw
                ph_green[a, k] = 1 if k in ts["green_phase_indices"] else 0
                ph_min[a, k] = ts["phases_min_max_times"][raw]["min_time"]
                ph_max[a, k] = ts["phases_min_max_times"][raw]["max_time"]
            inc_l.append([rn.lane_index[l] for l in ts["incoming_lanes"]])
            out_l.append([rn.lane_index[l] for l in ts["outgoing_lanes"]])
            nb, w = [], []
            # common/reward.py:83-87: k ascending, neighbours in parser order
            for k in sorted(parser.k_hop_neighbors[ts_id].keys()):
                if not (1 <= k <= len(rn.agent_ids) - 1):
                    continue  # the reference loops k over range(1, n_agents)
                for nid in parser.k_hop_neighbors[ts_id][k]:
                    nb.append(rn.agent_index[nid])
                    w.append(gamma ** k)
            nbr_l.append(nb)
            nbr_w.append(w)
        # the movements of every phase (include/pytsc_amd.h, tsc_net_t::agent_phase_moves): which of the signal's incoming
        # lanes a phase serves and, per such lane, the LAST outgoing lane its dict entry lists (controllers.py:153-178)
        moves = np.zeros((A, MAX_AGENT_PHASES, 4), dtype=np.uint32)
        moves_ok = all(len(l) <= 16 for l in inc_l) and all(len(l) <= 16 for l in out_l)
        if moves_ok:
            for a, ts_id in enumerate(rn.agent_ids):
                ts = parser.traffic_signals[ts_id]
                inc_pos = {l: k for k, l in enumerate(ts["incoming_lanes"])}
                out_pos = {l: k for k, l in enumerate(ts["outgoing_lanes"])}
                for k, raw in enumerate(ts["phases"]):
                    sel = 0
                    for lane, outs in ts["phase_to_inc_out_lanes"].get(raw, {}).items():
                        i = inc_pos[lane]
                        moves[a, k, 0] |= np.uint32(1 << i)
                        sel |= (out_pos[outs[-1]] & 15) << (4 * i)
                    moves[a, k, 1] = np.uint32(sel & 0xFFFFFFFF)
                    moves[a, k, 2] = np.uint32(sel >> 32)
        self.phase_moves = moves if moves_ok else None
        inc_ptr, inc_flat = _csr(inc_l)
        out_ptr, out_flat = _csr(out_l)
        nbr_ptr, nbr_flat = _csr(nbr_l)
        _, nbr_wflat = _csr(nbr_w, np.float64)
        if max(len(l) for l in inc_l) > MAX_CTRL_LANES:
            raise ValueError("more than 16 incoming lanes at one signal")

        route_ptr, route_drv = _csr(flow_compiler.routes)
        tmpl = np.asarray(flow_compiler.tmpls, dtype=np.float64).reshape(-1)
        if tmpl.size == 0:
            tmpl = np.zeros(TMPL_STRIDE)
        self.n_raw_phases_max = max(rn.agent_n_raw_phases) if rn.agent_n_raw_phases else 0
        if self.n_raw_phases_max > 32:
            raise ValueError("more than 32 raw light phases")

        a = self._a
        self.net = TscNet(
            nl, nk, A, len(flow_compiler.routes), max(len(flow_compiler.tmpls), 1),
            self.n_raw_phases_max,
            a(drv_length, _pd), a(drv_max_speed, _pd), a(drv_cap, _pi),
            a(lane_out_ptr, _pi), a(lane_out, _pi), a(lane_in_ptr, _pi), a(lane_in, _pi),
            a(rn.lane_len_pytsc.astype(np.float64), _pd), a(lane_bins, _pi), a(lane_static, _pi),
            a(link_start, _pi), a(link_end, _pi), a(link_type, _pi), a(link_agent, _pi),
            a(link_mask, _pu), a(cross_ptr, _pi), a(cross_other, _pi),
            a(cross_ds, _pd), a(cross_do, _pd),
            a(n_phases, _pi), a(ph_raw.reshape(-1), _pi), a(ph_green.reshape(-1), _pi),
            a(ph_min.reshape(-1), _pi), a(ph_max.reshape(-1), _pi),
            a(inc_ptr, _pi), a(inc_flat, _pi), a(out_ptr, _pi), a(out_flat, _pi),
            a(nbr_ptr, _pi), a(nbr_flat, _pi), a(nbr_wflat, _pd),
            a(route_ptr, _pi), a(route_drv, _pi), a(tmpl, _pd),
            a(moves.reshape(-1), _pu) if moves_ok else None,
        )
        self.drv_cap = drv_cap
        self.drv_length = drv_length
        self.lane_bins = lane_bins
        self.max_bins = int(max(lane_bins.max(), sig["visibility"]))

        # ---- flows -----------------------------------------------------------
        sim_len = int(sim["sim_length"]) + int(sim["initial_wait_time"])
        sets = flow_compiler.sets
        set_ptr = np.zeros(len(sets) + 1, dtype=np.int32)
        for i, s in enumerate(sets):
            set_ptr[i + 1] = set_ptr[i] + s.n_veh
        cat = lambda xs: (np.concatenate(xs) if sum(len(x) for x in xs) else np.zeros(1, np.int32)).astype(np.int32)
        veh_route = cat([s.veh_route for s in sets])
        veh_start = cat([s.veh_start for s in sets])
        veh_tmpl = cat([s.veh_tmpl for s in sets])
        # per set (numpy; a set object that occurs several times -- many replicas' worth of tiled demands -- is
        # processed once): the vehicles of each lane's spawn queue in creation order, the creation prefix tables
        route_first = np.asarray([r[0] for r in flow_compiler.routes] or [0], dtype=np.int64)
        per_set = {}
        spawn_cnt = np.zeros((len(sets), nl), dtype=np.int64)
        spawn_vids = []
        created_cnt = np.zeros((len(sets), sim_len + 2), dtype=np.int32)
        created_tsum = np.zeros((len(sets), sim_len + 2), dtype=np.int64)
        for si, s in enumerate(sets):
            got = per_set.get(id(s))
            if got is None:
                first = route_first[s.veh_route] if s.n_veh else np.zeros(0, np.int64)
                order = np.argsort(first, kind="stable").astype(np.int32)  # lane-major, creation order inside a lane
                counts = np.bincount(first, minlength=nl).astype(np.int64)
                starts = s.veh_start.astype(np.int64)  # ascending (creation order)
                csum = np.concatenate([[0], np.cumsum(starts)])
                idx = np.searchsorted(starts, np.arange(sim_len + 2), side="left")
                got = per_set[id(s)] = (order, counts, idx.astype(np.int32), csum[idx])
            spawn_vids.append(got[0])
            spawn_cnt[si] = got[1]
            created_cnt[si] = got[2]
            created_tsum[si] = got[3]
        spawn_ptr = np.zeros(len(sets) * nl + 1, dtype=np.int32)
        spawn_ptr[1:] = np.cumsum(spawn_cnt.reshape(-1))
        spawn_vid = cat(spawn_vids)
        # the last tie-break of Cross::canPass (include/pytsc_amd.h, tsc_flows_t::veh_priority): lower vehicle id (default) or a
        # restatement of the engine's mt19937 priority stream: every set is one engine run seeded with the config seed
        rule = str(sim.get("priority_rule", "vid"))
        if rule not in ("vid", "mt19937"):
            raise ValueError(f"priority_rule {rule!r}: expected vid or mt19937")
        prio = None
        if rule == "mt19937":
            from pytsc_amd.flows import mt19937_priorities

            prio = a(cat([mt19937_priorities(int(sim["seed"]), s.n_veh) for s in sets]), _pi)
        self.flows = TscFlows(
            len(sets), sim_len, a(set_ptr, _pi), a(veh_route, _pi), a(veh_start, _pi),
            a(veh_tmpl, _pi), a(spawn_ptr, _pi), a(spawn_vid, _pi),
            a(created_cnt.reshape(-1), _pi), a(created_tsum.reshape(-1), _pl), prio,
        )
        self.max_veh = int(max([s.n_veh for s in sets] + [1]))

        act = {"phase_selection": 0, "phase_switch": 1}[sig["action_space"]]
        rk = {"queue_length": 0, "max_pressure": 1}[sig["reward_function"]]
        self.cfg = TscConfig(
            int(n_replicas), int(sim["delta_time"]), int(sig["yellow_time"]), act,
            1 if sig["round_robin"] else 0, int(sig["visibility"]),
            int(sim["sim_length"]), int(sim["episode_limit"]),
            int(sim["initial_wait_time"]), int(set_offset),
            1 if sim.get("flow_rate_type", "constant") == "sequential" else 0, rk,
            interval, v_size, float(config.misc["flickering_coef"]),
            {"auto": 0, "list": 1, "resident": 2}[str(sim.get("kernel_family", "auto"))],
            int(sim.get("resident_capacity", 0)), int(sim.get("resident_threads", 0)),
            int(sim.get("resident_force", 0)), int(sim.get("resident_task_capacity", 0)),
            int(sim.get("resident_margin", 0)),
            int(sim.get("resident_spill_limit", 0)), int(sim.get("replica_base", 0)),
            0.0, 0,  # device-side observation noise: off (the batched environment turns it on, vec_env.py)
        )
        self.n_replicas = int(n_replicas)
        self.obs_size = MAX_CTRL_LANES * (int(sig["visibility"]) + LANE_STATIC) + MAX_PHASES
        self.state_size = MAX_CTRL_LANES * (3 + LANE_STATIC) + MAX_PHASES
        self.max_n_phases = int(n_phases.max()) if A else 0
        self.n_actions = 2 if act == 1 else self.max_n_phases

    def _a(self, arr, ptype):
        arr = np.ascontiguousarray(arr)
        self._keep.append(arr)
        return arr.ctypes.data_as(ptype)
