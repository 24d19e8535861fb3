"""ctypes binding of the CPU oracle (oracle/libtsc_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, tools/gen_golden.py,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never from the
pytsc_amd package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libtsc_oracle.so")


def build(force=False):
    src = os.path.join(HERE, "tsc_oracle.c")
    hdr = os.path.join(HERE, "..", "include", "pytsc_amd.h")
    if (force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src)
            or os.path.getmtime(LIB) < os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", HERE, "-s", "-B", "libtsc_oracle.so"])
    return LIB


def load():
    from pytsc_amd.tables import TscConfig, TscFlows, TscNet, TscStepOut

    lib = C.CDLL(build())
    lib.tsco_create.restype = C.c_void_p
    lib.tsco_create.argtypes = [C.POINTER(TscNet), C.POINTER(TscFlows), C.POINTER(TscConfig)]
    lib.tsco_destroy.argtypes = [C.c_void_p]
    lib.tsco_errors.argtypes = [C.c_void_p]
    lib.tsco_reset.argtypes = [C.c_void_p, C.c_void_p]
    lib.tsco_reset_sets.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.tsco_observe.argtypes = [C.c_void_p, C.POINTER(TscStepOut)]
    lib.tsco_step.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(TscStepOut)]
    lib.tsco_lane_detail.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    lib.tsco_snapshot.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]
    lib.tsco_lane_window_counts.argtypes = [C.c_void_p, C.c_void_p]
    lib.tsco_n_actions.argtypes = [C.c_void_p]
    lib.tsco_max_bins.argtypes = [C.c_void_p]
    lib.tsco_engine_next_step.argtypes = [C.c_void_p, C.c_int]
    lib.tsco_engine_set_tl_phase.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.tsco_engine_current_time.argtypes = [C.c_void_p, C.c_int]
    lib.tsco_engine_current_time.restype = C.c_double
    lib.tsco_engine_vehicle_count.argtypes = [C.c_void_p, C.c_int]
    lib.tsco_engine_average_travel_time.argtypes = [C.c_void_p, C.c_int]
    lib.tsco_engine_average_travel_time.restype = C.c_double
    lib.tsco_engine_lane_vehicles.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                              C.c_void_p, C.c_void_p]
    lib.tsco_engine_drv_count.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.tsco_engine_finished.argtypes = [C.c_void_p, C.c_int]
    lib.tsco_engine_created.argtypes = [C.c_void_p, C.c_int]
    return lib


class StepBuffers:
    """Host (numpy) output buffers laid out like the C-ABI's tsc_step_out_t."""

    def __init__(self, tables, n_actions):
        from pytsc_amd.tables import N_AGENT_STATS, N_STATS, TscStepOut

        R, A = tables.n_replicas, tables.n_agents
        self.obs = np.zeros((R, A, tables.obs_size), np.float32)
        self.state = np.zeros((R, A, tables.state_size), np.float32)
        self.mask = np.zeros((R, A, n_actions), np.uint8)
        self.local_reward = np.zeros((R, A), np.float32)
        self.global_reward = np.zeros((R,), np.float32)
        self.done = np.zeros((R,), np.uint8)
        self.terminated = np.zeros((R,), np.uint8)
        self.stats = np.zeros((R, N_STATS), np.float64)
        self.agent_stats = np.zeros((R, A, N_AGENT_STATS), np.float64)
        self.phase = np.zeros((R, A, 3), np.int32)
        self.local_reward64 = np.zeros((R, A), np.float64)
        self.global_reward64 = np.zeros((R,), np.float64)
        self.phase_scores = np.zeros((R, A, 20, 3), np.int16)  # [R, A, TSC_MAX_PHASES, 3]
        p = lambda a: a.ctypes.data
        self.struct = TscStepOut(
            p(self.obs), p(self.state), p(self.mask), p(self.local_reward), p(self.global_reward),
            p(self.done), p(self.terminated), p(self.stats), p(self.agent_stats), p(self.phase),
            p(self.local_reward64), p(self.global_reward64), None,
            p(self.phase_scores) if getattr(tables, "phase_moves", None) is not None else None,
        )


class Oracle:
    def __init__(self, tables):
        self.lib = load()
        self.tables = tables
        self.h = self.lib.tsco_create(C.byref(tables.net), C.byref(tables.flows), C.byref(tables.cfg))
        self.n_actions = self.lib.tsco_n_actions(self.h)
        self.max_bins = self.lib.tsco_max_bins(self.h)
        self.out = StepBuffers(tables, self.n_actions)

    def close(self):
        if self.h:
            self.lib.tsco_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def errors(self):
        return self.lib.tsco_errors(self.h)

    def reset(self, mask=None, sets=None):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        s = None if sets is None else np.ascontiguousarray(sets, np.int32)
        self.lib.tsco_reset_sets(self.h, None if m is None else m.ctypes.data,
                                 None if s is None else s.ctypes.data)

    def observe(self):
        self.lib.tsco_observe(self.h, C.byref(self.out.struct))
        return self.out

    def step(self, actions):
        a = np.ascontiguousarray(actions, np.int32)
        rc = self.lib.tsco_step(self.h, a.ctypes.data, C.byref(self.out.struct))
        if rc != 0:
            raise ValueError(f"tsco_step failed: {rc}")
        return self.out

    def lane_detail(self, with_posmat=True):
        R, nl = self.tables.n_replicas, self.tables.n_lanes
        counts = np.zeros((R, nl, 2), np.int32)
        speed = np.zeros((R, nl), np.float64)
        pm = np.zeros((R, nl, self.max_bins), np.float64) if with_posmat else None
        self.lib.tsco_lane_detail(self.h, counts.ctypes.data, speed.ctypes.data,
                                  pm.ctypes.data if with_posmat else None, self.max_bins)
        return counts, speed, pm

    def lane_window_counts(self):
        out = np.zeros((self.tables.n_replicas, self.tables.n_lanes, 3), np.int32)
        self.lib.tsco_lane_window_counts(self.h, out.ctypes.data)
        return out

    def snapshot(self, rep=0, cap=65536):
        drv = np.zeros(cap, np.int32)
        dist = np.zeros(cap, np.float64)
        speed = np.zeros(cap, np.float64)
        vid = np.zeros(cap, np.int32)
        n = self.lib.tsco_snapshot(self.h, rep, cap, drv.ctypes.data, dist.ctypes.data,
                                   speed.ctypes.data, vid.ctypes.data)
        return drv[:n], dist[:n], speed[:n], vid[:n]

    # engine-level
    def next_step(self, rep=0):
        self.lib.tsco_engine_next_step(self.h, rep)

    def set_tl_phase(self, rep, agent, raw):
        self.lib.tsco_engine_set_tl_phase(self.h, rep, agent, raw)

    def lane_vehicles(self, rep, lane, cap=256):
        vids = np.zeros(cap, np.int32)
        dist = np.zeros(cap, np.float64)
        speed = np.zeros(cap, np.float64)
        n = self.lib.tsco_engine_lane_vehicles(self.h, rep, lane, cap, vids.ctypes.data,
                                               dist.ctypes.data, speed.ctypes.data)
        return vids[:n], dist[:n], speed[:n]
