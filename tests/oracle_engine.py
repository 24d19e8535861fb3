"""Test-only stand-in for pytsc_amd.vec_env.VecEngine backed by the CPU oracle,
so the host-side façade logic can be exercised without a GPU.  Lives under
tests/: the package never imports it."""
import numpy as np
import torch

from oracle.pyoracle import Oracle


class OracleEngine:
    device = "cpu"

    def __init__(self, tables, device=0):
        self.tables = tables
        self.o = Oracle(tables)
        self.R, self.A = tables.n_replicas, tables.n_agents
        self.n_actions = self.o.n_actions

    def _buf(self, out):
        names = ("obs", "state", "mask", "local_reward", "global_reward", "done", "terminated", "stats",
                 "agent_stats", "phase", "local_reward64", "global_reward64")
        b = {k: torch.from_numpy(getattr(out, k).copy()) for k in names}
        if getattr(self.tables, "phase_moves", None) is not None:
            b["phase_scores"] = torch.from_numpy(out.phase_scores.copy())
        return b

    def reset(self, mask=None, sets=None):
        self.o.reset(None if mask is None else np.asarray(mask), None if sets is None else np.asarray(sets))

    def observe(self):
        return self._buf(self.o.observe())

    def step(self, actions):
        return self._buf(self.o.step(np.asarray(actions, dtype=np.int32)))

    def lane_detail(self, with_posmat=False):
        c, s, p = self.o.lane_detail(with_posmat)
        return torch.from_numpy(c), torch.from_numpy(s), (torch.from_numpy(p) if p is not None else None)

    def lane_window_counts(self):
        return torch.from_numpy(self.o.lane_window_counts())

    def snapshot(self, rep=0, cap=65536):
        return self.o.snapshot(rep, cap)

    def errors(self):
        return self.o.errors()

    def close(self):
        self.o.close()


def oracle_engine_factory(tables, device):
    return OracleEngine(tables, device)
