"""Rule-based controllers vectorised over [R, A] on device tensors.

Same decision rules as the reference's per-signal controllers
(pytsc/controllers/controllers.py:26-259), evaluated for every replica and signal
at once from the engine's phase state, its action masks and the per-lane window
counts (tsc_lane_window_counts).  All return target phase indices [R, A] int32
(the phase_selection action space, as controllers/evaluate.py:26,57-58 forces).
"""
import torch


class VecControllerContext:
    """Static per-network tensors shared by the controllers."""

    def __init__(self, engine):
        self.engine = engine
        t = engine.tables
        p = t.parser
        dev = engine.device
        self.A = t.n_agents
        self.P = t.max_n_phases
        ts_cfgs = [p.traffic_signals[a] for a in t.rn.agent_ids]
        n_ph = torch.tensor([c["n_phases"] for c in ts_cfgs], device=dev)
        self.n_phases = n_ph
        green = torch.zeros((self.A, self.P), dtype=torch.bool, device=dev)
        mn = torch.zeros((self.A, self.P), dtype=torch.int32, device=dev)
        mx = torch.zeros((self.A, self.P), dtype=torch.int32, device=dev)
        nl = t.n_lanes
        inc = torch.zeros((self.A, self.P, nl), dtype=torch.float32, device=dev)
        # MaxPressure quirk (controllers.py:166-176): for an incoming lane only the LAST
        # outgoing lane of the phase is counted
        last_out = torch.zeros((self.A, self.P, nl), dtype=torch.long, device=dev)
        for a, c in enumerate(ts_cfgs):
            for k, raw in enumerate(c["phases"]):
                green[a, k] = k in c["green_phase_indices"]
                mn[a, k] = c["phases_min_max_times"][raw]["min_time"]
                mx[a, k] = c["phases_min_max_times"][raw]["max_time"]
                for lane, outs in c["phase_to_inc_out_lanes"][raw].items():
                    li = t.rn.lane_index[lane]
                    inc[a, k, li] = 1.0
                    last_out[a, k, li] = t.rn.lane_index[outs[-1]]
        self.green, self.min_time, self.max_time = green, mn, mx
        self.phase_inc, self.phase_last_out = inc, last_out
        self.round_robin = bool(t.config.signal["round_robin"])

    def phase_state(self, buf):
        ph = buf["phase"]
        return ph[..., 0].long(), ph[..., 1]

    def allowable(self, buf):
        """[R, A, P] bool: TLSFree/RoundRobinPhaseSelectLogic (common/traffic_signal.py:329-404)."""
        cur, top = self.phase_state(buf)
        R = cur.shape[0]
        idx = torch.arange(self.P, device=cur.device).view(1, 1, -1)
        nxt = (cur + 1) % self.n_phases.view(1, -1)
        is_green = self.green.unsqueeze(0).expand(R, -1, -1).gather(2, cur.unsqueeze(-1)).squeeze(-1)
        mn = self.min_time.unsqueeze(0).expand(R, -1, -1).gather(2, cur.unsqueeze(-1)).squeeze(-1)
        mx = self.max_time.unsqueeze(0).expand(R, -1, -1).gather(2, cur.unsqueeze(-1)).squeeze(-1)
        is_cur = idx == cur.unsqueeze(-1)
        is_nxt = idx == nxt.unsqueeze(-1)
        g = (is_cur & (top < mx).unsqueeze(-1)) | (is_nxt & (top >= mn).unsqueeze(-1) & (top <= mx).unsqueeze(-1))
        if self.round_robin:
            y = is_nxt
        else:
            y = self.green.unsqueeze(0) & (idx != (cur - 1).unsqueeze(-1))
        valid = idx < self.n_phases.view(1, -1, 1)
        return torch.where(is_green.unsqueeze(-1), g, y) & valid


def fixed_time(ctx, buf, green_time=25):
    """FixedTimeController (controllers.py:39-54)."""
    cur, top = ctx.phase_state(buf)
    is_green = ctx.green.unsqueeze(0).expand(cur.shape[0], -1, -1).gather(2, cur.unsqueeze(-1)).squeeze(-1)
    nxt = (cur + 1) % ctx.n_phases.view(1, -1)
    return torch.where(is_green & (top < green_time), cur, nxt).to(torch.int32)


def _argmax_random_ties(score, allow, generator=None):
    s = torch.where(allow, score, torch.full_like(score, float("-inf")))
    best = s.max(-1, keepdim=True).values
    tie = (s == best) & allow
    u = torch.rand(s.shape, device=s.device, generator=generator)
    return (u * tie).argmax(-1)


# The three per-phase scores.  `counts`: either the step's `phase_scores` output [R, A, 20, 3] int16 (round 5: the step
# computes them where it assembles the observation rows -- tsc_step_out_t::phase_scores_dev; nothing larger than [R, A, P]
# exists on this side) or the per-lane window counts [R, L, 3] of tsc_lane_window_counts (the dense formulas of rounds 3-4:
# a second pass over the state and, for the pressure, an [R, A, P, L] tensor -- 1 GB per call on the 4x4 @ 4096, out of
# memory on the 16x16; kept as the independent restatement the fused scores are tested against at small R).
def _fused(ctx, counts, col):
    return counts[:, :, : ctx.P, col].float() if counts.dim() == 4 else None


def queue_scores(ctx, counts):
    """[R, A, P]: stopped vehicles (bins == 0.0) in the incoming windows of each phase's lanes
    (GreedyController._compute_queue_for_phase, controllers.py:93-112)."""
    f = _fused(ctx, counts, 0)
    return f if f is not None else torch.einsum("apl,rl->rap", ctx.phase_inc, counts[..., 0].float())


def pressure_scores(ctx, counts):
    """[R, A, P]: MaxPressureController._compute_pressure_for_phase (controllers.py:153-178), including its
    quirk that only the LAST outgoing lane of a movement is counted."""
    f = _fused(ctx, counts, 2)
    if f is not None:
        return f
    occ_in = counts[..., 1].float()   # [R, L]
    occ_out = counts[..., 2].float()  # [R, L]
    R = occ_in.shape[0]
    out_sel = occ_out[:, ctx.phase_last_out]  # [R, A, P, L]
    return ((occ_in.view(R, 1, 1, -1) - out_sel).abs() * ctx.phase_inc.unsqueeze(0)).sum(-1)


def flow_scores(ctx, counts):
    """[R, A, P]: vehicles (bins >= 0) in the incoming windows of each phase's lanes
    (SOTLController._compute_flow_for_phase, controllers.py:227-242)."""
    f = _fused(ctx, counts, 1)
    return f if f is not None else torch.einsum("apl,rl->rap", ctx.phase_inc, counts[..., 1].float())


def greedy(ctx, buf, counts, generator=None):
    """GreedyController (controllers.py:69-112): most stopped vehicles in the window."""
    cur, _ = ctx.phase_state(buf)
    score = queue_scores(ctx, counts)
    allow = ctx.allowable(buf)
    pick = _argmax_random_ties(score, allow, generator)
    is_green = ctx.green.unsqueeze(0).expand(cur.shape[0], -1, -1).gather(2, cur.unsqueeze(-1)).squeeze(-1)
    nxt = (cur + 1) % ctx.n_phases.view(1, -1)
    return torch.where(is_green, pick, nxt).to(torch.int32)


def max_pressure(ctx, buf, counts, generator=None):
    """MaxPressureController (controllers.py:127-178)."""
    cur, _ = ctx.phase_state(buf)
    R = cur.shape[0]
    score = pressure_scores(ctx, counts)
    allow = ctx.allowable(buf)
    pick = _argmax_random_ties(score, allow, generator)
    is_green = ctx.green.unsqueeze(0).expand(R, -1, -1).gather(2, cur.unsqueeze(-1)).squeeze(-1)
    nxt = (cur + 1) % ctx.n_phases.view(1, -1)
    return torch.where(is_green, pick, nxt).to(torch.int32)


def sotl(ctx, buf, counts, theta=3, mu=4, phi_min=5):
    """SOTLController (controllers.py:200-242)."""
    cur, top = ctx.phase_state(buf)
    R = cur.shape[0]
    flow = flow_scores(ctx, counts)
    nxt = (cur + 1) % ctx.n_phases.view(1, -1)
    nxt_green = (cur + 2) % ctx.n_phases.view(1, -1)
    allow = ctx.allowable(buf)
    can_stay = allow.gather(2, cur.unsqueeze(-1)).squeeze(-1)
    red = flow.gather(2, nxt_green.unsqueeze(-1)).squeeze(-1)
    green = flow.gather(2, cur.unsqueeze(-1)).squeeze(-1)
    switch = (top >= phi_min) & ~((green > 0) & (green < mu)) & (red >= theta)
    return torch.where(can_stay, torch.where(switch, nxt, cur), nxt).to(torch.int32)


def random_valid(ctx, buf, generator=None):
    """RandomController (controllers.py:245-268) on tensors; `VecEngine.random_valid_actions`
    is the same policy fused into one launch of the library."""
    allow = ctx.allowable(buf)
    u = torch.rand(allow.shape, device=allow.device, generator=generator).clamp_min(1e-6)
    return (u * allow).argmax(-1).to(torch.int32)
