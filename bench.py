"""Benchmark of the step() hot path: agent-env-steps/s on the 4x4 grid.

  python bench.py --gpus N --steps K --warmup W

A "step" is one env step of every replica: apply actions -> 5 engine sub-steps ->
measure -> observations / rewards / masks / info stats (tsc_step through the C-ABI),
driven by an on-device random-valid-action policy.  Inputs are synthetic (Poisson
demand) and resident in HBM before the timed region.

* N > 1 and no WORLD_SIZE in the environment: this process -- before any GPU call --
  starts N worker processes of itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set),
  waits for them and fails loudly if one fails or the node has fewer than N GPUs.
  Under `python -m torch.distributed.run` it is a worker straight away.
* Replicas shard across ranks with no collective on the step path (weak scaling:
  --replicas per GPU); torch.distributed (RCCL) is used only for the barrier around
  the timed region and the max-over-ranks of the elapsed time.
* Every run first rolls all replicas forward by a fixed 150 env steps (the network is then
  filled) OUTSIDE the timed region, whatever --warmup says; then W untimed warm-up steps,
  then exactly K timed steps.
* Rank 0 prints ONE JSON line: throughput, the roofline object of the dominant kernel
  (timed live with HIP events on its stream) and, at N = 1, the CPU baseline (the scalar
  oracle -- a restatement, not CityFlow -- on 1 host core and on all host cores,
  measured before the GPU is touched).
"""
import argparse
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
METRIC = "agent-env-steps/sec, 4x4 grid @4096 replicas, at 1/2/4/8 MI355X"
DEFAULT_OUTPUTS = "obs,mask,local_reward,global_reward,done,terminated,stats,valid"
MAX_SEQUENTIAL_DEMANDS = 64  # up to here: one poisson_flows() call per set (the sets of rounds 1-3); beyond: the array generator
MAX_TILED_DEMANDS = 64


def workload_tables(rows, cols, n_replicas, set_offset, rate, n_sets, scenario="grid"):
    from pytsc_amd.config import Config
    from pytsc_amd.scenario import build_tables

    if scenario == "jinan":
        # BASELINE config 4: the Jinan 3x4 network with its real flow file (irregular lanes / phases), as recorded in the
        # golden fixture tests/golden/jinan_3_4_random (road network + flows are data of the reference's scenario)
        sys.path.insert(0, os.path.join(REPO, "tests"))
        from fixtures import load_case

        return load_case("jinan_3_4_random", n_replicas=n_replicas)[4]

    hip = {"flow_rate": rate, "n_flow_sets": n_sets, "flow_seed": 0}
    if os.environ.get("PYTSC_AMD_RATE_MIX"):  # diagnostic: "700:0.97,1500:0.03" -- a few replicas under a heavier demand
        hip["flow_rate_mix"] = [[float(a), float(b)] for a, b in (t.split(":") for t in os.environ["PYTSC_AMD_RATE_MIX"].split(","))]
    if os.environ.get("PYTSC_AMD_CAPACITY"):  # diagnostic: record capacity per replica of the resident family
        hip["resident_capacity"] = int(os.environ["PYTSC_AMD_CAPACITY"])
    if os.environ.get("PYTSC_AMD_SPILL_LIMIT"):  # diagnostic: -1 = an `auto` handle without the spill pool (A/B of its cost)
        hip["resident_spill_limit"] = int(os.environ["PYTSC_AMD_SPILL_LIMIT"])
    if n_sets > MAX_SEQUENTIAL_DEMANDS:
        # One demand per replica ("seed = replica id for throughput runs"): every set its own Poisson draw,
        # generated with arrays (pytsc_amd/flows.py::poisson_flow_sets; set g of a rank is demand `set_offset + g` of the
        # job: ranks of a multi-GPU run hold different demands, a replica holds the same one whatever the sharding).  On
        # networks whose walk routes cannot be enumerated (large grids) the per-vehicle generator tiles 64 demands instead.
        hip.update(n_flow_sets_batch=n_sets, flow_set_base=set_offset)
        cfg = Config(f"syn_{rows}x{cols}", signal={"action_space": "phase_selection", "round_robin": False}, hip=hip)
        from pytsc_amd.flows import enumerate_walk_routes
        from pytsc_amd.roadnet import load_roadnet

        rn = load_roadnet(cfg)
        if enumerate_walk_routes(rn) is not None:
            tb = build_tables(cfg, n_replicas=n_replicas, roadnet=rn, set_offset=0)  # (set r of this rank <-> its replica r)
            tb.distinct_demands = n_sets
            return tb
        from pytsc_amd.flows import poisson_flows

        base = cfg.simulator
        horizon = int(base["sim_length"]) + int(base["initial_wait_time"])
        distinct = _tiled_demands(rn, rate, horizon, MAX_TILED_DEMANDS)
        cfg = Config(f"syn_{rows}x{cols}", signal={"action_space": "phase_selection", "round_robin": False}, hip={})
        # (set behind the constructor, which deep-copies its keyword arguments: 2.8 M flow entries)
        cfg.simulator["flow_entries"] = [distinct[(set_offset + i) % MAX_TILED_DEMANDS] for i in range(n_sets)]
        tb = build_tables(cfg, n_replicas=n_replicas, roadnet=rn, set_offset=0)
        tb.distinct_demands = min(n_sets, MAX_TILED_DEMANDS)
        return tb
    cfg = Config(f"syn_{rows}x{cols}", signal={"action_space": "phase_selection", "round_robin": False}, hip=hip)
    rn = None
    if rows * cols > 64 and n_sets > 2:  # (a large grid: the same demands -- poisson_flows(seed = i) -- from one process per core)
        from pytsc_amd.roadnet import load_roadnet

        rn = load_roadnet(cfg)
        cfg.simulator["flow_entries"] = _tiled_demands(rn, rate, int(cfg.simulator["sim_length"]) + int(cfg.simulator["initial_wait_time"]), n_sets)
    tb = build_tables(cfg, n_replicas=n_replicas, roadnet=rn, set_offset=set_offset)
    tb.distinct_demands = n_sets
    return tb


_POOL_RN = None


def _one_demand(a):
    from pytsc_amd.flows import poisson_flows

    rate, horizon, seed = a
    return poisson_flows(_POOL_RN, rate, horizon, seed=seed)


def _tiled_demands(rn, rate, horizon, n):
    """The n distinct demands of a large grid (per-vehicle generator: 10 s each on the 16x16) -- one worker process per host
    core when the GPU has not been touched yet in this process (a fork after that is not safe), else one after the other."""
    global _POOL_RN
    import multiprocessing as mp

    import torch

    cores = effective_cpus()[0]
    if cores < 2 or torch.cuda.is_initialized():
        from pytsc_amd.flows import poisson_flows

        return [poisson_flows(rn, rate, horizon, seed=i) for i in range(n)]
    _POOL_RN = rn
    try:
        with mp.get_context("fork").Pool(min(cores, n, 32)) as pool:
            return pool.map(_one_demand, [(rate, horizon, i) for i in range(n)], chunksize=1)
    finally:
        _POOL_RN = None


# ------------------------------------------------------------------ CPU baseline
def effective_cpus():
    """How many host cores this process may really use: the smaller of its scheduling affinity and its cgroup's CPU
    quota (cgroup v2 `cpu.max`, v1 `cpu.cfs_quota_us / cpu.cfs_period_us`, the process's own cgroup first, then the
    root of the hierarchy as mounted in the container).  Returns (cores, affinity, quota or None)."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None

    def v2(path):
        try:
            q, per = open(path).read().split()[:2]
            return None if q == "max" else float(q) / float(per)
        except (OSError, ValueError):
            return None

    def v1(d):
        try:
            q = float(open(os.path.join(d, "cpu.cfs_quota_us")).read())
            per = float(open(os.path.join(d, "cpu.cfs_period_us")).read())
            return None if q <= 0 else q / per
        except (OSError, ValueError):
            return None

    cands = []
    try:
        for line in open("/proc/self/cgroup"):
            _, ctrl, path = line.strip().split(":", 2)
            if ctrl == "":
                cands.append(v2("/sys/fs/cgroup" + path.rstrip("/") + "/cpu.max"))
            elif "cpu" in ctrl.split(","):
                cands.append(v1("/sys/fs/cgroup/cpu" + path.rstrip("/")))
                cands.append(v1("/sys/fs/cgroup/cpu,cpuacct" + path.rstrip("/")))
    except OSError:
        pass
    cands += [v2("/sys/fs/cgroup/cpu.max"), v1("/sys/fs/cgroup/cpu"), v1("/sys/fs/cgroup/cpu,cpuacct")]
    cands = [c for c in cands if c]
    if cands:
        quota = min(cands)
    cores = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    return cores, aff, quota


def _cpu_worker(tables, seed, warmup, budget_s, start_at, q):
    """One process: the scalar f64 oracle on one replica of the workload, random valid
    actions, `warmup` untimed env steps, then as many env steps as fit in budget_s."""
    import numpy as np

    from oracle.pyoracle import Oracle

    orc = Oracle(tables)
    rng = np.random.default_rng(seed)
    out = orc.observe()

    def act(mask):
        u = rng.random(mask.shape) * mask
        return u.argmax(-1).astype(np.int32)

    for _ in range(warmup):
        out = orc.step(act(out.mask[0])[None])
    while time.time() < start_at:  # all workers enter the timed region together
        time.sleep(0.001)
    n, t0 = 0, time.perf_counter()
    while True:
        out = orc.step(act(out.mask[0])[None])
        n += 1
        if out.terminated[0]:  # simulated hour over: restart, as pytsc restart() does
            orc.reset()
            out = orc.observe()
        dt = time.perf_counter() - t0
        if dt >= budget_s:
            break
    q.put((n, dt))


def cpu_baseline(rows, cols, rate, warmup, budget_s):
    """The CPU oracle ("port": a restatement, NOT CityFlow) on this box's host cores: one
    process on one core, then one process per core.  Runs BEFORE anything touches the GPU
    (fork is only safe then).  Test-infrastructure code, used here only as the measured baseline."""
    import multiprocessing as mp

    from oracle import pyoracle

    pyoracle.build()
    tables = workload_tables(rows, cols, 1, 0, rate, 1)  # built once, shared with the workers by fork
    A = tables.n_agents
    ctx = mp.get_context("fork")
    n_cores, affinity, quota = effective_cpus()

    def run(n_procs, budget):
        q = ctx.Queue()
        start_at = time.time() + 0.5 + 0.02 * n_procs
        procs = [ctx.Process(target=_cpu_worker, args=(tables, i, warmup, budget, start_at, q)) for i in range(n_procs)]
        for p in procs:
            p.start()
        res = [q.get() for _ in procs]
        for p in procs:
            p.join()
        return sum(A * n / dt for n, dt in res), sum(n for n, _ in res), max(dt for _, dt in res)

    v1, n1, dt1 = run(1, budget_s)
    vall, nall, dtall = (v1, n1, dt1) if n_cores == 1 else run(n_cores, budget_s)
    return {
        "value": vall, "unit": "agent-env-steps/s", "cores": n_cores, "kind": "port",
        "value_1core": v1,
        # oversubscription shows here: each of the n_cores processes should run near the 1-core rate
        "value_per_process": vall / n_cores, "per_process_vs_1core": (vall / n_cores) / v1 if v1 else None,
        "affinity_cores": affinity, "cgroup_quota_cores": quota,
        "sample": f"CPU restatement (oracle), not CityFlow: 1 replica of the same {rows}x{cols} workload per "
                  f"process, random valid actions, {warmup} untimed env steps (network filling) then {dtall:.1f} s "
                  f"timed; {n_cores} processes (cores = min(scheduling affinity {affinity}, cgroup CPU quota "
                  f"{'none' if quota is None else f'{quota:.1f}'})): {nall} env steps; 1 process on 1 core: "
                  f"{n1} env steps in {dt1:.1f} s",
    }


# ------------------------------------------------------------------ launcher
def count_gpus_without_hip():
    """GPUs of this node as the kernel driver lists them (KFD topology: a node with SIMDs is a GPU), without loading
    or initialising the HIP runtime in this process -- the launcher parent must stay clean of it, its children
    initialise the GPU themselves.  None when the topology is not readable (then the workers find out)."""
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        nodes = os.listdir(root)
    except OSError:
        return None
    n = 0
    for node in nodes:
        try:
            props = dict(line.split()[:2] for line in open(os.path.join(root, node, "properties")) if line.strip())
        except (OSError, ValueError):
            continue
        if int(props.get("simd_count", 0)) > 0:
            n += 1
    visible = os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES"))
    if visible is not None and visible.strip() != "":
        n = min(n, len([v for v in visible.split(",") if v.strip() != ""]))
    return n


def launch_workers(args, argv):
    """--gpus N > 1 without a distributed launcher: start N fresh worker processes.  This process never touches the
    HIP runtime (GPUs are counted from the KFD topology in sysfs).  The children are polled together: the first one that
    fails takes the others down at once (they would otherwise sit in the rendezvous until its timeout)."""
    n = args.gpus
    if not args.stub and not args.one_gpu_rehearsal:
        have = count_gpus_without_hip()
        if have is not None and have < n:
            sys.stderr.write(f"bench.py: --gpus {n} requested but this node has {have} GPU(s)\n")
            return 2
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rcs = [None] * n
    while any(rc is None for rc in rcs):
        for i, p in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p.poll()
        if any(rc not in (None, 0) for rc in rcs):
            for i, p in enumerate(procs):  # a rank failed: stop the rest now (fresh children only; never an exec)
                if rcs[i] is None:
                    p.terminate()
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    try:
                        rcs[i] = p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        rcs[i] = p.wait()
            break
        time.sleep(0.05)
    if any(rcs):
        sys.stderr.write(f"bench.py: worker exit codes {rcs}\n")
        return 1
    return 0


# ------------------------------------------------------------------ stub (launcher test only)
class _StubEngine:
    """CPU stand-in used ONLY by tests/test_bench_launcher.py (--stub) to drive the multi-process
    path (spawn, rendezvous, barrier, max-over-ranks, one JSON line) without a GPU.  Its line is
    labelled "data": "stub" and measures nothing."""

    def __init__(self, R, A):
        import torch

        self.R, self.A = R, A
        self.buf = {"mask": torch.ones((R, A, 2), dtype=torch.uint8), "terminated": torch.zeros(R, dtype=torch.uint8)}
        self.kernel = "stub"

    def observe(self):
        return self.buf

    def step(self, actions):
        time.sleep(0.001)
        return self.buf

    def random_valid_actions(self, mask, seed=0, out=None):
        return out

    def reset(self, mask=None):
        pass

    def mean_vehicles(self):
        return 0.0

    def errors(self):
        return 0

    def profile(self, on):
        pass

    def profile_read(self):
        return 0.0, 0

    def close(self):
        pass


# ------------------------------------------------------------------ worker
PREROLL_ENV_STEPS = 150


def preroll(eng, policy, buf, n_steps=PREROLL_ENV_STEPS):
    """Roll every replica forward by a fixed 150 env steps (750 simulated seconds) OUTSIDE the timed
    region, whatever --warmup says: the 4x4 / 700 veh/h network fills for about 100 env steps (the
    driver's `--warmup 5 --steps 20` would otherwise time a network one sixth full).  Under a random
    policy the vehicle count keeps creeping up afterwards (no true steady state short of gridlock), so
    the line reports the V_a the timed region actually ran at."""
    for _ in range(n_steps):
        buf = eng.step(policy(buf["mask"]))
    return buf, n_steps, eng.mean_vehicles()


def roofline_valu(kinfo, R, va):
    """What does limit the dominant kernel: vector-ALU issue and the share of it that does useful work, from the
    committed rocprofv3 --pmc pass of this round (counters cannot be read inside this process); quoted only when the
    run is in the state the pass was taken in (V_a within 10 %)."""
    path = os.path.join(REPO, "profiles", kinfo.get("valu_file", "none"))
    if not os.path.exists(path):
        return None
    with open(path) as f:
        recs = json.load(f)
    recs = [r for r in (recs if isinstance(recs, list) else [recs]) if r.get("replicas") == R and r.get("kernel") == kinfo.get("name")]
    if not recs:
        return None
    r = min(recs, key=lambda t: abs(va - t["V_a"]))
    if abs(va - r["V_a"]) > 0.10 * r["V_a"]:
        return None
    return {k: r[k] for k in ("kernel", "V_a", "valu_busy_frac", "active_lane_frac", "useful_valu_frac", "salu_per_valu",
                              "wait_frac", "lds_bank_conflict_ratio", "waves_per_simd", "avg_launch_us_under_pmc",
                              "avg_launch_us_undisturbed", "valu_busy_frac_undisturbed", "useful_valu_frac_undisturbed",
                              "valu_insts_per_launch", "how") if k in r} | {
        "source": f"profiles/{kinfo['valu_file']}",
        # hardware counters cannot be read from inside this process: a quotation of the committed separate --pmc pass,
        # tied to this run by its state only (same kernel, same replicas, V_a within 10 %)
        "observed_in_this_run": False}


def run_hour(eng, R, A, acts, family_requested="auto", n_steps=720, win=60):
    """One simulated hour (720 env steps) of every replica of `eng` from an EMPTY network under the random policy, then
    the restart: throughput per 60-step window (HIP events at the window boundaries; the handle's state is whatever the
    caller left -- this restarts every replica first).  Returns the figures as a dict."""
    import torch

    dev = eng.device
    eng.reset()
    buf = eng.observe()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_steps // win + 2)]
    fam_steps, left_at = {"resident": 0, "list": 0}, None
    pool_steps, first_promotion = 0, None
    small_share = []  # two-class handles: share of the replicas in the small class at the start of every window
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for s in range(n_steps):
        if s % win == 0:
            if hasattr(eng, "class_counts"):
                cs, cb = eng.class_counts()  # (synchronises: twelve times an hour)
                small_share.append(round(cs / max(cs + cb, 1), 3))
            ev[s // win].record()
        buf = eng.step(eng.random_valid_actions(buf["mask"], seed=1234, out=acts))
        fam = eng.family  # (a host-side field: no synchronisation)
        fam_steps[fam] += 1
        if fam != "resident" and left_at is None and family_requested != "list":
            left_at = s
        live = eng.spill_counts()["live"] if fam == "resident" else 0  # (host-side too)
        pool_steps += live
        if live and first_promotion is None:
            first_promotion = s
    ev[n_steps // win].record()
    eng.reset(buf["terminated"])  # the restart of the hour (every replica: sim_step == 3600)
    buf = eng.observe()
    ev[n_steps // win + 1].record()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    windows = []
    for w in range(n_steps // win):
        ms = ev[w].elapsed_time(ev[w + 1])
        windows.append({"env_steps": [w * win, (w + 1) * win], "ms_per_step": ms / win, "value": R * A * win / (ms * 1e-3)})
    return {
        "value": R * A * n_steps / elapsed, "elapsed_s": elapsed, "steps": n_steps,
        "restart_ms": ev[n_steps // win].elapsed_time(ev[n_steps // win + 1]), "windows": windows,
        "left_resident_family_at_env_step": left_at,
        "share_of_replica_steps_off_resident": fam_steps["list"] / n_steps if family_requested != "list" else None,
        "stalled_replicas": eng.path_counts()["stalled"], "device_errors": eng.errors(),
        "share_of_replicas_in_the_small_class_at_window_start": small_share,
        "class_changes": eng.path_counts().get("class_change"), "steps_redone_by_the_full_size_launch": eng.path_counts().get("class_redo"),
        "mean_live_vehicles_per_replica_at_the_end": eng.mean_vehicles(),
        # per-replica promotion (include/pytsc_amd.h, TSC_FAMILY_AUTO): replica-steps served from the spill pool
        "spill_pool": dict(eng.spill_counts(), first_promotion_at_env_step=first_promotion,
                           share_of_replica_steps_in_the_pool=pool_steps / (R * n_steps)),
    }


def hour(args):
    """Sustained measurement over the reference's episode structure (pytsc/__init__.py:52-62,164-176;
    backends/cityflow/simulator.py:20-30): one simulated hour = 720 env steps from an EMPTY network under the random
    policy, then the restart of every replica.  Throughput per 60-step window, the env step at which the handle left the
    resident family (if it did) and the share of replica-steps spent off it."""
    import torch

    from pytsc_amd.vec_env import VecEngine

    R = args.replicas
    n_sets = flow_sets_of(args)
    tables = workload_tables(args.rows, args.cols, R, 0, args.rate, n_sets, args.scenario)  # (before the GPU is touched: it may fork)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    A = tables.n_agents
    eng = VecEngine(tables, device=0, want=tuple(args.outputs.split(",")), family=args.family)
    acts = torch.empty((R, A), device=dev, dtype=torch.int32)
    h = run_hour(eng, R, A, acts, args.family)
    windows = h["windows"]
    line = {
        "metric": "sustained " + METRIC, "value": h["value"], "unit": "agent-env-steps/s", "n_gpus": 1,
        "steps": h["steps"], "higher_is_better": True, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.rows}x{args.cols} synthetic grid ({A} agents), Poisson {args.rate:g} veh/h per entry road, "
                               f"{R} env replicas, {n_sets} flow sets ({getattr(tables, 'distinct_demands', n_sets)} distinct demands), "
                               f"random valid actions, one simulated hour from an empty network + the restart",
                   "replicas_per_gpu": R, "agents": A,
                   "family_requested": args.family, "resident_capacity": int(eng.tables.cfg.resident_capacity) or "default"},
        **{k: h[k] for k in ("elapsed_s", "restart_ms", "windows", "left_resident_family_at_env_step",
                             "share_of_replica_steps_off_resident", "stalled_replicas", "device_errors",
                             "mean_live_vehicles_per_replica_at_the_end", "spill_pool")},
        "vs_filled_state": {"filled_window": "env steps 120-180 (what bench.py's default timed region starts from)",
                            "value_filled": windows[2]["value"], "sustained_over_filled": h["value"] / windows[2]["value"],
                            "best_window": max(w["value"] for w in windows)},
    }
    print(json.dumps(line), flush=True)
    eng.close()
    return 1 if h["device_errors"] else 0  # (a stall that was repaired is reported -- `stalled_replicas` --, not an error)


def flow_sets_of(args):
    """--flow-sets 0 (the default): one demand per replica of this rank ("seed = replica id for throughput
    runs"); the real-network scenario has the one flow file the reference ships."""
    if args.scenario != "grid":
        return 1
    return args.replicas if args.flow_sets <= 0 else args.flow_sets


def worker(args):
    import torch

    from pytsc_amd import sharding

    rank, local_rank, world_size = sharding.init(backend="gloo" if (args.stub or args.one_gpu_rehearsal) else None)
    if args.one_gpu_rehearsal:
        local_rank = 0  # every rank on GPU 0: the N > 1 code path on one card (tests/test_hip_multirank.py), not a measurement
    if world_size != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_size}")
    cpu_line = None
    if not args.stub and not args.no_cpu_baseline and world_size == 1:
        # first, before the GPU is initialised in this process (the baseline forks workers)
        cpu_line = cpu_baseline(args.rows, args.cols, args.rate, args.cpu_warmup, args.cpu_budget)
    R = args.replicas
    if args.stub:
        dev, A = "cpu", 16
        eng = _StubEngine(R, A)
        t_tab = t_create = 0.0
        if args.stub_tables:  # (tests: what every rank of an N-GPU job does before it can create its handle)
            t_tab = time.perf_counter()
            tb_ = workload_tables(args.rows, args.cols, R, rank * R, args.rate, flow_sets_of(args), args.scenario)
            t_tab = time.perf_counter() - t_tab
            assert tb_.flows.n_sets == flow_sets_of(args)
            del tb_
    else:
        from pytsc_amd.vec_env import VecEngine

        n_sets = flow_sets_of(args)
        t_tab = time.perf_counter()
        tables = workload_tables(args.rows, args.cols, R, rank * R, args.rate, n_sets, args.scenario)  # (before the GPU is touched: it may fork)
        t_tab = time.perf_counter() - t_tab
        dev = torch.device("cuda", local_rank)
        torch.cuda.set_device(dev)
        tables.cfg.replica_base = rank * R  # the policy's draws follow the GLOBAL replica index, like the flow sets
        A = tables.n_agents
        t_create = time.perf_counter()
        want = tuple(args.outputs.split(","))
        if args.policy != "random":  # the rules read the phase state and the per-phase scores of the step just made
            want = want + tuple(k for k in ("phase", "phase_scores") if k not in want)
        eng = VecEngine(tables, device=local_rank, want=want, family=args.family)
        t_create = time.perf_counter() - t_create
    red_dev = dev if (world_size > 1 and not args.stub and not args.one_gpu_rehearsal) else "cpu"
    torch.manual_seed(0)
    buf = eng.observe()
    acts = torch.empty((R, A), device=dev, dtype=torch.int32)

    def policy(mask):
        # uniform among mask-valid actions, on device, one launch (tsc_random_valid_actions) -- or one of the reference's
        # rule-based controllers, also one launch (tsc_controller_actions, from the step's phase state and per-phase scores)
        if args.policy == "random" or args.stub:
            return eng.random_valid_actions(mask, seed=1234, out=acts)
        return eng.controller_actions(args.policy, seed=1234, out=acts)

    n_env_steps = 0

    def advance(buf, n):
        nonlocal n_env_steps
        for _ in range(n):
            buf = eng.step(policy(buf["mask"]))
            n_env_steps += 1
            if n_env_steps % 180 == 0:
                # restart() of replicas whose simulated hour is over (pytsc/__init__.py:164-176);
                # the restarted replicas are measured again so that masks describe the fresh state
                eng.reset(buf["terminated"])
                buf = eng.observe()
        return buf

    pre_steps = 0
    if not args.no_preroll and not args.stub:
        buf, pre_steps, _ = preroll(eng, policy, buf)
        n_env_steps += pre_steps
    buf = advance(buf, args.warmup)
    if not args.stub:
        torch.cuda.synchronize(dev)
    v0 = eng.mean_vehicles()
    eng.profile(4)  # HIP-event bracket around the dominant kernel on every 4th launch
    sharding.barrier()
    if not args.stub:
        torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    buf = advance(buf, args.steps)
    if not args.stub:
        torch.cuda.synchronize(dev)
    sharding.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = sharding.max_over_ranks(elapsed, device=red_dev)
    k_ms, k_n = eng.profile_read()
    eng.profile(False)
    v1 = eng.mean_vehicles()
    errors = eng.errors()
    stalled = 0 if args.stub else eng.path_counts()["stalled"]
    # two-class handles (include/pytsc_amd.h, tsc_class_counts): where the replicas are at the end of the timed region
    classes = None
    if not args.stub and hasattr(eng, "class_counts"):
        pc = eng.path_counts()
        cs, cb = eng.class_counts()
        classes = {"small_class_512_threads_two_per_cu": cs, "full_size_class": cb,
                   "class_changes_since_create": pc.get("class_change"), "steps_redone_by_the_full_size_launch": pc.get("class_redo")}
    if args.dump_state and not args.stub:
        # what this rank's replicas look like right behind the timed region (tests: a sharded job == the unsharded one)
        import numpy as np

        np.savez(f"{args.dump_state}.rank{rank}.npz", rank=rank, replica_base=rank * R,
                 stats=buf["stats"].cpu().numpy(), obs=buf["obs"].cpu().numpy(), mask=buf["mask"].cpu().numpy(),
                 counts=eng.replica_counts().cpu().numpy())
    # ---- beside the contract's numbers (N = 1 only, outside the timed region): the spread of single steps, and the same
    #      measurement in the state the default command line times (V_a grows with the env steps under a random policy)
    spread, other_state, eight_sets, hour_state = None, None, None, None
    if world_size == 1 and not args.stub and not args.no_extras:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        for i in range(args.steps):
            ev[i].record()
            buf = advance(buf, 1)
        ev[args.steps].record()
        torch.cuda.synchronize(dev)
        per = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps))
        spread = {"min": per[0], "median": per[len(per) // 2], "max": per[-1], "n": len(per),
                  "how": f"HIP events around each of {args.steps} further env steps (policy launch included) right after "
                         "the timed region; an event record costs a few microseconds of pipeline bubble, which is why "
                         "the timed region itself carries none"}
        target = PREROLL_ENV_STEPS + 20 + 240  # where the default command line (--warmup 20 --steps 240) ends
        if pre_steps and n_env_steps + 20 <= target:
            buf = advance(buf, target - 20 - n_env_steps)
            torch.cuda.synchronize(dev)
            va0 = eng.mean_vehicles()
            t1 = time.perf_counter()
            buf = advance(buf, 20)
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t1
            va1 = eng.mean_vehicles()
            other_state = {"after_env_steps": n_env_steps, "steps": 20, "ms_per_step": dt / 20 * 1e3,
                           "value": R * A * 20 / dt, "V_a": 0.5 * (va0 + va1) / A,
                           "note": "the last 20 of the env steps the default command line (--warmup 20 --steps 240) times"}
        errors = eng.errors()  # (cumulative since tsc_create: the last read stands for the whole run -- ADVICE r3)
        stalled = eng.path_counts()["stalled"]
        # ---- the same timed region on the 8 shared flow sets of rounds 1-3 (a labelled state, not `value`), and the
        #      sustained hour on this handle (0.6 s of GPU time)
        if args.scenario == "grid" and n_sets != 8 and not args.no_states:
            eng8 = VecEngine(workload_tables(args.rows, args.cols, R, 0, args.rate, 8), device=local_rank,
                             want=tuple(args.outputs.split(",")), family=args.family)
            b8 = eng8.observe()
            for _ in range(PREROLL_ENV_STEPS + args.warmup):
                b8 = eng8.step(eng8.random_valid_actions(b8["mask"], seed=1234, out=acts))
            torch.cuda.synchronize(dev)
            va80 = eng8.mean_vehicles()
            t8 = time.perf_counter()
            for _ in range(args.steps):
                b8 = eng8.step(eng8.random_valid_actions(b8["mask"], seed=1234, out=acts))
            torch.cuda.synchronize(dev)
            dt8 = time.perf_counter() - t8
            eight_sets = {"after_env_steps": PREROLL_ENV_STEPS + args.warmup + args.steps, "steps": args.steps,
                          "ms_per_step": dt8 / args.steps * 1e3, "value": R * A * args.steps / dt8,
                          "V_a": 0.5 * (va80 + eng8.mean_vehicles()) / A, "flow_sets": 8,
                          "note": "the same timed region with 8 flow sets shared by all replicas (the bench lines of rounds "
                                  "1-3: per-set static tables stay L2-resident)", "device_errors": eng8.errors()}
            errors += eight_sets["device_errors"]
            eng8.close()
        if args.scenario == "grid" and not args.no_states:
            hour_state = run_hour(eng, R, A, acts, args.family)
            errors = max(errors, hour_state["device_errors"])
            stalled = hour_state["stalled_replicas"]
    # start-up of the slowest rank (every rank generates its own demands and uploads its own tables)
    t_tab = sharding.max_over_ranks(t_tab, device=red_dev)
    t_create = sharding.max_over_ranks(t_create, device=red_dev)
    # ---- the reference's rule-based controllers driving the same workload (N = 1 extras): one states[]
    #      entry per rule -- an engine with the phase state and the per-phase scores among its outputs, restarted, rolled
    #      forward like the timed region (150 + warm-up env steps) under that rule, then the same number of timed steps
    policy_states = []
    if world_size == 1 and not args.stub and not args.no_extras and not args.no_states and args.scenario == "grid" and args.policy == "random":
        wantc = tuple(args.outputs.split(",")) + tuple(k for k in ("phase", "phase_scores") if k not in args.outputs.split(","))
        engc = VecEngine(tables, device=local_rank, want=wantc, family=args.family)
        for kind in ("random", "fixed_time", "greedy", "max_pressure", "sotl"):
            engc.reset()
            bc = engc.observe()
            step_c = (lambda b: engc.random_valid_actions(b["mask"], seed=1234, out=acts)) if kind == "random" else \
                     (lambda b, k=kind: engc.controller_actions(k, b, seed=1234, out=acts))
            for _ in range(PREROLL_ENV_STEPS + args.warmup):
                bc = engc.step(step_c(bc))
            torch.cuda.synchronize(dev)
            vc0 = engc.mean_vehicles()
            tc = time.perf_counter()
            for _ in range(args.steps):
                bc = engc.step(step_c(bc))
            torch.cuda.synchronize(dev)
            dtc = time.perf_counter() - tc
            policy_states.append({"policy": kind, "after_env_steps": PREROLL_ENV_STEPS + args.warmup + args.steps, "steps": args.steps,
                                  "ms_per_step": dtc / args.steps * 1e3, "value": R * A * args.steps / dtc,
                                  "V_a": 0.5 * (vc0 + engc.mean_vehicles()) / A, "device_errors": engc.errors(),
                                  "note": "env + controller, both on the device: the rule's one launch (tsc_controller_actions) reads "
                                          "the step's phase state and per-phase scores (phase_scores_dev); outputs of the timed "
                                          "region + phase + phase_scores; from a restart, 150 + warm-up env steps under the rule"})
            errors = max(errors, policy_states[-1]["device_errors"])
        engc.close()
    total_units = sharding.sum_over_ranks(R * A * args.steps, device=red_dev)
    v_mean = sharding.sum_over_ranks(0.5 * (v0 + v1), device=red_dev) / world_size
    value = total_units / elapsed
    ms_per_step = elapsed / args.steps * 1e3

    if rank == 0:
        kinfo = getattr(eng, "kernel_info", None) or {}
        # roofline of the dominant kernel: algorithmic bytes per agent-env-step
        # B = 48*V_a + 1356; one launch advances R*A agents by
        # kinfo["substeps_per_launch"] of the delta_time sub-steps.
        va = v_mean / A
        bytes_per_unit = 48.0 * va + 1356.0
        units_per_launch = R * A * kinfo.get("substeps_per_launch", 1) / 5.0
        avg_launch_s = (k_ms / max(k_n, 1)) * 1e-3
        achieved = bytes_per_unit * units_per_launch / avg_launch_s / 1e9 if k_n else None
        # HBM-side bytes per launch of that kernel: PMC counters cannot be read inside this process;
        # they come from the committed separate rocprofv3 --pmc passes (profiles/), and are quoted only
        # when this run is in the state those passes were taken in (V_a within 10 %).
        traffic, traffic_src = None, None
        tpath = os.path.join(REPO, "profiles", kinfo.get("traffic_file", "none"))
        if k_n and os.path.exists(tpath):
            with open(tpath) as f:
                tjs = json.load(f)
            tjs = tjs if isinstance(tjs, list) else [tjs]
            # the record taken in the state closest to this run's
            tjs = [t for t in tjs if t.get("replicas") == R and t.get("V_a") and t.get("kernel") == kinfo.get("name")]
            tj = min(tjs, key=lambda t: abs(va - t["V_a"])) if tjs else {}
            if tj and abs(va - tj["V_a"]) <= 0.10 * tj["V_a"]:
                traffic = tj["bytes_per_launch"] / avg_launch_s / 1e9
                traffic_src = f"profiles/{kinfo['traffic_file']}: {tj.get('formula', '')} at V_a {tj['V_a']:.1f}"
                if traffic > HBM_PEAK_GBS:  # cannot be: the counters over-count this access pattern
                    traffic, traffic_src = None, traffic_src + " (rejected: above peak)"
        if args.stub:
            workload_name = "stub, "
        elif args.scenario == "jinan":
            workload_name = f"Jinan 3x4 real network ({A} agents), its real flow file, {R} env replicas per GPU, "
        else:
            distinct = getattr(tables, "distinct_demands", n_sets)
            workload_name = (f"{args.rows}x{args.cols} synthetic grid ({A} agents), Poisson {args.rate:g} veh/h per entry road, "
                             f"policy {args.policy}, {n_sets} flow sets per GPU ({distinct} distinct demands" +
                             (": one per replica, drawn from its global index" if n_sets == R and distinct == R else "") +
                             f"), {R} env replicas per GPU, ")
        line = {
            "metric": METRIC if not args.stub else "STUB (launcher test)",
            "value": value, "unit": "agent-env-steps/s", "n_gpus": world_size, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "stub" if args.stub else "synthetic",
            "config": {
                "workload": workload_name + f"phase_selection action space, random valid actions, outputs {args.outputs}",
                "replicas_per_gpu": R, "agents": A, "sub_steps_per_env_step": 5,
                "state": "filled" if pre_steps else "as-started",
                "preroll_env_steps": pre_steps,
                "mean_live_vehicles_per_replica": v_mean, "V_a": va, "device_errors": errors,
                "stalled_replicas": stalled, "kernel_family": kinfo.get("family"),
                "flow_sets_per_gpu": None if args.stub else n_sets,
                "distinct_demands_per_gpu": None if args.stub else getattr(tables, "distinct_demands", n_sets),
                "startup_s": {"table_build_max_over_ranks": round(t_tab, 2), "tsc_create_max_over_ranks": round(t_create, 2)},
                "replica_classes_after_timed_region": classes,
            },
            "roofline": {
                # the bound the contract names for this tier; the path is NOT bandwidth-bound (the step would take ~50 us
                # at the peak): see roofline_valu for what limits it
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                "traffic_source": traffic_src,
                # `achieved` and `avg_launch_us` are measured in this run (HIP events on the launch stream); `traffic` is a
                # quotation of the committed separate --pmc passes, never a counter read by this process
                "traffic_observed_in_this_run": False,
                "bound_note": "contractual: HBM is the bound SURVEY.md 8(d) names for this tier; what limits the kernel is "
                              "the length of its barrier-separated phases (roofline_valu, DESIGN.md 5)",
                "kernel": kinfo.get("name"), "kernel_note": kinfo.get("note"),
                "avg_launch_us": avg_launch_s * 1e6, "launches": k_n,
                "bytes_per_agent_env_step": bytes_per_unit, "V_a": va,
            },
        }
        if spread is not None:
            line["ms_per_step_spread"] = spread
        states = [{"after_env_steps": pre_steps + args.warmup + args.steps, "steps": args.steps, "ms_per_step": ms_per_step,
                   "value": value, "V_a": va, "note": "the timed region (value)"}]
        if other_state is not None:
            states.append(other_state)
        if eight_sets is not None:
            states.append(eight_sets)
        if hour_state is not None:
            w = hour_state["windows"]
            states.append({"after_env_steps": 720, "steps": 720, "ms_per_step": hour_state["elapsed_s"] / 720 * 1e3,
                           "value": hour_state["value"], "note": "SUSTAINED: one simulated hour (720 env steps) from an empty "
                           "network + the restart of every replica, on this handle right after the timed region "
                           "(pytsc/__init__.py:52-62,164-176)",
                           "last_window_value": w[-1]["value"], "best_window_value": max(x["value"] for x in w),
                           "windows_M": [round(x["value"] / 1e6, 2) for x in w],
                           "small_class_share_at_window_start": hour_state.get("share_of_replicas_in_the_small_class_at_window_start"),
                           "left_resident_family_at_env_step": hour_state["left_resident_family_at_env_step"],
                           "share_of_replica_steps_off_resident": hour_state["share_of_replica_steps_off_resident"],
                           "mean_live_vehicles_per_replica_at_the_end": hour_state["mean_live_vehicles_per_replica_at_the_end"]})
        states += policy_states
        if len(states) > 1:
            line["states"] = states
        line["roofline_valu"] = roofline_valu(kinfo, R, va)
        if cpu_line is not None:
            line["cpu_baseline"] = cpu_line
        print(json.dumps(line), flush=True)
    eng.close()
    if world_size > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()
    if errors or stalled:  # a stalled replica returns early: such a run measures less work than it claims
        sys.stderr.write(f"bench.py: rank {rank}: device errors {errors}, stalled replicas {stalled}: the run is invalid\n")
        return 1
    return 0


def parse(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=240)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--replicas", type=int, default=4096, help="replicas per GPU")
    ap.add_argument("--rows", type=int, default=4)
    ap.add_argument("--cols", type=int, default=4)
    ap.add_argument("--rate", type=float, default=700.0, help="veh/h per entry road")
    ap.add_argument("--flow-sets", type=int, default=0,
                    help="flow sets per GPU; 0 (default) = one demand per replica (SURVEY.md 8(d): seed = replica id for "
                         "throughput runs); 8 = the shared sets of the bench lines of rounds 1-3")
    ap.add_argument("--scenario", default="grid", help="grid (rows x cols, Poisson demand) | jinan (BASELINE config 4)")
    ap.add_argument("--family", default="auto", help="kernel family: auto | list | resident")
    ap.add_argument("--policy", default="random", help="what drives the env in the timed region: random (uniform among the valid "
                    "actions, tsc_random_valid_actions) | fixed_time | greedy | max_pressure | sotl (the reference's rule-based "
                    "controllers, one fused launch each: tsc_controller_actions)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-preroll", action="store_true", help="diagnostic: time the network as started")
    ap.add_argument("--no-extras", action="store_true", help="skip the per-step spread and the other states (N = 1 extras)")
    ap.add_argument("--no-states", action="store_true", help="skip the 8-flow-set state and the sustained hour (N = 1 extras)")
    ap.add_argument("--hour", action="store_true",
                    help="sustained measurement: one simulated hour (720 env steps) from an empty network, then the restart")
    ap.add_argument("--cpu-budget", type=float, default=10.0, help="seconds per CPU-baseline leg")
    ap.add_argument("--cpu-warmup", type=int, default=150, help="untimed env steps of the CPU baseline")
    ap.add_argument("--outputs", default=DEFAULT_OUTPUTS)
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)  # launcher test only
    ap.add_argument("--stub-tables", action="store_true", help=argparse.SUPPRESS)  # ... with every rank's REAL table generation (start-up)
    ap.add_argument("--one-gpu-rehearsal", action="store_true",
                    help="tests: run the --gpus N code path with every rank on GPU 0 (gloo for the barrier / reductions)")
    ap.add_argument("--dump-state", default=None, help="tests: write <path>.rank<k>.npz with each rank's outputs behind the timed region")
    return ap.parse_args(argv)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_workers(args, argv)
    if args.hour:
        return hour(args)
    return worker(args)


if __name__ == "__main__":
    sys.exit(main())
