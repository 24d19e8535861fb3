"""Scenario assembly: Config -> RoadNet -> NetworkParser -> flow sets -> Tables."""
import os

from pytsc_amd.flows import FlowCompiler, load_flow_file, poisson_flows
from pytsc_amd.network_parser import NetworkParser
from pytsc_amd.roadnet import load_roadnet
from pytsc_amd.tables import Tables


def flow_entry_sets(config, rn):
    """List of flow-entry lists: the scenario's flow file(s), or synthetic
    Poisson demand for syn_RxC scenarios."""
    sim = config.simulator
    if "flow_entries" in sim:  # in-memory demand (tests, bench)
        return list(sim["flow_entries"])
    files = []
    kind = sim.get("flow_rate_type", "constant")
    if kind == "constant" and sim.get("flow_file"):
        files = [sim["flow_file"]]
    elif sim.get("flow_files"):
        files = list(sim["flow_files"])
    elif sim.get("flow_file"):
        files = [sim["flow_file"]]
    if files:
        return [load_flow_file(os.path.join(config.dir, f)) for f in files]
    n_sets = int(sim.get("n_flow_sets", 1))
    rate = float(sim.get("flow_rate", 700.0))
    horizon = int(sim["sim_length"]) + int(sim["initial_wait_time"])
    seed = int(sim.get("flow_seed", sim["seed"]))
    return [poisson_flows(rn, rate, horizon, seed=seed + i) for i in range(n_sets)]


def build_tables(config, n_replicas=None, roadnet=None, set_offset=0):
    rn = roadnet if roadnet is not None else load_roadnet(config)
    parser = NetworkParser(config, roadnet=rn)
    fc = FlowCompiler(rn, interval=float(config.simulator["interval"]))
    horizon = int(config.simulator["sim_length"]) + int(config.simulator["initial_wait_time"])
    sim = config.simulator
    batch = int(sim.get("n_flow_sets_batch", 0) or 0)
    done = False
    if batch > 0 and config.synthetic_grid is not None and "flow_entries" not in sim:
        # many distinct synthetic demands (one per replica in throughput runs): the array generator; set g of
        # the batch is demand `flow_set_base + g`, whatever the batch (ranks of a multi-GPU run generate their own shards)
        from pytsc_amd.flows import poisson_flow_sets

        base = int(sim.get("flow_set_base", 0))
        mix = sim.get("flow_rate_mix")  # [[rate, share], ...]: demand g runs at the rate its index falls to (a heterogeneous batch)
        if mix:
            from pytsc_amd.flows import enumerate_walk_routes

            done = enumerate_walk_routes(rn) is not None
            bounds, acc = [], 0.0
            for rate, share in mix:
                acc += float(share)
                bounds.append((acc, float(rate)))
            if done:
                groups = {}
                for g in range(base, base + batch):
                    u = ((g * 2654435761) % 4294967296) / 4294967296.0 * acc  # (a fixed pseudo-random position of demand g)
                    rate = next(r for b, r in bounds if u < b or b == bounds[-1][0])
                    groups.setdefault(rate, []).append(g)
                by_g, n0 = {}, len(fc.sets)
                for rate, gs in groups.items():  # one call per rate (the walk routes are enumerated once per call)
                    poisson_flow_sets(fc, rate, horizon, gs, base_seed=int(sim.get("flow_seed", sim["seed"])))
                    by_g.update(zip(gs, fc.sets[-len(gs):]))
                fc.sets[n0:] = [by_g[g] for g in range(base, base + batch)]
        else:
            done = poisson_flow_sets(fc, float(sim.get("flow_rate", 700.0)), horizon, range(base, base + batch),
                                     base_seed=int(sim.get("flow_seed", sim["seed"])))
    if not done:
        for entries in flow_entry_sets(config, rn):
            fc.add(entries, horizon)
    if n_replicas is None:
        n_replicas = int(config.simulator.get("n_replicas", 1))
    return Tables(parser, fc, config, n_replicas, set_offset=set_offset)
