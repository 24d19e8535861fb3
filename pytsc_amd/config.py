"""Configuration: default YAML <- scenario YAML <- kwargs, three-level dict merge.

Same behaviour as the reference (pytsc/common/config.py:37-70,
pytsc/backends/cityflow/config.py:20-76): sections ``network``, ``signal``,
``misc`` and the backend section, exposed as ``.network/.signal/.misc/.simulator``.
Scenario directories in the reference's layout (``config.yaml`` + roadnet +
flow JSONs) are accepted as-is: a ``cityflow:`` section is read as the backend
section, so a PyTSC user can point ``scenario_dir`` at their existing scenarios.
Synthetic scenarios named ``syn_<rows>x<cols>`` that have no directory are
generated on the fly (pytsc_amd/grid.py, pytsc_amd/flows.py).
"""
import copy
import os
import random
import re
from itertools import cycle

import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
SCENARIO_ROOT = os.path.join(HERE, "scenarios")
BACKEND = "hip"


def recursively_update_dict(d, u):
    for k, v in u.items():
        if isinstance(v, dict):
            d[k] = recursively_update_dict(d.get(k, {}), v)
        else:
            d[k] = v
    return d


class Config:
    def __init__(self, scenario, scenario_dir=None, debug=True, **kwargs):
        self.scenario = scenario
        self.debug = debug
        self._additional_config = kwargs
        with open(os.path.join(SCENARIO_ROOT, "default", "config.yaml"), "r") as f:
            cfg = yaml.safe_load(f)
        if scenario_dir is None:
            scenario_dir = os.path.join(SCENARIO_ROOT, BACKEND, scenario)
        self.dir = os.path.join(os.path.abspath(scenario_dir), "")
        scen_yaml = os.path.join(self.dir, "config.yaml")
        if os.path.exists(scen_yaml):
            with open(scen_yaml, "r") as f:
                scen = yaml.safe_load(f) or {}
            scen = copy.deepcopy(scen)
            if "cityflow" in scen:  # reference-layout scenario
                recursively_update_dict(cfg[BACKEND], scen.pop("cityflow"))
            recursively_update_dict(cfg, scen)
        extra = copy.deepcopy(kwargs)
        if "cityflow" in extra:
            recursively_update_dict(cfg[BACKEND], extra.pop("cityflow"))
        for k in list(extra.keys()):
            if k not in ("network", "signal", "misc", BACKEND):
                extra.pop(k)
        recursively_update_dict(cfg, extra)
        self.network = cfg["network"]
        self.signal = cfg["signal"]
        self.misc = cfg["misc"]
        self.simulator = cfg[BACKEND]
        self.synthetic_grid = None
        m = re.fullmatch(r"syn_(\d+)x(\d+)", scenario)
        if "roadnet_json" in self.simulator:  # in-memory roadnet dict (tests, fixtures)
            self.roadnet_file = None
        elif "roadnet_file" in self.simulator:
            self.roadnet_file = os.path.join(self.dir, self.simulator["roadnet_file"])
        elif m:
            self.synthetic_grid = (int(m.group(1)), int(m.group(2)))
            self.roadnet_file = None
        else:
            raise FileNotFoundError(
                f"scenario {scenario!r}: no roadnet_file in {scen_yaml} and not a syn_RxC name"
            )
        self.flow_files_cycle = cycle(self.simulator.get("flow_files", []))
        self.flow_file = None
        # backends/cityflow/config.py:58-61
        assert (
            self.signal["yellow_time"] == self.simulator["delta_time"]
        ), "Delta time and yellow times must be fixed to 5 seconds."
        random.seed(self.simulator["seed"])

    def _set_flow_file(self):
        """Which flow file the next (re)start uses (reference backends/cityflow/config.py:63-76):
        `constant` -> flow_file; `random` -> a draw of Python's global `random` (seeded in
        __init__, as the reference does) over flow_files; `sequential` -> the next of the cycle."""
        kind = self.simulator.get("flow_rate_type", "constant")
        pick = {
            "constant": lambda: self.simulator.get("flow_file"),
            "random": lambda: random.choice(self.simulator["flow_files"]),
            "sequential": lambda: next(self.flow_files_cycle),
        }.get(kind)
        if pick is None:
            raise ValueError(f"flow_rate_type {kind!r}: expected constant, random or sequential")
        self.flow_file = pick()
        return self.flow_file


class DisruptedConfig(Config):
    """Domain-randomised demand (reference: pytsc/backends/cityflow/config.py:106-175, selected by
    `TrafficSignalNetwork(..., disrupted=True)`, pytsc/__init__.py:26-30): the backend section holds, per `mode`
    (train / test), `{domain: {disruption value: [flow files]}}` with the files under `<mode>/<domain>/<value>/`.  Every
    (re)start draws a domain, a disruption value of it and one of its files with Python's global `random` (seeded in
    __init__, as the reference does) -- or takes the pinned `domain_class` = (domain, value) and draws the file only.
    `domain_classes` lists the (domain, value) pairs in config order; `current_domain_class` is the index of the last draw.

    On top of the reference's attributes: every file of the mode is a flow set of the device tables
    (`simulator["flow_files"]`, in domain-class order), so a restart is a set index handed to `tsc_reset_sets`;
    `draw_sets(n, generator)` makes the same three-level draw for n replicas at once (the batched environment)."""

    def __init__(self, scenario, mode="train", **kwargs):
        self.mode = mode
        self.domain_class = kwargs.get("domain_class", None)
        super().__init__(scenario, **kwargs)
        section = self.simulator[mode]
        self.domains = list(section.keys())
        self.disrup_values = {d: list(section[d].keys()) for d in self.domains}
        self.domain_classes = [(d, v) for d in self.domains for v in self.disrup_values[d]]
        files, owner = [], []
        for ci, (d, v) in enumerate(self.domain_classes):
            for f in section[d][v]:
                files.append(os.path.join(mode, d, v, f))
                owner.append(ci)
        self.simulator["flow_files"] = files          # set index <-> file, for the facade's restart and the tables
        self.simulator["flow_rate_type"] = "random"   # (the reference's _set_flow_file ignores the key: always a draw)
        self.file_domain_class = owner
        self.flow_files_cycle = cycle(files)
        self.current_domain_class = None
        random.seed(self.simulator["seed"])

    def _set_flow_file(self):
        if self.domain_class is None:
            domain = random.choice(self.domains)
            value = random.choice(self.disrup_values[domain])
        else:
            domain, value = self.domain_class[0], self.domain_class[1]
        self.current_domain_class = self.domain_classes.index((domain, value))
        name = random.choice(self.simulator[self.mode][domain][value])
        self.flow_file = os.path.join(self.mode, domain, value, name)
        return self.flow_file

    def set_domain_class(self, domain_class):
        self.domain_class = domain_class

    def draw_sets(self, n, generator=None):
        """The draw of `_set_flow_file` for n replicas at once: ([n] int32 flow-set indices, [n] int64 domain-class indices)
        as torch tensors -- uniform over the domains, then over the domain's disruption values, then over that class's
        files (NOT uniform over files: classes hold different numbers of them); a pinned domain_class draws the file only."""
        import torch

        sizes = torch.tensor([len(self.simulator[self.mode][d][v]) for d, v in self.domain_classes])
        first = torch.cumsum(sizes, 0) - sizes
        if self.domain_class is None:
            n_val = torch.tensor([len(self.disrup_values[d]) for d in self.domains])
            class0 = torch.cumsum(n_val, 0) - n_val
            dom = torch.randint(0, len(self.domains), (n,), generator=generator)
            val = (torch.rand(n, generator=generator) * n_val[dom]).long().clamp_(max=n_val.max() - 1)
            val = torch.minimum(val, n_val[dom] - 1)
            cls = class0[dom] + val
        else:
            cls = torch.full((n,), self.domain_classes.index((self.domain_class[0], self.domain_class[1])), dtype=torch.long)
        k = torch.minimum((torch.rand(n, generator=generator) * sizes[cls]).long(), sizes[cls] - 1)
        return (first[cls] + k).to(torch.int32), cls
