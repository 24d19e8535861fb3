"""Record what the REFERENCE's DisruptedConfig (pytsc/backends/cityflow/config.py:106-175) and the façade around it
(pytsc/__init__.py:26-30,153-155,173-174) do with a `train:` / `test:` section, into tests/golden/disrupted_config/
expected.json: the domain classes in their order, and the sequence of (flow file, current_domain_class) its
`_set_flow_file()` draws from Python's global `random` after `random.seed(seed)` -- free draws, a pinned domain class
given at construction, and one set later through `set_domain_class`.  The reference is imported from /root/reference
(never copied); the section itself is this repository's (the reference ships its own commented out).

Run (build container only):  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_disrupted_golden.py
"""
import contextlib
import io
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
sys.dont_write_bytecode = True
os.environ.setdefault("SUMO_HOME", "/tmp/none")
sys.path[:0] = [os.path.join(HERE, "ref_stubs"), REPO, REF]

SECTION = {
    "train": {
        "flow_disrupted": {
            "600": [f"{i}__jinan_3_4__gaussian_600_flows.json" for i in range(3)],
            "700": [f"{i}__jinan_3_4__gaussian_700_flows.json" for i in range(5)],
            "0_2": [f"{i}__jinan_3_4__fd_0.2__gaussian_700_flows.json" for i in range(2)],
        },
        "link_disrupted": {
            "0_1": [f"{i}__jinan_3_4__dr_0.1__gaussian_700_flows.json" for i in range(4)],
            "0_3": [f"{i}__jinan_3_4__dr_0.3__gaussian_700_flows.json" for i in range(2)],
        },
    },
    "test": {
        "link_disrupted": {"0_2": [f"{i}__jinan_3_4__dr_0.2__gaussian_700_flows.json" for i in range(3)]},
        "flow_disrupted": {"0_4": [f"{i}__jinan_3_4__fd_0.4__gaussian_700_flows.json" for i in range(3)]},
    },
}


def main():
    from pytsc.backends.cityflow.config import DisruptedConfig

    # The reference's DisruptedConfig.__init__ never calls BaseConfig.__init__, so `self.debug` -- read by _load_config --
    # does not exist and the class raises AttributeError as shipped (a latent bug: the shipped scenarios have their
    # `train:` sections commented out).  A class attribute stands in for the missing assignment; nothing else is touched.
    DisruptedConfig.debug = True
    out = {"scenario": "jinan_3_4", "section": SECTION, "cases": []}
    for mode, seed, pinned, later in (("train", 0, None, None), ("train", 7, None, ["link_disrupted", "0_3"]),
                                      ("test", 3, None, None), ("train", 1, ["flow_disrupted", "700"], None)):
        kw = {"cityflow": dict(SECTION, seed=seed)}
        if pinned is not None:
            kw["domain_class"] = tuple(pinned)
        with contextlib.redirect_stdout(io.StringIO()):
            cfg = DisruptedConfig("jinan_3_4", mode=mode, **kw)
            draws = []
            for i in range(12):
                if later is not None and i == 6:
                    cfg.set_domain_class(tuple(later))
                cfg._set_flow_file()
                draws.append([cfg.flow_file, cfg.current_domain_class])
        out["cases"].append({"mode": mode, "seed": seed, "domain_class": pinned, "set_domain_class_before_draw_6": later,
                             "domains": cfg.domains, "disrup_values": cfg.disrup_values,
                             "domain_classes": [list(c) for c in cfg.domain_classes], "draws": draws})
    with open(os.path.join(HERE, "disrupted_config", "expected.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(out["cases"]), "cases")


if __name__ == "__main__":
    main()
