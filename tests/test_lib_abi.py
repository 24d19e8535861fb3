"""The C-ABI library builds for gfx950, loads without a GPU and exports every
symbol include/pytsc_amd.h declares (no compute call here)."""
import ctypes
import os
import re

from conftest import REPO


def test_header_symbols_are_exported():
    from pytsc_amd import _lib, build

    lib_path = build.build()
    assert os.path.exists(lib_path)
    hdr = open(os.path.join(REPO, "include", "pytsc_amd.h")).read()
    declared = set(re.findall(r"^\s*int\s+(tsc_[a-z_0-9]+)\s*\(", hdr, flags=re.M))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.load()
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert lib.tsc_abi_version() == 5


def test_struct_layouts_match_header():
    """ctypes mirrors of tsc_net_t / tsc_flows_t / tsc_config_t / tsc_step_out_t
    have the field order of the header."""
    from pytsc_amd import tables

    hdr = open(os.path.join(REPO, "include", "pytsc_amd.h")).read()

    def fields(struct_name):
        body = re.search(r"typedef struct %s \{(.*?)\} %s_t;" % (struct_name, struct_name), hdr, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            m = re.search(r"([A-Za-z_0-9]+)\s*$", decl.strip())
            if m and decl.strip():
                names.append(m.group(1))
        return names

    for cname, cls in (("tsc_net", tables.TscNet), ("tsc_flows", tables.TscFlows),
                       ("tsc_config", tables.TscConfig), ("tsc_step_out", tables.TscStepOut)):
        assert fields(cname) == [f[0] for f in cls._fields_], cname


def test_product_path_fails_loudly_without_gpu():
    import pytest
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pytsc_amd import TrafficSignalNetwork

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        TrafficSignalNetwork("syn_2x2", "hip")


def test_package_does_not_import_the_oracle():
    pkg = os.path.join(REPO, "pytsc_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc")):
                src = open(os.path.join(root, f)).read()
                assert "pyoracle" not in src and "tsc_oracle" not in src and "from oracle" not in src, f
