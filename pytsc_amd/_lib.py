"""ctypes loader of the HIP library (pytsc_amd/lib/libpytsc_amd.so).

There is NO CPU fallback: if the library is missing or no GPU is usable the
product path raises.  The library is built in-tree by pytsc_amd/build.py
(__graft_entry__.build()).
"""
import ctypes as C
import os

from pytsc_amd.tables import TscConfig, TscFlows, TscNet, TscStepOut

HERE = os.path.dirname(os.path.abspath(__file__))
# (PYTSC_AMD_LIB: another build of the same library -- same-box A/B of kernel variants, tools/ab_bench.py)
LIB_PATH = os.environ.get("PYTSC_AMD_LIB") or os.path.join(HERE, "lib", "libpytsc_amd.so")

# every symbol include/pytsc_amd.h declares
SYMBOLS = (
    "tsc_abi_version", "tsc_create", "tsc_destroy", "tsc_reset", "tsc_step", "tsc_observe",
    "tsc_lane_detail", "tsc_snapshot", "tsc_errors", "tsc_mean_vehicles", "tsc_obs_size",
    "tsc_state_size", "tsc_n_actions", "tsc_profile_enable", "tsc_profile_read",
    "tsc_lane_window_counts", "tsc_random_valid_actions", "tsc_reset_sets", "tsc_replica_counts", "tsc_kernel_family", "tsc_kernel_variant", "tsc_check_tables", "tsc_path_counts", "tsc_park", "tsc_spill_counts", "tsc_controller_actions", "tsc_class_counts",
)

ERRORS = {-1: "TSC_EINVAL (bad argument / inconsistent tables)",
          -2: "TSC_ENODEV (no usable GPU / HIP failure)",
          -3: "TSC_ENOMEM", -4: "TSC_EUNSUP (tables use a feature the HIP path rejects)"}

_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -m pytsc_amd.build` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback."
        )
    # torch first: the library must bind to the SAME HIP runtime (libamdhip64) that
    # torch has loaded, since it operates on torch-owned device buffers and streams.
    import torch  # noqa: F401

    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    lib.tsc_abi_version.restype = C.c_int
    lib.tsc_create.argtypes = [C.POINTER(TscNet), C.POINTER(TscFlows), C.POINTER(TscConfig), C.c_int,
                               C.POINTER(vp)]
    lib.tsc_destroy.argtypes = [vp]
    lib.tsc_reset.argtypes = [vp, vp, vp]
    lib.tsc_reset_sets.argtypes = [vp, vp, vp, vp]
    lib.tsc_replica_counts.argtypes = [vp, vp, vp]
    lib.tsc_park.argtypes = [vp, vp, vp]
    lib.tsc_spill_counts.argtypes = [vp, C.POINTER(C.c_int32)]
    lib.tsc_step.argtypes = [vp, vp, C.POINTER(TscStepOut), vp]
    lib.tsc_observe.argtypes = [vp, C.POINTER(TscStepOut), vp]
    lib.tsc_lane_detail.argtypes = [vp, vp, vp, vp, C.c_int32, vp]
    lib.tsc_lane_window_counts.argtypes = [vp, vp, vp]
    lib.tsc_random_valid_actions.argtypes = [vp, vp, C.c_uint64, vp, vp]
    lib.tsc_snapshot.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, vp, vp]
    lib.tsc_errors.argtypes = [vp, C.POINTER(C.c_int32)]
    lib.tsc_path_counts.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.tsc_mean_vehicles.argtypes = [vp, C.POINTER(C.c_double)]
    lib.tsc_profile_enable.argtypes = [vp, C.c_int]
    lib.tsc_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    for f in ("tsc_obs_size", "tsc_state_size", "tsc_n_actions", "tsc_kernel_family", "tsc_kernel_variant"):
        getattr(lib, f).argtypes = [vp]
    lib.tsc_check_tables.argtypes = [C.POINTER(TscNet), C.POINTER(TscFlows), C.POINTER(TscConfig),
                                     C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.tsc_class_counts.argtypes = [vp, C.POINTER(C.c_int32)]
    lib.tsc_controller_actions.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, vp, vp, vp, vp]
    if lib.tsc_abi_version() != 5:
        raise RuntimeError("libpytsc_amd.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {ERRORS.get(rc, rc)}")
