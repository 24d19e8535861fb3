"""Small helpers behaving as the reference's (pytsc/common/utils.py:9-112): the logger facade, the allowed-value
check of the config strings and the integer-truncating padding."""
import logging

import numpy as np

log = logging.getLogger("pytsc_amd")
if not log.hasHandlers():  # (hasHandlers also sees the root logger's: an application that configured logging gets each
    _h = logging.StreamHandler()  # message once, through its own handlers -- ADVICE r3)
    _h.setFormatter(logging.Formatter("%(asctime)s - %(levelname)s - %(message)s"))
    log.addHandler(_h)
    log.propagate = False
log.setLevel(logging.INFO)


def _get_logger(name=None, level=None):
    """The package logger; a level, when given, is set -- as the reference's get_logger does on first use."""
    if level is not None:
        log.setLevel(level)
    return log


class EnvLogger:
    """Name kept for callers of the reference's helper (EnvLogger.set_log_level(logging.WARNING) is how its users
    silence the config dump); everything goes to the package logger `pytsc_amd`."""

    get_logger = staticmethod(_get_logger)
    log_info = staticmethod(log.info)
    log_warning = staticmethod(log.warning)
    set_log_level = staticmethod(log.setLevel)


def validate_input_against_allowed(value, allowed):
    """The reference rejects an unknown config string with an AssertionError (utils.py:52-55); so does this."""
    allowed = list(allowed)
    if value not in allowed:
        raise AssertionError(f"`{value}` is not one of the supported values {allowed}")


def pad_list(inp_list, size, pad_value=0):
    """Reference semantics (utils.py:91-112): pads through an array created with
    np.full(size, pad_value) -- an INTEGER pad value therefore truncates every
    float entry toward zero whenever padding happens."""
    arr = np.asarray(np.array(inp_list))
    if len(arr) >= size:
        return arr.tolist()
    out = np.full(size, pad_value)
    out[: len(arr)] = arr
    return out.tolist()
