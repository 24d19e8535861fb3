"""Small helpers with the reference's behaviour (pytsc/common/utils.py)."""
import logging

import numpy as np


class EnvLogger:
    logger = None

    @staticmethod
    def get_logger(name=__name__, level=logging.INFO):
        if EnvLogger.logger is None:
            EnvLogger.logger = logging.getLogger(name)
            if not EnvLogger.logger.hasHandlers():
                h = logging.StreamHandler()
                h.setFormatter(logging.Formatter("%(asctime)s - %(levelname)s - %(message)s"))
                EnvLogger.logger.addHandler(h)
            EnvLogger.logger.setLevel(level)
        return EnvLogger.logger

    @staticmethod
    def log_info(msg):
        EnvLogger.get_logger().log(logging.INFO, msg)

    @staticmethod
    def log_warning(msg):
        EnvLogger.get_logger().log(logging.WARNING, msg)

    @staticmethod
    def set_log_level(level):
        EnvLogger.get_logger().setLevel(level)


def validate_input_against_allowed(inp_cfg_str, allowed_cfg_strs):
    assert (
        inp_cfg_str in allowed_cfg_strs
    ), f"Invalid action space `{inp_cfg_str}`. Allowed values are {allowed_cfg_strs}."


def pad_list(inp_list, size, pad_value=0):
    """Reference semantics (utils.py:91-112): pads through an array created with
    np.full(size, pad_value) — an INTEGER pad value therefore truncates every
    float entry toward zero whenever padding happens."""
    arr = np.asarray(np.array(inp_list))
    if len(arr) >= size:
        return arr.tolist()
    out = np.full(size, pad_value)
    out[: len(arr)] = arr
    return out.tolist()
