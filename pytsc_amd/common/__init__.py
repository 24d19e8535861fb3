from pytsc_amd.common.actions import PhaseSelectionActionSpace, PhaseSwitchActionSpace
from pytsc_amd.common.observations import LaneFeatures, PositionMatrix
from pytsc_amd.common.reward import MaxPressure, QueueLength

# same registries / keys as the reference (pytsc/common/__init__.py:5-18)
ACTION_SPACES = {
    "phase_selection": PhaseSelectionActionSpace,
    "phase_switch": PhaseSwitchActionSpace,
}
OBSERVATION_SPACES = {
    "position_matrix": PositionMatrix,
    "lane_features": LaneFeatures,
}
REWARD_FUNCTIONS = {
    "queue_length": QueueLength,
    "max_pressure": MaxPressure,
}
