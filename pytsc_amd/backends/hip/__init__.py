"""The MI355X-native backend: same registry keys as the reference's
CITYFLOW_MODULES (pytsc/backends/cityflow/__init__.py:18-26)."""
from pytsc_amd.backends.hip.metrics import MetricsParser
from pytsc_amd.backends.hip.retriever import Retriever
from pytsc_amd.backends.hip.simulator import Simulator
from pytsc_amd.common.traffic_signal import TrafficSignal
from pytsc_amd.config import Config, DisruptedConfig
from pytsc_amd.network_parser import NetworkParser

HIP_MODULES = {
    "config": Config,
    "disrupted_config": DisruptedConfig,
    "metrics_parser": MetricsParser,
    "network_parser": NetworkParser,
    "retriever": Retriever,
    "simulator": Simulator,
    "traffic_signal": TrafficSignal,
}
