"""finenvs_amd -- MI355X-native hot path of hmomin/FinEnvs' TimeSeriesEnv.

Only the vectorised financial time-series environment is here (the path named by
BASELINE.json's north_star); agents stay plain PyTorch-ROCm user code.
"""
from .environments.time_series_env import TimeSeriesEnv, shard_range  # noqa: F401

__all__ = ["TimeSeriesEnv", "shard_range"]
__version__ = "0.1.0"
