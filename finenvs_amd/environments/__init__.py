from .time_series_env import TimeSeriesEnv, shard_range  # noqa: F401
