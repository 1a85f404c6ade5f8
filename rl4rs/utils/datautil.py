from rl4rs_amd.utils.datautil import FeatureUtil  # noqa: F401
