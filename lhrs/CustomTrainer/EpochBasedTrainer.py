"""lhrs.CustomTrainer.EpochBasedTrainer (EpochBasedTrainer.py:56-109)."""
from lhrs_bot_amd.trainer import EpochBasedTrainer  # noqa: F401
