"""lhrs.CustomTrainer.IterBasedTrainer (IterBasedTrainer.py:49-91)."""
from lhrs_bot_amd.trainer import IterBasedTrainer  # noqa: F401
