"""lhrs.Dataset.meterml (meterml.py)."""
from lhrs_bot_amd.eval_datasets import METERMLDataset  # noqa: F401
