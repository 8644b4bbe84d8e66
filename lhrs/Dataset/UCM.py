"""lhrs.Dataset.UCM (UCM.py)."""
from lhrs_bot_amd.eval_datasets import UCM  # noqa: F401
