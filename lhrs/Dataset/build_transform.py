"""lhrs.Dataset.build_transform (build_transform.py:43-45)."""
from lhrs_bot_amd.datasets import build_vlp_transform  # noqa: F401
