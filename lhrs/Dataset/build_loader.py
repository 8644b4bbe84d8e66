"""lhrs.Dataset.build_loader (build_loader.py:25-57, 60-162, 202-212)."""
from lhrs_bot_amd.datasets import build_loader, build_loader_hepler, build_vlp_loader  # noqa: F401
