"""lhrs.Dataset.build_loader (build_loader.py:25-57, 60-162, 164-199, 202-212)."""
from lhrs_bot_amd.datasets import build_loader, build_loader_hepler, build_vlp_loader  # noqa: F401
from lhrs_bot_amd.eval_datasets import build_zero_shot_loader  # noqa: F401
