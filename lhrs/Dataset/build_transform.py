"""lhrs.Dataset.build_transform (build_transform.py:9-40, 43-45)."""
from lhrs_bot_amd.datasets import build_vlp_transform  # noqa: F401
from lhrs_bot_amd.eval_datasets import build_cls_transform  # noqa: F401
