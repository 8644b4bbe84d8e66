"""lhrs.Dataset.ImageFolderInstance (ImageFolderInstance.py)."""
from lhrs_bot_amd.eval_datasets import CLASS_NAME_MAP, ImageFolderInstance  # noqa: F401
