"""lhrs.utils (utils/__init__.py -> eval_utils.py:4-56)."""
import torch

from lhrs_bot_amd.eval_utils import KeywordsStoppingCriteria  # noqa: F401

type_dict = {"float32": torch.float32, "float16": torch.float16, "bfloat16": torch.bfloat16}
