"""lhrs.Dataset.millionaid_eval (millionaid_eval.py)."""
from lhrs_bot_amd.eval_datasets import MillionAidEval  # noqa: F401
