"""lhrs.optimizer (optimizer/__init__.py)."""
from lhrs_bot_amd.boundary import build_optimizer, get_param_group  # noqa: F401
