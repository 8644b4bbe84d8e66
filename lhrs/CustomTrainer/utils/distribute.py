"""lhrs.CustomTrainer.utils.distribute: the rank helpers the evaluation scripts import (distribute.py:17-20, 391-483, 502-560)."""
from lhrs_bot_amd.boundary import deepspeed_init_distributed, init_distributed  # noqa: F401
from lhrs_bot_amd.evaluation import get_rank, get_world_size, is_distributed, is_main_process  # noqa: F401
