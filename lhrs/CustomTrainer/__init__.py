"""lhrs.CustomTrainer (lhrs/CustomTrainer/__init__.py): trainers, hooks, distributed init - plus `initialize`, this engine's stand-in
for `deepspeed.initialize` (same keywords and 4-tuple; see lhrs_bot_amd/boundary.py and INTEGRATION.md)."""
from lhrs_bot_amd.boundary import deepspeed_init_distributed, init_distributed, initialize, setup_logger  # noqa: F401
from lhrs_bot_amd.evaluation import get_rank, get_world_size, is_distributed, is_main_process  # noqa: F401
from lhrs_bot_amd.trainer import (CosineAnnealingLrUpdaterHook, DistributedHook, EngineStepHook, EpochBasedTrainer, FixedLrUpdaterHook,  # noqa: F401
                                  HookBase, IterBasedTrainer, IterCheckpointerHook, LoggerHook, Trainer)
from .utils import ConfigArgumentParser  # noqa: F401

DeepSpeedHook = EngineStepHook  # hook/deepspeed_hook.py:4-19: backward + step after every iteration
