"""lhrs.CustomTrainer.utils (utils/__init__.py): the names the entry scripts import."""
from lhrs_bot_amd.boundary import auto_resume_helper, deepspeed_init_distributed, init_distributed, setup_logger  # noqa: F401
from lhrs_bot_amd.evaluation import get_rank, get_world_size, is_distributed, is_main_process  # noqa: F401
from lhrs_bot_amd.datasets import InfiniteSampler  # noqa: F401
from lhrs_bot_amd.trainer import ConfigArgumentParser, ConfigDict, str2bool  # noqa: F401
