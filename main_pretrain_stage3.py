"""Stage-3 entry point: the reference's main_pretrain_stage3.py differs from stage 1 only in the YAML it is given (`stage`, `lora`,
`bits`, `optimizer: adamw`) and - stage 3 - in using IterBasedTrainer(max_iters=config.epochs); the shared driver reads all of that
from the config (/root/reference main_pretrain_stage3.py:180-245)."""
import json
import logging
import os

import torch

import main_pretrain_stage1 as drv

if __name__ == "__main__":
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")
    config = drv.parse_option()
    config.setdefault("stage", 3)
    config.rank, config.local_rank, config.world_size = drv.init_distributed()
    config.is_distribute = config.world_size > 1
    config.seed = config.seed + config.rank
    torch.manual_seed(config.seed)
    os.makedirs(config.output, exist_ok=True)
    if config.rank == 0:
        with open(os.path.join(config.output, "config.json"), "w") as f:
            json.dump({k: v for k, v in config.items() if not isinstance(v, torch.device)}, f, indent=1, default=str)
    drv.main(config)
