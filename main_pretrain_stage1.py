#!/usr/bin/env python
"""Stage-1 pretraining driver on the gfx950 engine - the reference's `main_pretrain_stage1.py` surface
(/root/reference main_pretrain_stage1.py:88-309) kept flag-compatible for the keys that reach the hot path:

    python -m torch.distributed.run --nproc-per-node 8 main_pretrain_stage1.py -c Config/multi_modal_stage1.yaml \
        --batch-size 8 --workers 4 --data-path <dir> --output <dir> --accelerator gpu --enable-amp True --use-checkpoint

build_model -> prepare_for_training -> engine (replaces deepspeed.initialize) -> EpochBasedTrainer.train -> FINAL.pt.
No tokenizer / dataset files exist offline, so `--data-path synthetic` (default) feeds the stage-1 batch contract from
SyntheticStage1Loader; a real loader only has to yield the same dict (SURVEY.md §8 a12, row f-2 is out of scope).
"""
import json
import logging
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lhrs_bot_amd.engine import LHRSEngine  # noqa: E402
from lhrs_bot_amd.trainer import (ConfigArgumentParser, ConfigDict, EpochBasedTrainer, IterBasedTrainer, SyntheticStage1Loader,  # noqa: E402
                                  init_distributed, str2bool)
from lhrs_bot_amd.unibind import build_model  # noqa: E402

logger = logging.getLogger("train")


def parse_option():
    p = ConfigArgumentParser()
    p.add_argument("--batch-size", type=int, default=8, help="batch size for single GPU")
    p.add_argument("--data-path", type=str, default="synthetic")
    p.add_argument("--workers", type=int, default=4)
    p.add_argument("--auto-resume", action="store_true")
    p.add_argument("--resume-path", type=str, default=None)
    p.add_argument("--model-path", type=str, default=None, help="FINAL.pt to initialise the projector from")
    p.add_argument("--accelerator", type=str, default="gpu", choices=["cpu", "gpu", "mps"])
    p.add_argument("--output", type=str, default="work_dir")
    p.add_argument("--seed", type=int, default=322)
    p.add_argument("--gpus", type=int, default=0)
    p.add_argument("--enable-amp", type=str2bool, default=True)
    p.add_argument("--use-checkpoint", action="store_true", help="accepted for flag compatibility; unused (288 GB HBM)")
    p.add_argument("--wandb", type=str2bool, default=False)
    p.add_argument("--accumulation-steps", type=int, default=1, help="gradient accumulation steps")
    p.add_argument("--local_rank", type=int, default=0)
    # knobs of the synthetic run
    p.add_argument("--epoch-len", type=int, default=20, help="iterations per epoch of the synthetic loader")
    p.add_argument("--llama-layers", type=int, default=32)
    p.add_argument("--log-period", type=int, default=5)
    cfg = ConfigDict(p.parse_args(wandb=True))
    return cfg


def main(config):
    model = build_model(config, activate_modal=("rgb", "text"), device=torch.device("cuda", config.local_rank),
                        llama_layers=config.get("llama_layers", 32)).init_random(seed=0)
    model.prepare_for_training(freeze_vision=not config.get("tune_rgb_bk", False), freeze_text=not config.get("lora", {}).get("enable", False),
                               tune_rgb_pooler=config.get("tune_rgb_pooler", True), model_path=config.get("model_path"),
                               tune_im_start=config.get("tune_im_start", False))
    loader = SyntheticStage1Loader(batch_size=config.batch_size, epoch_len=config.epoch_len, seed=config.seed)
    betas = config.get("betas")
    engine = LHRSEngine(model, optimizer=config.get("optimizer", "adanp"), lr=float(config.get("lr", 2e-4)),
                        weight_decay=float(config.get("wd", 0.0)), max_grad_norm=float(config.get("max_grad_norm", 0.3)),
                        betas=tuple(betas) if betas else None,
                        gradient_accumulation_steps=int(config.get("accumulation_steps", 1) or 1))
    common = dict(model=engine, optimizer=engine.optimizer, lr_scheduler=config.get("schedule", {"name": "const"}), data_loader=loader,
                  work_dir=config.output, log_period=config.log_period, save_ckpt_by="iter", accelerator=config.accelerator,
                  enable_amp=config.enable_amp, wandb=config.wandb, gpus=config.gpus, max_num_checkpoints=1,
                  clip_grad_norm=config.get("max_grad_norm", 0.3), is_distributed=config.is_distribute, deepspeed=True)
    if int(config.get("stage", 1)) >= 3:   # main_pretrain_stage3.py:225-231: IterBasedTrainer(max_iters=config.epochs), ckpt_period 100
        trainer = IterBasedTrainer(max_iters=int(config.get("epochs", 1) or 1), ckpt_period=100, **common)
    else:                                  # main_pretrain_stage{1,2}.py: EpochBasedTrainer(max_epochs=config.epochs)
        trainer = EpochBasedTrainer(max_epochs=int(config.get("epochs", 1) or 1), ckpt_period=1000 if int(config.get("stage", 1)) == 1 else 100,
                                    **common)
    trainer.train(load_checkpoint=config.get("resume_path"))
    if config.rank == 0:
        model.custom_save_checkpoint(os.path.join(config.output, "checkpoints"))
        with open(os.path.join(config.output, "history.json"), "w") as f:
            json.dump(trainer.history, f, indent=1)
    return trainer


if __name__ == "__main__":
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")
    config = parse_option()
    config.rank, config.local_rank, config.world_size = init_distributed()
    config.is_distribute = config.world_size > 1
    config.seed = config.seed + config.rank  # main_pretrain_stage1.py:281-287
    torch.manual_seed(config.seed)
    os.makedirs(config.output, exist_ok=True)
    if config.rank == 0:
        with open(os.path.join(config.output, "config.json"), "w") as f:
            json.dump({k: v for k, v in config.items() if not isinstance(v, torch.device)}, f, indent=1, default=str)
    main(config)
