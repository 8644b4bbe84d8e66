#!/usr/bin/env python
"""Stage-1 pretraining on the gfx950 engine: the reference's `main_pretrain_stage1.py` call sequence over the `lhrs.*` surface.

    python -m torch.distributed.run --nproc-per-node 8 main_pretrain_stage1.py -c Config/multi_modal_stage1.yaml \\
        --batch-size 8 --workers 4 --data-path <dir of NAME_Image/ + NAME.json> --output <dir> --accelerator gpu --enable-amp True

Same imports, same order of calls as /root/reference main_pretrain_stage1.py:13-23, 178-258 -
`build_model -> build_loader -> prepare_for_training -> build_optimizer -> initialize -> EpochBasedTrainer -> auto_resume_helper ->
train -> custom_save_checkpoint` - with ONE substitution: `deepspeed.initialize` is `lhrs.CustomTrainer.initialize` (same keywords and
return tuple).  The reference's own script runs on this engine after that one-line change.

Offline additions (no dataset / weights ship with the repo): `--data-path synthetic` feeds the stage-1 batch contract from
`SyntheticStage1Loader`; weight paths that do not exist leave the towers random-initialised with a warning (`load_base_weights`);
`--llama-layers N` truncates the decoder for smoke runs.  Stages 2 and 3 (main_pretrain_stage{2,3}.py) share this file: they differ
from stage 1 by the YAML (`stage`, `lora`, `bits`, `optimizer`, `betas`), the checkpoint period and - stage 3 - the trainer class.
"""
import json
import logging
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lhrs.CustomTrainer import deepspeed_init_distributed, initialize  # noqa: E402
from lhrs.CustomTrainer.EpochBasedTrainer import EpochBasedTrainer  # noqa: E402
from lhrs.CustomTrainer.IterBasedTrainer import IterBasedTrainer  # noqa: E402
from lhrs.CustomTrainer.utils import ConfigArgumentParser, ConfigDict, auto_resume_helper, setup_logger, str2bool  # noqa: E402
from lhrs.Dataset.build_loader import build_loader  # noqa: E402
from lhrs.models import build_model  # noqa: E402
from lhrs.optimizer import build_optimizer  # noqa: E402

logger = logging.getLogger("train")


_ZERO2 = {"adamw": dict(stage=2, sub_group_size=1e9, contiguous_gradients=True, overlap_comm=True, stage3_gather_16bit_weights_on_model_save=True),
          "other": dict(stage=2, offload_optimizer={"device": "cpu"}, offload_param={"device": "cpu"})}


def build_ds_config(config):
    """The engine configuration in DeepSpeed's vocabulary, key for key what main_pretrain_stage1.py:28-85 hands to `deepspeed.initialize`
    (pinned by tests/golden/yaml_surface.json): AdamW is described inside the dict with the YAML's fp16 / bf16 switches passed through as
    they are, any other optimizer arrives as the object `build_optimizer` made and forces bf16 autocast; accumulation and clipping ride
    along.  The ZeRO / offload sections are emitted too - `initialize` names them in its "ignored" log line instead of this function
    deciding what the engine gets to see."""
    adamw = str(config.optimizer).lower() == "adamw"
    ds = {"train_micro_batch_size_per_gpu": config.batch_size,
          "gradient_accumulation_steps": config.get("accumulation_steps", 1),
          "gradient_clipping": config.max_grad_norm,
          "zero_optimization": dict(_ZERO2["adamw" if adamw else "other"])}
    if adamw:
        ds["optimizer"] = {"type": "AdamW", "params": {"lr": config.lr, "eps": 1e-8, "betas": (0.9, 0.95), "weight_decay": config.wd}}
        ds["fp16"] = {"enabled": bool(config.get("fp16", False)), "auto_cast": False, "initial_scale_power": 16, "loss_scale_window": 500}
        ds["bf16"] = {"enabled": bool(config.get("bf16", False)), "auto_cast": False}
    else:
        ds["bf16"] = {"enabled": True, "auto_cast": True}
        ds.update(zero_force_ds_cpu_optimizer=False, zero_allow_untested_optimizer=True)
    return ds


def parse_option(args=None):
    p = ConfigArgumentParser()
    p.add_argument("--opts", default=None, nargs="+", help="Modify config options by adding 'KEY VALUE' pairs.")
    p.add_argument("--batch-size", type=int, default=8, help="batch size for single GPU")
    p.add_argument("--data-path", type=str, default="synthetic", help="path to dataset ('synthetic': generated stage-1 batches)")
    p.add_argument("--eval-data-path", type=str, help="path to evaluate dataset")
    p.add_argument("--workers", type=int, default=8, help="workers of dataloader")
    p.add_argument("--auto-resume", action="store_true", help="resume from checkpoint")
    p.add_argument("--resume-path", type=str, default=None, help="resume checkpoint path")
    p.add_argument("--model-path", type=str, default=None, help="pretrained checkpoint path for model (maybe stage 1)")
    p.add_argument("--accumulation-steps", type=int, default=1, help="gradient accumulation steps")
    p.add_argument("--use-checkpoint", action="store_true", help="accepted; activations fit in 288 GB, nothing is recomputed")
    p.add_argument("--enable-amp", type=str2bool, default=False, help="mixed precision")
    p.add_argument("--output", default="output", type=str, metavar="PATH", help="root of output folder")
    p.add_argument("--seed", type=int, default=322, help="random seed")
    p.add_argument("--gpus", type=int, default=0, help="gpus ID")
    p.add_argument("--inf_sampler", type=str2bool, default=False, help="infinite sampler (iteration-based training)")
    p.add_argument("--torch-compile", type=str2bool, default=False, help="accepted and ignored: the engine is hand-written HIP")
    p.add_argument("--wandb", type=str2bool, default=False, help="wandb logger (not available offline: must stay False)")
    for flag, default in (("--entity", "pumpkinn"), ("--project", "MultiModal"), ("--job-type", "vlm_test"), ("--name", "first_run"), ("--notes", None)):
        p.add_argument(flag, type=str, default=default, help="wandb run metadata (accepted for flag compatibility; wandb is not available offline)")
    p.add_argument("--tags", type=str, default="MultiModal", nargs="+", help="wandb tags (accepted, unused)")
    p.add_argument("--accelerator", default="gpu", type=str, choices=["cpu", "gpu", "mps"], help="accelerator")
    p.add_argument("--local_rank", type=int)
    # knobs of this engine's offline runs
    p.add_argument("--epoch-len", type=int, default=20, help="iterations per epoch of the synthetic loader")
    p.add_argument("--llama-layers", type=int, default=32, help="decoder layers to build (smoke runs)")
    p.add_argument("--log-period", type=int, default=1)
    config = ConfigDict(p.parse_args(wandb=True, args=args))
    opts = config.get("opts") or []
    if len(opts) % 2:
        p.error("--opts takes KEY VALUE pairs")
    import yaml
    for k, v in zip(opts[0::2], opts[1::2]):
        config[k] = yaml.safe_load(v)
    return config


def main(config):
    stage = int(config.get("stage", 1))
    logger.info("Creating model")
    model = build_model(config, activate_modal=("rgb", "text"))

    logger.info("Building Dataset")
    if str(config.get("data_path", "synthetic")) == "synthetic":
        from lhrs_bot_amd.trainer import SyntheticStage1Loader
        logger.warning("--data-path synthetic: random stage-1 batches (ids, 224x224 noise); no corpus is read")
        data_loader_train = SyntheticStage1Loader(batch_size=config.batch_size, epoch_len=config.get("epoch_len", 20), seed=config.get("seed", 322))
    else:
        data_loader_train = build_loader(config, mode="pretrain", tokenizer=model.text.tokenizer, prompt_type=config.get("prompt_template", "plain"))

    compute_dtype = torch.float16 if config.get("fp16", False) else (torch.bfloat16 if config.get("bf16", True) else torch.float32)
    model.prepare_for_training(freeze_vision=not config.get("tune_rgb_bk", False), freeze_text=not config.get("lora", {}).get("enable", False),
                               tune_rgb_pooler=config.get("tune_rgb_pooler", True), model_path=config.get("model_path"),
                               tune_im_start=config.get("tune_im_start", False), compute_dtype=compute_dtype)

    optimizer = None if str(config.optimizer).lower() == "adamw" else build_optimizer(model, config, is_pretrain=True)
    model_engine, optimizer, _, _ = initialize(config=build_ds_config(config), model=model, optimizer=optimizer, model_parameters=None)

    common = dict(model=model_engine, optimizer=optimizer, lr_scheduler=config.get("schedule", {"name": "const"}), data_loader=data_loader_train,
                  work_dir=config.output, log_period=config.get("log_period", 1), save_ckpt_by="iter",
                  ckpt_period=1000 if stage == 1 else 100, accelerator=config.get("accelerator", "gpu"), enable_amp=config.get("enable_amp", False),
                  wandb=config.get("wandb", False), gpus=0, max_num_checkpoints=1, clip_grad_norm=config.max_grad_norm,
                  is_distributed=config.get("is_distribute", False), torch_compile=config.get("torch_compile", False), dtype=compute_dtype,
                  deepspeed=True)
    if stage >= 3:  # main_pretrain_stage3.py:225-231
        trainer = IterBasedTrainer(max_iters=int(config.epochs), **common)
    else:
        trainer = EpochBasedTrainer(max_epochs=int(config.epochs), **common)

    if config.get("auto_resume", False):
        resume_file = auto_resume_helper(config.output)
        if resume_file:
            if config.get("resume_path") is not None:
                logger.warning(f"auto-resume changing resume file from {config.resume_path} to {resume_file}")
            config.resume_path = resume_file
            logger.info(f"auto resuming from {resume_file}")
        else:
            logger.info(f"no checkpoint found in {config.output}/checkpoint, ignoring auto resume")

    trainer.train(load_checkpoint=config.get("resume_path"))

    if config.get("local_rank", 0) in (0, -1, None) or config.get("rank", 0) == 0:
        model.custom_save_checkpoint(os.path.join(config.output, "checkpoints"))  # writes FINAL.pt (+ TextLoRA/) itself
        with open(os.path.join(config.output, "history.json"), "w") as f:
            json.dump(trainer.history, f, indent=1)
    return trainer


def run(default_stage: int):
    config = parse_option()
    config.setdefault("stage", default_stage)
    if config.get("wandb", False):
        raise SystemExit("--wandb True: wandb is not available offline")
    config.rank, config.local_rank, config.world_size = deepspeed_init_distributed()
    config.is_distribute = config.world_size > 1
    setup_logger("train", output=config.output, rank=config.rank)
    os.makedirs(os.path.join(config.output, "checkpoints"), exist_ok=True)
    seed = config.seed + dist.get_rank() if config.is_distribute else config.seed  # main_pretrain_stage1.py:281-287
    torch.manual_seed(seed)
    np.random.seed(seed)
    import random
    random.seed(seed)
    if config.rank == 0:
        path = os.path.join(config.output, "config.json")
        with open(path, "w") as f:
            json.dump(dict(config), f, indent=4, default=str)
        logger.info(f"Full config saved to {path}")
    return main(config)


if __name__ == "__main__":
    run(1)
