"""Hook-driven training loop + config surface of the reference, driving the gfx950 engine.

Mirrors /root/reference lhrs/CustomTrainer: `Trainer.train / train_on_iter / _log_iter_metrics` (trainer.py:390-500),
`EpochBasedTrainer` (EpochBasedTrainer.py:56-109), `IterBasedTrainer` (IterBasedTrainer.py:49-91), the hook protocol
(hook/hookbase.py:35-64), `DeepSpeedHook.after_iter` (hook/deepspeed_hook.py:4-19), `CosineAnnealingLrUpdaterHook`
(hook/lr_scheduler_hook.py), `IterCheckpointerHook` (hook/checkpoint_hook.py:60-70), `LoggerHook` (hook/logger_hook.py),
and `ConfigArgumentParser` (utils/config_parser.py:13-54).  Same hook order as the reference realises:
[checkpoint, engine-step, lr, distributed, logger].

Deliberate change: the reference does `loss.cpu().item()` + a pickled gloo gather EVERY iteration,
which stalls the device; here the loss stays a device scalar and is read (and averaged over ranks with one tiny
all-reduce) only every `log_period` iterations.
"""
from __future__ import annotations

import argparse
import json
import logging
import os
import time
from typing import Dict, List, Optional

import torch
import yaml

from .engine import LHRSEngine, cosine_warmup_lr

logger = logging.getLogger("train")


# ------------------------------------------------------------------------------------------------ config surface
class ConfigDict(dict):
    """Attribute-access dict standing in for ml_collections.ConfigDict (main_pretrain_stage1.py:172-173)."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = ConfigDict(v) if isinstance(v, dict) and not isinstance(v, ConfigDict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = dict.__setitem__


class ConfigArgumentParser(argparse.ArgumentParser):
    """`-c/--config FILE` YAML merged with the CLI; with wandb=True the CLI overrides the YAML, else the YAML overrides the
    CLI (config_parser.py:39-54)."""

    def __init__(self, *args, **kwargs):
        self.config_parser = argparse.ArgumentParser(add_help=False)
        self.config_parser.add_argument("-c", "--config", default=None, metavar="FILE", help="where to load YAML configuration")
        self.option_names: List[str] = []
        super().__init__(*args, parents=[self.config_parser], formatter_class=argparse.RawDescriptionHelpFormatter, **kwargs)

    def add_argument(self, *args, **kwargs):
        arg = super().add_argument(*args, **kwargs)
        self.option_names.append(arg.dest)
        return arg

    def parse_args(self, wandb=False, args=None):
        res, remaining = self.config_parser.parse_known_args(args)
        if res.config is None:
            return vars(super().parse_args(remaining))
        with open(res.config, "r") as f:
            config_vars = yaml.safe_load(f)
        namespace = vars(super().parse_args(remaining))
        if wandb:
            config_vars.update(namespace)
            return config_vars
        namespace.update(config_vars)
        return namespace


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


def init_distributed():
    """deepspeed_init_distributed (utils/distribute.py:502-522): RCCL process group from the launcher's env."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    if world > 1 and not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "nccl" if torch.cuda.is_available() else "gloo"
        torch.distributed.init_process_group(backend, rank=rank, world_size=world)
        torch.distributed.barrier()
    return rank, local_rank, world


# ------------------------------------------------------------------------------------------------ synthetic stage-1 data
class SyntheticStage1Loader:
    """Batches with the stage-1 contract of DataCollatorForSupervisedDataset (lhrs/Dataset/cap_dataset.py:775-810) and
    preprocess_plain (:955-974): ids = [BOS, <image>=-200, caption..., pad=0], labels = ids with the first two and the
    padding masked to -100, attention_mask = ids != pad, rgb [B,3,224,224].  Lengths vary so the ragged path is used."""

    def __init__(self, batch_size=8, epoch_len=100, caption_tokens=(16, 128), seed=322, vocab=32000, device="cpu"):
        self.bs, self.epoch_len, self.cap, self.seed, self.vocab, self.device = batch_size, epoch_len, caption_tokens, seed, vocab, device
        self.sampler = None

    def __len__(self):
        return self.epoch_len

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        for _ in range(self.epoch_len):
            lens = torch.randint(self.cap[0], self.cap[1] + 1, (self.bs,), generator=g)
            T = int(lens.max()) + 2
            ids = torch.zeros(self.bs, T, dtype=torch.int64)
            for b in range(self.bs):
                n = int(lens[b])
                ids[b, 0], ids[b, 1] = 1, -200
                ids[b, 2:2 + n] = torch.randint(3, self.vocab, (n,), generator=g)
            labels = ids.clone()
            labels[:, :2] = -100
            labels[ids == 0] = -100
            yield {"rgb": torch.randn(self.bs, 3, 224, 224, generator=g).to(self.device), "input_ids": ids.to(self.device),
                   "labels": labels.to(self.device), "attention_mask": ids.ne(0).to(self.device)}


# ------------------------------------------------------------------------------------------------ hooks
class HookBase:
    trainer: "Trainer" = None
    priority = 50

    def before_train(self): pass
    def after_train(self): pass
    def before_epoch(self): pass
    def after_epoch(self): pass
    def before_iter(self): pass
    def after_iter(self): pass

    @property
    def class_name(self):
        return self.__class__.__name__

    def state_dict(self):
        return {}

    def load_state_dict(self, sd):
        pass


class EngineStepHook(HookBase):
    """DeepSpeedHook.after_iter: engine.backward(total_loss); engine.step(); publish the grad norm."""

    def after_iter(self):
        t = self.trainer
        t.model.backward(t.loss_dict["total_loss"])
        t.model.step()
        t._last_grad_norm_sq = t.model.optimizer._global_grad_norm


class CosineAnnealingLrUpdaterHook(HookBase):
    def __init__(self, by_epoch=False, warmup="linear", warmup_iters=0, warmup_ratio=0.1, min_lr=0.0):
        assert not by_epoch, "the reference builds it with by_epoch=False (EpochBasedTrainer.py:73-80)"
        self.warmup, self.warmup_iters, self.warmup_ratio, self.min_lr = warmup, warmup_iters, warmup_ratio, min_lr

    def before_train(self):
        for g in self.trainer.optimizer.param_groups:
            g.setdefault("initial_lr", g["lr"])

    def before_iter(self):
        t = self.trainer
        for g in t.optimizer.param_groups:
            g["lr"] = cosine_warmup_lr(t.cur_iter, g["initial_lr"], t.max_iters, self.min_lr, self.warmup_iters, self.warmup_ratio,
                                       self.warmup)


class FixedLrUpdaterHook(HookBase):
    pass


class DistributedHook(HookBase):
    def before_epoch(self):  # hook/distributed_hook.py:4-13
        s = getattr(self.trainer.data_loader, "sampler", None)
        if s is not None and hasattr(s, "set_epoch"):
            s.set_epoch(self.trainer.epoch)


class IterCheckpointerHook(HookBase):
    def __init__(self, period: int, max_to_keep: Optional[int] = None):
        self.period, self.max_to_keep, self.recent = period, max_to_keep, []

    def after_iter(self):
        t = self.trainer
        if self.period and (t.cur_iter + 1) % self.period == 0:
            name = f"iter_{t.cur_iter}"
            t.save_checkpoint(name)
            self.recent.append(name)
            if self.max_to_keep and len(self.recent) > self.max_to_keep:
                old = self.recent.pop(0)
                p = os.path.join(t.ckpt_dir, old + ".pth")
                if t.rank == 0 and os.path.exists(p):
                    os.remove(p)


class LoggerHook(HookBase):
    def __init__(self, period=50):
        self.period = period

    def after_iter(self):
        t = self.trainer
        if (t.cur_iter + 1) % self.period and t.cur_iter + 1 != t.max_iters:
            return
        loss = t.loss_dict["total_loss"].detach().float().clone()
        if t.world > 1:
            torch.distributed.all_reduce(loss)
            loss /= t.world
        now = time.perf_counter()
        dt = (now - t._log_t0) / max(1, t.cur_iter + 1 - t._log_it0)
        t._log_t0, t._log_it0 = now, t.cur_iter + 1
        gn = t.model.grad_norm() if hasattr(t.model, "grad_norm") else float("nan")
        rec = dict(iter=t.cur_iter + 1, epoch=t.epoch, loss=float(loss.item()), lr=t.lr, iter_time=dt, grad_norm=gn,
                   samples_per_s=t.world * t.samples_per_iter / dt if dt > 0 else 0.0)
        t.history.append(rec)
        if t.rank == 0:
            logger.info("Epoch [%d] Iter [%d/%d] loss %.4f lr %.3e grad_norm %.3f iter_time %.4fs (%.1f samples/s)", rec["epoch"],
                        rec["iter"], t.max_iters, rec["loss"], rec["lr"], rec["grad_norm"], dt, rec["samples_per_s"])


# ------------------------------------------------------------------------------------------------ trainers
class Trainer:
    def __init__(self, model: LHRSEngine, optimizer=None, lr_scheduler=None, data_loader=None, work_dir="work_dir", log_period=50,
                 save_ckpt_by="iter", ckpt_period=1000, max_num_checkpoints=None, clip_grad_norm=0.0, deepspeed=True,
                 accelerator="gpu", enable_amp=True, wandb=False, gpus=0, is_distributed=False, torch_compile=False, dtype=None,
                 **_unused):
        if not isinstance(model, LHRSEngine):
            raise TypeError("model must be the LHRSEngine returned by lhrs_bot_amd.engine (the reference passes the DeepSpeed engine)")
        self.model, self.optimizer = model, model.optimizer
        self.lr_scheduler = ConfigDict(lr_scheduler or {"name": "const"})
        self.data_loader, self.work_dir, self.log_period = data_loader, work_dir, log_period
        self.ckpt_period, self.max_num_checkpoints = ckpt_period, max_num_checkpoints
        self.ckpt_dir = os.path.join(work_dir, "checkpoints")
        self.rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0
        self.world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
        self.device = model.pool.device
        self.epoch = self.start_epoch = self.inner_iter = 0
        self.history: List[Dict] = []
        self._hooks: List[HookBase] = []
        self.loss_dict = None
        self.samples_per_iter = getattr(data_loader, "bs", 0) or getattr(data_loader, "batch_size", 0) or 0

    # -- properties the hooks use
    @property
    def lr(self):
        return self.optimizer.param_groups[0]["lr"]

    @property
    def epoch_len(self):
        return len(self.data_loader)

    @property
    def cur_iter(self):
        return self.epoch * self.epoch_len + self.inner_iter

    def register_hook(self, hooks):
        for h in hooks:
            if h is None:
                continue
            h.trainer = self
            if self._hooks and isinstance(self._hooks[-1], LoggerHook):
                self._hooks.insert(len(self._hooks) - 1, h)  # keep the logger last (trainer.py:238-253)
            else:
                self._hooks.append(h)

    def _call_hooks(self, name):
        for h in self._hooks:
            getattr(h, name)()

    def _build_lr_hook(self):
        s = self.lr_scheduler
        if s.get("name", "const") == "cosine":
            return CosineAnnealingLrUpdaterHook(by_epoch=False, warmup=s.get("warmup_method", "linear"),
                                                warmup_ratio=s.get("warmup_factor", 0.1), min_lr=s.get("min_lr", 0.0),
                                                warmup_iters=s.get("warmup_epochs", 0))
        if s.get("name") == "const":
            return FixedLrUpdaterHook()
        raise NotImplementedError(f"Unsupported lr scheduler: {s.get('name')}")

    def _prepare_for_training(self):
        os.makedirs(self.ckpt_dir, exist_ok=True)
        self.register_hook([IterCheckpointerHook(self.ckpt_period, self.max_num_checkpoints), LoggerHook(self.log_period)])
        self.register_hook([EngineStepHook(), self._build_lr_hook(), DistributedHook()])
        self._data_iter = iter(self.data_loader)
        self._log_t0, self._log_it0 = time.perf_counter(), self.cur_iter

    def put_input_to_device(self, batch):
        """trainer.py:400-417.  `rgb` may be a list of uint8 [H,W,3] pictures of different sizes (decoded by the DataLoader workers, see
        lhrs_bot_amd/datasets.py): each goes up as it is, the model's `_pixels` runs the CLIP transform on the device."""
        # The small INTEGER tensors (input_ids, labels, attention_mask) stay where the DataLoader delivered them - in host memory: the engine does its integer
        # bookkeeping (splice lengths, supervised rows, targets) on the host and uploads only what the kernels read.  Moving them up here (as the reference's
        # put_input_to_device does for its torch modules) would cost a device -> host copy and a stream synchronisation at the start of EVERY step
        # (TextModal._ints_to_host) - and a drained queue at every step start is exactly the trigger of the shared-device NaN of DESIGN.md 6.
        def move(v):
            if not torch.is_tensor(v) or (not v.is_floating_point() and v.dtype != torch.uint8):
                return v
            return v.to(self.device, non_blocking=True)
        return {k: [move(x) for x in v] if isinstance(v, (list, tuple)) else move(v) for k, v in batch.items()}

    def train_on_iter(self):
        batch = self.put_input_to_device(next(self._data_iter))
        out = self.model(batch)
        self.loss_dict = {"total_loss": out} if torch.is_tensor(out) else out

    # -- checkpoint (trainer.py:261-384): engine state + client_state
    def save_checkpoint(self, name: str):
        if self.rank != 0:
            return
        hooks = {h.class_name: h.state_dict() for h in self._hooks if h.state_dict()}
        state = dict(cur_stat=self.cur_iter + 1, num_gpus=self.world, hooks=hooks, engine=self.model.state_dict())
        torch.save(state, os.path.join(self.ckpt_dir, name + ".pth"))

    def load_checkpoint(self, path: str):
        state = torch.load(path, map_location="cpu")
        assert state["num_gpus"] == self.world, f"checkpoint was written with {state['num_gpus']} GPUs, running {self.world}"
        self.model.load_state_dict(state["engine"])
        self.load_cur_stat(state["cur_stat"])
        for _ in range(self.inner_iter):  # fast-forward the data iterator (trainer.py:356-358)
            next(self._data_iter)

    def train(self, load_checkpoint: Optional[str] = None):
        self._prepare_for_training()
        if load_checkpoint is not None:
            self.load_checkpoint(load_checkpoint)
        self._call_hooks("before_train")
        self.sub_classes_train()
        self._call_hooks("after_train")


class EpochBasedTrainer(Trainer):
    def __init__(self, max_epochs: int, **kw):
        super().__init__(**kw)
        self.max_epochs = max_epochs

    @property
    def max_iters(self):
        return self.max_epochs * self.epoch_len

    def load_cur_stat(self, value):
        self.epoch = self.start_epoch = value // self.epoch_len
        self.inner_iter = value % self.epoch_len

    def sub_classes_train(self):
        for self.epoch in range(self.start_epoch, self.max_epochs):
            self.model.train()  # EpochBasedTrainer.py:91 - every epoch; with adapters loaded from TextLoRA/ this re-arms lora_dropout
            self._call_hooks("before_epoch")
            for self.inner_iter in range(self.inner_iter, self.epoch_len):
                self._call_hooks("before_iter")
                self.train_on_iter()
                self._call_hooks("after_iter")
            self.inner_iter = 0
            self._data_iter = iter(self.data_loader)
            self._call_hooks("after_epoch")
        self.epoch = self.max_epochs - 1
        self.inner_iter = self.epoch_len - 1


class IterBasedTrainer(Trainer):
    def __init__(self, max_iters: int, **kw):
        super().__init__(**kw)
        self._max_iters = max_iters

    @property
    def max_iters(self):
        return self._max_iters

    @property
    def cur_iter(self):
        return self.inner_iter

    def load_cur_stat(self, value):
        self.inner_iter = value

    def sub_classes_train(self):
        self.model.train()  # IterBasedTrainer.py:87
        self._call_hooks("before_epoch")
        for self.inner_iter in range(self.inner_iter, self._max_iters):
            self._call_hooks("before_iter")
            try:
                self.train_on_iter()
            except StopIteration:
                self._data_iter = iter(self.data_loader)
                self.train_on_iter()
            self._call_hooks("after_iter")
        self._call_hooks("after_epoch")


def dump_history(trainer: Trainer, path: str):
    with open(path, "w") as f:
        json.dump(trainer.history, f, indent=1)
