"""The reference's driver-facing helpers, re-built for this engine ("Python surface the entry scripts import").

What `main_pretrain_stage{1,2,3}.py` reach besides the model and the trainers:

  * `build_optimizer(model, config, is_pretrain)`        lhrs/optimizer/build_optimizer.py:76-86 (timm `create_optimizer_v2`)
  * `deepspeed.initialize(config=..., model=..., optimizer=..., model_parameters=...)`   main_pretrain_stage1.py:215-220, fed by
    `build_ds_config` (:28-85)  ->  here `initialize(...)`, same keywords, same 4-tuple, returning the `LHRSEngine`
  * `auto_resume_helper(output_dir)`                     lhrs/CustomTrainer/utils/misc.py:16-30
  * `setup_logger(name, output, log_level, rank, ...)`   lhrs/CustomTrainer/utils/logger.py:26-140
  * `deepspeed_init_distributed()`                       lhrs/CustomTrainer/utils/distribute.py:502-522

north_star forbids dual backends: there is no DeepSpeed underneath.  `initialize` reads the few keys of the DeepSpeed config that mean
something for ONE node of 288 GB GPUs - optimizer type / hyper-parameters, `gradient_accumulation_steps`, `gradient_clipping`, the
16-bit switches - and says what it ignores: ZeRO partitioning and CPU offload exist to fit 80 GB parts; 80 M trainable fp32 parameters x
6 optimizer states are 1.9 GB here and stay replicated.
"""
from __future__ import annotations

import logging
import os
import sys
from typing import Dict, List, Optional, Tuple

import torch

logger = logging.getLogger("train")


# ------------------------------------------------------------------------------------------------ optimizer description
class OptimizerSpec:
    """What `build_optimizer` hands to `initialize`: the optimizer NAME and hyper-parameters plus the decay / no-decay split of the
    trainable tensors (build_optimizer.py:18-38: 1-D tensors and biases do not decay).  The update itself is the fused HIP kernel of
    the engine (`lhrs_adan_step` / `lhrs_adamw_step`); this object only carries the description, shaped like a torch optimizer where the
    reference looks at it (`param_groups[i]["lr" | "weight_decay" | "params"]`)."""

    def __init__(self, opt: str, lr: float, weight_decay: float, decay: List[Tuple[str, torch.Tensor]], no_decay: List[Tuple[str, torch.Tensor]],
                 betas=None, eps: float = 1e-8):
        self.opt, self.lr, self.weight_decay, self.betas, self.eps = opt.lower(), float(lr), float(weight_decay), betas, float(eps)
        self.decay_names, self.no_decay_names = [n for n, _ in decay], [n for n, _ in no_decay]
        self.param_groups = [dict(params=[p for _, p in decay], lr=self.lr, weight_decay=self.weight_decay),
                             dict(params=[p for _, p in no_decay], lr=self.lr, weight_decay=0.0)]
        self.defaults = dict(lr=self.lr, weight_decay=self.weight_decay)


_TIMM_NAMES = {"adanp": "adanp", "adan": "adan", "adanw": "adan", "adamw": "adamw"}  # timm 0.9.12 optim_factory names on this path


def get_param_group(model, is_pretrain: bool = True):
    """-> (decay, no_decay) lists of (name, tensor) over the model's TRAINABLE tensors (get_pretrain_param_groups / set_weight_decay)."""
    decay, no_decay = [], []
    skip = set(model.no_weight_decay()) if hasattr(model, "no_weight_decay") else set()
    for name, p in model.named_parameters():
        (no_decay if (p.dim() == 1 or name.endswith(".bias") or name.split(".")[-1] in skip) else decay).append((name, p))
    return decay, no_decay


def build_optimizer(model, config, is_pretrain: bool = True) -> OptimizerSpec:
    opt = str(config["optimizer"]).lower()
    if opt not in _TIMM_NAMES:
        raise ValueError(f"optimizer {config['optimizer']!r}: this engine implements adanp / adan (timm Adan, no_prox / prox) and adamw")
    decay, no_decay = get_param_group(model, is_pretrain)
    betas = config.get("betas") if hasattr(config, "get") else None
    return OptimizerSpec(_TIMM_NAMES[opt] if opt != "adanw" else "adan", config["lr"], config["wd"], decay, no_decay, betas=tuple(betas) if betas else None)


# ------------------------------------------------------------------------------------------------ deepspeed.initialize-shaped factory
_IGNORED_ZERO_KEYS = ("stage", "sub_group_size", "contiguous_gradients", "overlap_comm", "stage3_gather_16bit_weights_on_model_save",
                      "offload_optimizer", "offload_param")


def initialize(args=None, model=None, optimizer=None, model_parameters=None, training_data=None, lr_scheduler=None, config=None,
               config_params=None, comm_dtype=None, **_unused):
    """`model_engine, optimizer, _, _ = initialize(config=build_ds_config(cfg), model=model, optimizer=opt_or_None, model_parameters=None)`.

    config keys honoured: `optimizer.{type,params.{lr,eps,betas,weight_decay}}` (the AdamW branch of build_ds_config), or an
    `OptimizerSpec` from `build_optimizer` (the Adan branch); `gradient_accumulation_steps`; `gradient_clipping`;
    `train_micro_batch_size_per_gpu` (recorded).  16-bit switches: the engine computes in bf16 with fp32 accumulation whatever they say.
    `fp16.enabled` (the shipped stage-2/3 YAMLs: `fp16: True, bf16: False`, Config/multi_modal_stage2.yaml:75-80) is taken as "16-bit
    compute" and answered with ONE warning naming the deviation - bf16 has fp32's exponent, so DeepSpeed's dynamic loss scaling
    (`initial_scale_power`, `loss_scale_window`) has nothing to scale and is not run; the reference itself forces bf16 for its non-AdamW
    branch (main_pretrain_stage1.py:66-71).  `engine.precision_request` records what was asked for.  `zero_optimization.*`, `zero_force_ds_cpu_optimizer`,
    `zero_allow_untested_optimizer` are accepted and ignored (logged once).  The gradient all-reduce runs in fp32 (320 MB per step over xGMI for
    the projector: cheap) unless DeepSpeed's `communication_data_type` key ("bf16" / "fp16" -> bf16 wire) or `comm_dtype=` asks otherwise."""
    from .engine import LHRSEngine
    cfg: Dict = dict(config if config is not None else (config_params or {}))
    if model is None:
        raise ValueError("initialize() needs model=<UniBind>")
    fp16_req, bf16_req = bool((cfg.get("fp16") or {}).get("enabled")), bool((cfg.get("bf16") or {}).get("enabled"))
    if fp16_req:
        logger.warning("initialize: fp16.enabled=True is run as bf16 compute with fp32 accumulation and fp32 master weights (this engine has one "
                       "16-bit path); DeepSpeed's dynamic loss scaling (initial_scale_power / loss_scale_window) is not needed at bf16 range and is skipped")
    elif not bf16_req and ("fp16" in cfg or "bf16" in cfg):
        logger.warning("initialize: fp16 and bf16 both disabled (fp32 training requested): the engine still computes in bf16 with fp32 accumulation")
    if comm_dtype is None:
        cdt = str(cfg.get("communication_data_type") or "fp32").lower()
        comm_dtype = torch.bfloat16 if cdt in ("bf16", "bfloat16", "fp16", "float16", "half") else torch.float32
    zero = cfg.get("zero_optimization") or {}
    ignored = [f"zero_optimization.{k}" for k in zero if k in _IGNORED_ZERO_KEYS] + [k for k in ("zero_force_ds_cpu_optimizer", "zero_allow_untested_optimizer") if k in cfg]
    if ignored:
        logger.info("initialize: ignoring %s (parameters and optimizer state stay replicated in HBM; nothing is offloaded)", ", ".join(ignored))
    ds_opt = cfg.get("optimizer")
    if optimizer is not None:
        if not isinstance(optimizer, OptimizerSpec):
            raise TypeError("optimizer must come from lhrs.optimizer.build_optimizer (an OptimizerSpec), not a torch optimizer: the "
                            "parameter update is a fused HIP kernel over the engine's flat fp32 master buffer")
        name, lr, wd, betas, eps = optimizer.opt, optimizer.lr, optimizer.weight_decay, optimizer.betas, optimizer.eps
    elif ds_opt:
        if str(ds_opt.get("type", "")).lower() != "adamw":
            raise ValueError(f"DeepSpeed optimizer type {ds_opt.get('type')!r}: build_ds_config only ever asks for AdamW")
        pr = ds_opt.get("params", {})
        name, lr, wd, betas, eps = "adamw", float(pr["lr"]), float(pr.get("weight_decay", 0.0)), tuple(pr.get("betas", (0.9, 0.95))), float(pr.get("eps", 1e-8))
    else:
        raise ValueError("no optimizer: pass build_optimizer(...)'s result or an `optimizer` section in the config")
    engine = LHRSEngine(model, optimizer=name, lr=lr, weight_decay=wd, max_grad_norm=float(cfg.get("gradient_clipping", 0.0) or 0.0), betas=betas,
                        eps=eps, gradient_accumulation_steps=int(cfg.get("gradient_accumulation_steps", 1) or 1), comm_dtype=comm_dtype)
    engine.train_micro_batch_size_per_gpu = cfg.get("train_micro_batch_size_per_gpu")
    engine.precision_request = "fp16" if fp16_req else ("bf16" if bf16_req or not ("fp16" in cfg or "bf16" in cfg) else "fp32")
    return engine, engine.optimizer, None, None


# ------------------------------------------------------------------------------------------------ resume / logging / process group
def auto_resume_helper(output_dir: str) -> Optional[str]:
    """Newest `*.pth` (by mtime) under `<output_dir>/checkpoints`, or None."""
    ckpt_dir = os.path.join(output_dir, "checkpoints")
    names = [n for n in (os.listdir(ckpt_dir) if os.path.isdir(ckpt_dir) else []) if n.endswith("pth")]
    logger.info("All checkpoints founded in %s: %s", ckpt_dir, names)
    if not names:
        return None
    latest = max((os.path.join(ckpt_dir, n) for n in names), key=os.path.getmtime)
    logger.info("The latest checkpoint founded: %s", latest)
    return latest


_LOGGERS: Dict[str, bool] = {}


def setup_logger(name: Optional[str] = None, output: Optional[str] = None, log_level: int = logging.DEBUG, rank: int = 0, color: bool = True,
                 rank_zero_output: bool = True) -> logging.Logger:
    """Console handler on rank 0 (+ `<output>/log.txt`, or `<output>` itself when it ends in .txt / .log; other ranks write
    `log.txt.rank<r>` only when rank_zero_output is off).  Idempotent per logger name.  `color` is accepted; termcolor is not required."""
    lg = logging.getLogger(name)
    if _LOGGERS.get(name or ""):
        return lg
    lg.setLevel(log_level)
    lg.propagate = False
    fmt = logging.Formatter("[%(asctime)s %(name)s %(levelname)s]: %(message)s", datefmt="%m/%d %H:%M:%S")
    if rank == 0:
        h = logging.StreamHandler(stream=sys.stdout)
        h.setLevel(log_level)
        h.setFormatter(fmt)
        lg.addHandler(h)
    if output is not None and (rank == 0 or not rank_zero_output):
        path = output if output.endswith((".txt", ".log")) else os.path.join(output, "log.txt")
        if rank > 0:
            path += f".rank{rank}"
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        fh = logging.FileHandler(path)
        fh.setLevel(log_level)
        fh.setFormatter(fmt)
        lg.addHandler(fh)
    _LOGGERS[name or ""] = True
    return lg


def init_distributed(auto: bool = False) -> Tuple[int, int, int]:
    """lhrs/CustomTrainer/utils/distribute.py:525-560, the evaluation scripts' variant: same environment contract and return value as
    `deepspeed_init_distributed`; `auto` (pick a free MASTER_PORT) is accepted - torchrun already hands every rank a free port."""
    return deepspeed_init_distributed()


def deepspeed_init_distributed() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the launcher's environment (torchrun / `deepspeed --num_gpus` export RANK, WORLD_SIZE,
    LOCAL_RANK; SLURM_PROCID / SLURM_NTASKS otherwise); one process per GPU, process group = RCCL ("nccl") over xGMI, barrier, device
    bound.  Without a launcher: (0, 0, 1), not distributed."""
    env = os.environ
    if "RANK" in env and "WORLD_SIZE" in env:
        rank, world, local = int(env["RANK"]), int(env["WORLD_SIZE"]), int(env.get("LOCAL_RANK", "0"))
    elif "SLURM_PROCID" in env:
        rank, world = int(env["SLURM_PROCID"]), int(env["SLURM_NTASKS"])
        local = rank % max(1, torch.cuda.device_count())
    else:
        print("Not using distributed mode.")
        return 0, 0, 1
    print(f"| distributed init (rank {rank})", flush=True)
    share = env.get("LHRS_SHARE_GPU") == "1" and world > torch.cuda.device_count()  # plumbing tests on a 1-GPU box
    if torch.cuda.is_available():
        torch.cuda.set_device(0 if share else local)
    if not torch.distributed.is_initialized():
        env.setdefault("MASTER_ADDR", "127.0.0.1")
        env.setdefault("MASTER_PORT", "29500")
        backend = "nccl" if torch.cuda.is_available() and not share else "gloo"
        kw = dict(device_id=torch.device("cuda", local)) if backend == "nccl" else {}
        torch.distributed.init_process_group(backend, rank=rank, world_size=world, **kw)
    torch.distributed.barrier()
    return rank, local, world
