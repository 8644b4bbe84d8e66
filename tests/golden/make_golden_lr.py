"""LR-schedule golden: drive the REFERENCE's own CosineAnnealingLrUpdaterHook (lhrs/CustomTrainer/hook/lr_scheduler_hook.py)
exactly as EpochBasedTrainer.get_specific_hooks builds it (EpochBasedTrainer.py:71-80) with the stage-1 YAML values, and
store lr(it).  Build-container only; the committed lr_schedule.npz is what the tests read."""
import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
pkg = types.ModuleType("refhook")
pkg.__path__ = [f"{REF}/lhrs/CustomTrainer/hook"]
sys.modules["refhook"] = pkg
for _n in ("lhrs", "lhrs.CustomTrainer"):  # HookBase.__init__ does `import lhrs.CustomTrainer` only for a type annotation
    sys.modules[_n] = types.ModuleType(_n)
sys.modules["lhrs"].CustomTrainer = sys.modules["lhrs.CustomTrainer"]
mod = importlib.import_module("refhook.lr_scheduler_hook")
cfg = yaml.safe_load(open(f"{REF}/Config/multi_modal_stage1.yaml"))
sch = cfg["schedule"]


class Opt:
    def __init__(self, lr):
        self.param_groups = [dict(lr=lr), dict(lr=lr)]


class Trainer:
    pass


out = {}
for max_iters in (1000, 20000):
    tr = Trainer()
    tr.optimizer = Opt(cfg["lr"])
    tr.max_iters = max_iters
    tr.cur_iter = 0
    hook = mod.CosineAnnealingLrUpdaterHook(by_epoch=False, warmup=sch["warmup_method"], warmup_ratio=sch["warmup_factor"],
                                            warmup_by_epoch=False, min_lr=sch["min_lr"], warmup_iters=sch["warmup_epochs"])
    hook.trainer = tr
    hook.before_train()
    its = list(range(0, 400, 7)) + list(range(400, max_iters, max(1, max_iters // 50)))
    lrs = []
    for it in its:
        tr.cur_iter = it
        hook.before_iter()
        lrs.append(tr.optimizer.param_groups[0]["lr"])
    out[f"its_{max_iters}"] = np.array(its)
    out[f"lrs_{max_iters}"] = np.array(lrs, dtype=np.float64)
np.savez(os.path.join(HERE, "lr_schedule.npz"), base_lr=np.array(cfg["lr"]), min_lr=np.array(sch["min_lr"]),
         warmup_iters=np.array(sch["warmup_epochs"]), warmup_ratio=np.array(sch["warmup_factor"]), **out)
print("lr golden:", out["lrs_1000"][:3], out["lrs_1000"][-2:])

# ---- config-parser golden: the reference's own ConfigArgumentParser on the shipped stage-1 YAML + a CLI overlay
spec = importlib.util.spec_from_file_location("ref_config_parser", f"{REF}/lhrs/CustomTrainer/utils/config_parser.py")
cp = importlib.util.module_from_spec(spec)
spec.loader.exec_module(cp)
import json

def build(parser_cls):
    p = parser_cls()
    p.add_argument("--batch-size", type=int, default=8)
    p.add_argument("--lr", type=float, default=None)
    p.add_argument("--output", type=str, default="out")
    return p

argv = ["-c", f"{REF}/Config/multi_modal_stage1.yaml", "--batch-size", "4", "--output", "xyz"]
cli_wins = build(cp.ConfigArgumentParser).parse_args(wandb=True, args=argv)
yaml_wins = build(cp.ConfigArgumentParser).parse_args(wandb=False, args=argv)
json.dump({"yaml": yaml.safe_load(open(f"{REF}/Config/multi_modal_stage1.yaml")), "argv_tail": argv[2:], "cli_wins": cli_wins,
           "yaml_wins": yaml_wins}, open(os.path.join(HERE, "config_parser.json"), "w"), indent=1, default=str)
print("config golden keys:", len(cli_wins), cli_wins["batch_size"], cli_wins["lr"], yaml_wins["lr"])
