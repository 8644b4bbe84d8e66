"""CPU: the reference's import block and call sequence replayed against this repo's `lhrs` package.

tests/golden/import_surface.json holds, read from the reference's sources with `ast` (make_golden_surface.py): every `from lhrs... import`
of main_pretrain_stage{1,2,3}.py / cli_qa.py / main_cls.py / main_vqa.py / main_vg.py / main_bench_gen.py, every call inside their `main()` with its positional count and keyword names, and the
parameter names of the functions those calls land on.  Here: (1) every imported name resolves, (2) every call binds to this repo's
callable, (3) every reference parameter is accepted under the same name.  What the calls COMPUTE is the GPU suites' business
(tests/test_surface_gpu.py runs the same sequence on a tiny model)."""
import importlib
import inspect
import json
import os

import pytest

G = os.path.join(os.path.dirname(__file__), "golden")
Z = json.load(open(os.path.join(G, "import_surface.json")))


def _resolve(script, callee):
    """The object in THIS repo a call of the reference script lands on, or None for calls outside the lhrs surface (stdlib, torch,
    logging, the script's own helpers)."""
    from lhrs.CustomTrainer import initialize
    from lhrs_bot_amd.conversation import Conversation
    from lhrs_bot_amd.data import CLIPImageProcessorHIP
    from lhrs_bot_amd.trainer import Trainer
    from lhrs_bot_amd.unibind import UniBind
    for imp in Z["scripts"][script]["imports"]:
        if callee in imp["names"]:
            return getattr(importlib.import_module(imp["module"]), callee)
    head, _, tail = callee.partition(".")
    table = {"model": UniBind, "trainer": Trainer, "conv": Conversation, "default_conversation": Conversation, "dummy_conv": Conversation}
    if callee == "deepspeed.initialize":
        return initialize  # the ONE substitution: `from lhrs.CustomTrainer import initialize`
    if callee in ("vision_processor", "vis_transform"):
        return CLIPImageProcessorHIP.__call__
    if head in table and tail and "." not in tail:
        return getattr(table[head], tail, "MISSING")
    return None


# reached only on branches this engine's objects never take: `config.hf_model` (a HF-hub model class the reference also ships) and the
# `hasattr(model, "custom_load_state_dict")` fallback for plain nn.Modules
NOT_ON_PATH = {"model.get_image_processor", "model.load_state_dict"}


@pytest.mark.parametrize("script", sorted(Z["scripts"]))
def test_import_block_resolves(script):
    for imp in Z["scripts"][script]["imports"]:
        mod = importlib.import_module(imp["module"])
        for name in imp["names"]:
            assert hasattr(mod, name), f"{script}: from {imp['module']} import {name}"


@pytest.mark.parametrize("script", sorted(Z["scripts"]))
def test_call_sequence_binds(script):
    checked = []
    for call in Z["scripts"][script]["main_calls"] + Z["scripts"][script]["entry_calls"]:
        if call["callee"] in NOT_ON_PATH:
            continue
        obj = _resolve(script, call["callee"])
        if obj is None:
            continue
        assert obj != "MISSING", f"{script}:{call['line']}: {call['callee']} has no counterpart"
        fn = obj.__init__ if inspect.isclass(obj) else obj
        sig = inspect.signature(fn)
        args = [None] * call["n_pos"]
        if inspect.isclass(obj) or (inspect.isfunction(fn) and list(sig.parameters)[:1] == ["self"]):
            args = [None] + args  # self
        try:
            sig.bind(*args, **{k: None for k in call["kw"]})
        except TypeError as e:
            raise AssertionError(f"{script}:{call['line']}: {call['callee']}({call['n_pos']} positional, {call['kw']}) does not bind: {e}")
        checked.append(call["callee"])
    want = {"main_pretrain_stage1.py": {"build_model", "build_loader", "model.prepare_for_training", "build_optimizer", "deepspeed.initialize",
                                        "EpochBasedTrainer", "auto_resume_helper", "trainer.train", "model.custom_save_checkpoint",
                                        "deepspeed_init_distributed", "setup_logger"},
            "cli_qa.py": {"build_model", "build_vlp_transform", "model.to", "default_conversation.copy", "model.custom_load_state_dict",
                          "vision_processor", "conv.append_message", "conv.get_prompt", "tokenizer_image_token", "KeywordsStoppingCriteria",
                          "model.generate"},
            "main_cls.py": {"build_model", "build_zero_shot_loader", "default_conversation.copy", "conv.append_message", "conv.get_prompt",
                            "tokenizer_image_token", "model.generate", "model.custom_load_state_dict", "init_distributed", "setup_logger"},
            "main_vqa.py": {"build_model", "RSVQAHR", "RSVQALR", "DataCollatorForVQASupervisedDataset", "model.generate", "is_main_process", "init_distributed"},
            "main_vg.py": {"build_model", "VGEvalDataset", "DataCollatorForVGSupervisedDataset", "model.generate", "is_main_process", "init_distributed"},
            "main_bench_gen.py": {"build_model", "default_conversation.copy", "dummy_conv.append_message", "dummy_conv.get_prompt", "tokenizer_image_token",
                                  "vis_transform", "model.generate", "init_distributed"}}
    for name in want.get(script, ()):
        assert name in checked, f"{script}: {name} was not exercised (fixture or resolver changed?)"


def test_reference_parameter_names_are_accepted():
    from lhrs_bot_amd.conversation import Conversation
    from lhrs_bot_amd.trainer import ConfigArgumentParser, Trainer
    from lhrs_bot_amd.unibind import UniBind
    local = {"Trainer": Trainer, "UniBind": UniBind, "Conversation": Conversation, "ConfigArgumentParser": ConfigArgumentParser}
    for qual, ref in Z["signatures"].items():
        head = qual.split(".")[0]
        if head in local:
            obj = getattr(local[head], qual.split(".", 1)[1])
        else:
            mod, _, name = qual.rpartition(".")
            obj = getattr(importlib.import_module(mod), name)
        fn = obj.__init__ if inspect.isclass(obj) else obj
        sig = inspect.signature(fn)
        names = set(sig.parameters)
        var_kw = any(p.kind is p.VAR_KEYWORD for p in sig.parameters.values())
        for prm in ref["params"] + ref["kwonly"]:
            assert prm in names or var_kw, f"{qual}: reference parameter {prm!r} is not accepted"
        ours_required = [n for n, p in sig.parameters.items() if n != "self" and p.default is p.empty and p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
        assert set(ours_required) <= set(ref["required"]) | {"config"} or qual.endswith("Trainer"), \
            f"{qual}: requires {ours_required}, the reference only {ref['required']}"


def test_engine_object_has_the_deepspeed_surface_the_trainer_and_hooks_use():
    """trainer.py:297,319 / deepspeed_hook.py:6-18: engine(batch), backward, step, save_checkpoint(dir, tag, client_state), load_checkpoint,
    optimizer.param_groups / _global_grad_norm."""
    from lhrs_bot_amd.engine import LHRSEngine, _Optimizer
    for m in ("__call__", "backward", "step", "save_checkpoint", "load_checkpoint", "train"):
        assert callable(getattr(LHRSEngine, m))
    inspect.signature(LHRSEngine.save_checkpoint).bind(None, "dir", "tag", client_state={})
    inspect.signature(LHRSEngine.load_checkpoint).bind(None, "path")
    o = _Optimizer(1e-3, 0.0, 10, 2)
    assert {"lr", "weight_decay"} <= set(o.param_groups[0]) and hasattr(o, "_global_grad_norm")


def test_initialize_rejects_what_it_cannot_honour():
    from lhrs.CustomTrainer import initialize
    with pytest.raises(ValueError, match="fp16"):
        initialize(config={"fp16": {"enabled": True}}, model=object(), optimizer=None)
    with pytest.raises(ValueError, match="no optimizer"):
        initialize(config={"bf16": {"enabled": True}}, model=object(), optimizer=None)
    with pytest.raises(TypeError, match="OptimizerSpec"):
        import torch
        initialize(config={}, model=object(), optimizer=torch.optim.SGD([torch.zeros(1, requires_grad=True)], lr=0.1))
