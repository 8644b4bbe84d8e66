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
    with pytest.raises(ValueError, match="no optimizer"):
        initialize(config={"bf16": {"enabled": True}}, model=object(), optimizer=None)
    with pytest.raises(TypeError, match="OptimizerSpec"):
        import torch
        initialize(config={}, model=object(), optimizer=torch.optim.SGD([torch.zeros(1, requires_grad=True)], lr=0.1))


# ------------------------------------------------------------------------------------------------ the shipped YAMLs, unchanged
def _yaml_surface():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "yaml_surface.json")))


def _tuples_to_lists(o):
    if isinstance(o, dict):
        return {k: _tuples_to_lists(v) for k, v in o.items()}
    return [_tuples_to_lists(v) for v in o] if isinstance(o, (list, tuple)) else o


@pytest.mark.parametrize("name", ["stage1", "stage2", "stage3"])
def test_shipped_yaml_through_parse_build_ds_config_initialize(name, tmp_path, monkeypatch, caplog):
    """Config/multi_modal_stage{1,2,3}.yaml as the reference ships them (tests/golden/yaml_surface.json holds their parsed trees and what the
    reference's own build_ds_config returns for them): parse_option -> build_ds_config gives the SAME dict, and `initialize` takes it -
    `fp16: True, bf16: False, optimizer: adamw` of stages 2/3 included - with a warning naming the bf16 deviation, never an error
    (main_pretrain_stage2.py:28-85, Config/multi_modal_stage2.yaml:75-89)."""
    import logging
    import yaml
    import main_pretrain_stage1 as drv
    import lhrs_bot_amd.engine as eng
    from lhrs.CustomTrainer import initialize
    from lhrs.optimizer import build_optimizer
    z = _yaml_surface()
    path = tmp_path / f"multi_modal_{name}.yaml"
    path.write_text(yaml.safe_dump(z["yaml"][name]))
    launch = z["launch"][name]
    config = drv.parse_option(["-c", str(path), "--batch-size", str(launch["batch_size"]), "--accumulation-steps", str(launch["accumulation_steps"]),
                               "--output", str(tmp_path / "out"), "--accelerator", "gpu", "--enable-amp", "True", "--use-checkpoint"])
    for k, v in z["yaml"][name].items():                      # every YAML key reaches the config, unmodified
        assert _tuples_to_lists(config[k]) == v, k
    ds = drv.build_ds_config(config)
    assert _tuples_to_lists(ds) == z["ds_config"][name]

    seen = {}

    class FakeEngine:                                          # no GPU here: record what `initialize` derives from the dict
        def __init__(self, model, **kw):
            seen.update(kw)
            self.optimizer = object()

    class FakeModel:
        def named_parameters(self):
            import torch
            return [("rgb_pooler.out_proj.weight", torch.zeros(2, 2)), ("rgb_pooler.out_proj.bias", torch.zeros(2))]

    monkeypatch.setattr(eng, "LHRSEngine", FakeEngine)
    model = FakeModel()
    opt = None if str(config.optimizer).lower() == "adamw" else build_optimizer(model, config, is_pretrain=True)
    logging.getLogger("train").propagate = True
    with caplog.at_level(logging.INFO, logger="train"):
        engine, optimizer, _, _ = initialize(config=ds, model=model, optimizer=opt, model_parameters=None)
    y = z["yaml"][name]
    assert seen["optimizer"] == {"adanp": "adanp", "adamw": "adamw"}[y["optimizer"]]
    assert seen["lr"] == y["lr"] and seen["weight_decay"] == y["wd"] and seen["max_grad_norm"] == y["max_grad_norm"]
    assert seen["gradient_accumulation_steps"] == 1 and seen["comm_dtype"] is __import__("torch").float32
    if y["optimizer"] == "adamw":
        assert tuple(seen["betas"]) == (0.9, 0.95)
        assert engine.precision_request == "fp16" and any("fp16.enabled=True is run as bf16" in r.message for r in caplog.records)
    else:
        assert engine.precision_request == "bf16"
    assert any("ignoring zero_optimization.stage" in r.getMessage() for r in caplog.records)


def test_eval_yaml_keys_reach_the_model_config():
    """Config/multi_modal_eval.yaml: the keys cli_qa.py / the evaluation scripts hand to build_model are the ones UniBind reads."""
    from lhrs_bot_amd.unibind import _get
    y = _yaml_surface()["yaml"]["eval"]
    assert _get(y, "rgb_vision.attn_pooler.num_query", None) == 144 and _get(y, "rgb_vision.attn_pooler.num_layers", None) == 6
    assert float(_get(y, "text.rms_norm_eps", 0)) == 1e-5 and _get(y, "text.hidden_size", None) == 4096
    assert {"dtype", "bits", "double_quant", "quant_type", "fp16", "bf16", "lora", "stage"} <= set(y)
