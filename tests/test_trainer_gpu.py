"""The reference's training loop surface on the HIP engine: hooks, LR schedule, checkpoint round trip (tiny model)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from lhrs_bot_amd.engine import LHRSEngine, cosine_warmup_lr  # noqa: E402
from lhrs_bot_amd.trainer import EpochBasedTrainer, IterBasedTrainer, SyntheticStage1Loader  # noqa: E402
from lhrs_bot_amd.unibind import build_model  # noqa: E402


def make_engine(lr=2e-3):
    model = build_model(None, device="cuda", llama_layers=1, vit_layers=24).init_random(seed=0)
    model.prepare_for_training()
    return LHRSEngine(model, optimizer="adanp", lr=lr, max_grad_norm=0.3)


def test_epoch_trainer_trains_and_resumes(tmp_path):
    sched = dict(name="cosine", min_lr=2e-5, warmup_epochs=4, warmup_method="linear", warmup_factor=0.1)
    loader = SyntheticStage1Loader(batch_size=3, epoch_len=6, caption_tokens=(5, 20), seed=1)
    eng = make_engine()
    tr = EpochBasedTrainer(model=eng, optimizer=eng.optimizer, lr_scheduler=sched, data_loader=loader, max_epochs=2,
                           work_dir=str(tmp_path), log_period=1, ckpt_period=6, max_num_checkpoints=1, deepspeed=True)
    tr.train()
    assert len(tr.history) == 12 and all(torch.isfinite(torch.tensor(h["loss"])) for h in tr.history)
    # lr follows the reference schedule; the last logged lr belongs to iteration 11
    assert tr.history[-1]["lr"] == pytest.approx(cosine_warmup_lr(11, 2e-3, 12, 2e-5, 4, 0.1, "linear"))
    assert tr.history[0]["lr"] == pytest.approx(cosine_warmup_lr(0, 2e-3, 12, 2e-5, 4, 0.1, "linear"))
    # same data every epoch (fixed seed): the projector must have learned something
    assert tr.history[-1]["loss"] < tr.history[0]["loss"]
    ckpts = os.listdir(os.path.join(str(tmp_path), "checkpoints"))
    assert ckpts == ["iter_11.pth"]  # rotation kept one
    # resume.  As in the reference, the checkpoint hook runs BEFORE the engine step of its iteration (hook order
    # [ckpt, step, lr, dist, logger]): iter_11.pth holds the state after 11 steps and cur_stat = 12.
    path = os.path.join(str(tmp_path), "checkpoints", "iter_11.pth")
    saved = torch.load(path, map_location="cpu")
    assert saved["cur_stat"] == 12 and saved["engine"]["global_steps"] == 11
    eng2 = make_engine()
    tr2 = EpochBasedTrainer(model=eng2, optimizer=eng2.optimizer, lr_scheduler=sched, data_loader=loader, max_epochs=2,
                            work_dir=str(tmp_path / "r"), log_period=1, ckpt_period=0, deepspeed=True)
    tr2.train(load_checkpoint=path)  # nothing left to do: restored state must be exactly the saved one
    assert eng2.global_steps == 11 and len(tr2.history) == 0
    assert torch.equal(eng2.pool.master.cpu(), saved["engine"]["master"])
    assert torch.equal(eng2.pool.shadow, eng2.pool.master.to(torch.bfloat16))
    assert eng.global_steps == 12


def test_iter_trainer_and_final_checkpoint(tmp_path):
    eng = make_engine()
    loader = SyntheticStage1Loader(batch_size=2, epoch_len=2, caption_tokens=(4, 9), seed=2)
    tr = IterBasedTrainer(model=eng, optimizer=eng.optimizer, lr_scheduler={"name": "const"}, data_loader=loader, max_iters=5,
                          work_dir=str(tmp_path), log_period=2, ckpt_period=0, deepspeed=True)
    tr.train()
    assert eng.global_steps == 5 and [h["iter"] for h in tr.history] == [2, 4, 5]
    ck = eng.model.custom_save_checkpoint(str(tmp_path / "final"))
    assert set(ck) == {"rgb_ckpt", "other_ckpt"} and ck["other_ckpt"]["rgb_pooler"]["query"].shape == (1, 144, 1024)
    fresh = build_model(None, device="cuda", llama_layers=1).init_random(seed=5)
    fresh.custom_load_state_dict(str(tmp_path / "final" / "FINAL.pt"))
    assert torch.equal(fresh.rgb_pooler.master, eng.pool.master)


def test_gradient_accumulation_equals_one_step_on_the_mean_gradient():
    """DeepSpeed gradient_accumulation_steps = 2 (main_pretrain_stage1.py:61,115): two micro-batches, loss scaled by 1/2, ONE
    optimizer step at the boundary == a plain step on the mean of the two micro-batch gradients; step() is a no-op inside the window."""
    from bench import make_batch
    from lhrs_bot_amd.engine import LHRSEngine
    from lhrs_bot_amd.unibind import UniBind

    def build(gas):
        m = UniBind(("rgb", "text"), None, device="cuda", llama_layers=1).init_random(seed=3)
        m.prepare_for_training()
        return m, LHRSEngine(m, optimizer="adanp", lr=1e-3, weight_decay=0.0, max_grad_norm=0.3, gradient_accumulation_steps=gas)

    bA, bB = make_batch(2, 20, torch.device("cuda"), seed=1), make_batch(2, 20, torch.device("cuda"), seed=2)
    m2, e2 = build(2)
    before = m2.rgb_pooler.master.clone()
    e2(bA); e2.backward(); assert not e2.is_gradient_accumulation_boundary() or True
    e2.step()
    assert torch.equal(m2.rgb_pooler.master, before) and e2.global_steps == 0      # inside the window: nothing moves
    e2(bB); e2.backward(); e2.step()
    assert e2.global_steps == 1
    # reference: gradients of the two micro-batches computed separately, averaged, one step
    m1, e1 = build(1)
    e1(bA); e1.backward(); gA = m1.rgb_pooler.grad.clone()
    e1(bB); e1.backward(); gB = m1.rgb_pooler.grad.clone()
    m1.rgb_pooler.grad.copy_((gA + gB) * 0.5)
    e1.step()
    a, b = m2.rgb_pooler.master, m1.rgb_pooler.master
    assert ((a - b).norm() / (b - before).norm()).item() < 2e-3          # bf16 rounding of the 1/2-scaled backward vs scaling afterwards
    assert (b - before).norm().item() > 0


def test_stage2_and_stage3_drivers_from_yaml_keys(tmp_path):
    """The shared driver with the stage-2 / stage-3 YAML keys (`stage`, `lora`, `bits: 8`, `optimizer: adamw`, `betas`, `epochs`): LoRA comes
    from the config (text_modal.py:133-151), the base goes to 8 bit, stage 3 runs IterBasedTrainer(max_iters=epochs) on adapters
    loaded from the stage-2 checkpoint directory (UniBind.custom_load_state_dict -> TextLoRA/)."""
    import main_pretrain_stage1 as drv
    from lhrs_bot_amd.trainer import ConfigDict

    def cfg(stage, out, **kw):
        c = ConfigDict(dict(stage=stage, batch_size=2, epoch_len=3, seed=1, llama_layers=1, log_period=1, output=str(out), accelerator="gpu",
                            enable_amp=True, wandb=False, gpus=0, local_rank=0, rank=0, world_size=1, is_distribute=False,
                            optimizer="adamw", lr=1e-4, wd=0.0, max_grad_norm=1.0, betas=[0.9, 0.95], epochs=1, bits=8,
                            tune_rgb_pooler=stage == 2, lora=dict(enable=stage == 2, lora_r=8, lora_alpha=16, lora_dropout=0.05)))
        c.update(kw)
        return c

    t2 = drv.main(cfg(2, tmp_path / "s2"))
    m2 = t2.model.module
    assert m2.text.lora is not None and m2.text.lora.r == 8 and m2.text.lora.dropout == 0.05 and m2.text.base_int8
    assert {s.name for s in t2.model.stores} == {"rgb_pooler", "lora"} and t2.model.global_steps == 3
    assert os.path.isdir(tmp_path / "s2" / "checkpoints" / "TextLoRA")
    t3 = drv.main(cfg(3, tmp_path / "s3", epochs=4, model_path=str(tmp_path / "s2" / "checkpoints" / "FINAL.pt")))
    m3 = t3.model.module
    assert m3.text.lora is not None and m3.text.lora.r == 8 and m3.text.base_int8    # adapters came from TextLoRA/, base re-quantised (LLM.int8: the YAML default)
    assert {s.name for s in t3.model.stores} == {"lora"} and t3.model.global_steps == 4
    assert len(t3.history) == 4 and all(torch.isfinite(torch.tensor(h["loss"])) for h in t3.history)


def test_stage2_and_stage3_drivers_from_the_shipped_yaml_trees(tmp_path):
    """Boundary (b), config surface: Config/multi_modal_stage{2,3}.yaml AS SHIPPED (`fp16: True, bf16: False, optimizer: adamw, bits: 8,
    dtype: float16, double_quant, quant_type, lora r=128`; trees from tests/golden/yaml_surface.json) through the reference's call sequence
    parse_option -> build_model -> prepare_for_training -> build_ds_config -> initialize -> trainer (main_pretrain_stage2.py:28-85,178-258).
    Only launcher flags are added (1 decoder layer, 3 / 4 iterations, synthetic batches)."""
    import json
    import yaml
    import main_pretrain_stage1 as drv
    z = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "yaml_surface.json")))

    def run(name, stage, out, *extra):
        path = tmp_path / f"multi_modal_{name}.yaml"
        path.write_text(yaml.safe_dump(z["yaml"][name]))
        config = drv.parse_option(["-c", str(path), "--batch-size", "2", "--output", str(out), "--accelerator", "gpu", "--enable-amp", "True",
                                   "--use-checkpoint", "--llama-layers", "1", "--epoch-len", "3", "--data-path", "synthetic", *extra])
        assert config.fp16 is True and config.bf16 is False and config.optimizer == "adamw" and config.bits == 8 and config.stage == stage
        config.rank, config.local_rank, config.world_size, config.is_distribute = 0, 0, 1, False
        os.makedirs(os.path.join(config.output, "checkpoints"), exist_ok=True)
        return drv.main(config)

    t2 = run("stage2", 2, tmp_path / "s2")
    m2 = t2.model.module
    assert m2.text.lora is not None and m2.text.lora.r == 128 and m2.text.lora.dropout == 0.05 and m2.text.base_int8
    assert t2.model.opt_name == "adamw" and t2.model.precision_request == "fp16" and t2.model.max_grad_norm == 1.0
    assert {s.name for s in t2.model.stores} == {"rgb_pooler", "lora"} and t2.model.global_steps == 3
    assert all(torch.isfinite(torch.tensor(h["loss"])) for h in t2.history)
    t3 = run("stage3", 3, tmp_path / "s3", "--model-path", str(tmp_path / "s2" / "checkpoints" / "FINAL.pt"), "--opts", "epochs", "4")
    m3 = t3.model.module
    assert m3.text.lora is not None and m3.text.lora.r == 128 and m3.text.base_int8     # adapters from TextLoRA/, `lora.enable: False` in the YAML
    assert {s.name for s in t3.model.stores} == {"lora"} and t3.model.global_steps == 4 and len(t3.history) == 4


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("opt,wd", [("adanp", 0.02), ("adamw", 0.05)])
def test_three_optimizer_steps_track_the_oracle_training_loop(opt, wd):
    """The WHOLE stage-1 step, three times over: forward, backward, global-norm clipping, the optimizer rule with the reference's decay groups
    (build_optimizer.py:41-73: 1-D tensors and biases do not decay), bf16 refresh - against the oracle's autograd + restated Adan / AdamW in fp64 on the
    same three batches.  Losses agree step by step (1e-3); the accumulated parameter change of the projector points the same way (cosine) and has
    the same size; the bf16 gradients' sign noise on near-zero entries is what keeps an Adam-type update from agreeing element by element."""
    import math
    from lhrs_bot_amd.engine import LHRSEngine
    from lhrs_bot_amd.unibind import UniBind
    from oracle import lhrs_oracle as O
    from oracle import params as OP
    from oracle.optim_oracle import adamw_step_ref, adan_step_ref, clip_coef

    nl, lr, max_norm = 2, 2e-4, 0.3          # the stage-1 YAML's lr and the DeepSpeed gradient_clipping of build_ds_config
    P = {"vit": OP.make_vit_params(seed=2), "pooler": OP.make_pooler_params(seed=1), "llama": OP.make_llama_params(seed=3, layers=nl)}
    model = UniBind(("rgb", "text"), None, device="cuda", llama_layers=nl).load_params(P)
    model.prepare_for_training()
    eng = LHRSEngine(model, optimizer=opt, lr=lr, weight_decay=wd, max_grad_norm=max_norm)
    names = [n for n, _ in model.rgb_pooler.named_parameters()]
    start = {n: v.detach().cpu().double().clone() for n, v in model.rgb_pooler.named_parameters()}

    def batch(seed):
        g = torch.Generator().manual_seed(seed)
        B, T = 2, 24
        ids = torch.randint(3, 32000, (B, T), generator=g)
        ids[:, 0], ids[:, 1] = 1, -200
        ids[1, 20:] = 0
        labels = ids.clone()
        labels[:, :2] = -100
        labels[ids == 0] = -100
        return dict(rgb=torch.randn(B, 3, 224, 224, generator=g), input_ids=ids, labels=labels, attention_mask=ids.ne(0))

    batches = [batch(40 + i) for i in range(3)]
    got_losses = []
    for b in batches:
        out = eng(b)
        eng.backward()
        eng.step()
        got_losses.append(out["total_loss"].item())
    torch.cuda.synchronize()

    ref = OP.pooler_to_ref(P["pooler"])                      # reference state-dict names -> the oracle's leaf tensors
    leaves = [P["pooler"]["query"], P["pooler"]["out_proj_w"], P["pooler"]["out_proj_b"]] + [v for L in P["pooler"]["layers"] for v in L.values() if torch.is_tensor(v)]
    for v in leaves:
        v.requires_grad_(True)
    decays = {n: not (ref[n].squeeze(0).dim() == 1 or n.endswith(".bias")) for n in names}
    state = {n: dict(p=ref[n].detach().double().clone(), m=torch.zeros_like(ref[n], dtype=torch.float64), v=torch.zeros_like(ref[n], dtype=torch.float64),
                     n=torch.zeros_like(ref[n], dtype=torch.float64), pre=None) for n in names}
    want_losses = []
    for step, b in enumerate(batches, 1):
        for v in leaves:
            v.grad = None
        loss = O.unibind_forward(P, b)
        loss.backward()
        want_losses.append(loss.item())
        grads = {n: (P["pooler"]["query"].grad[None] if n == "query" else ref[n].grad).double() for n in names}
        coef = clip_coef(torch.cat([g.reshape(-1) for g in grads.values()]), max_norm)
        for n in names:
            rule = adan_step_ref if opt == "adanp" else adamw_step_ref
            rule(state[n], grads[n] * coef, step, lr=lr, wd=wd if decays[n] else 0.0)
            with torch.no_grad():
                (P["pooler"]["query"] if n == "query" else ref[n]).copy_(state[n]["p"].reshape(ref[n].shape if n != "query" else P["pooler"]["query"].shape).float()
                                                                         if n != "query" else state[n]["p"][0].float())
    print("losses", got_losses, want_losses)
    for a, w in zip(got_losses, want_losses):
        assert abs(a - w) < 1e-3 * w, (got_losses, want_losses)
    d_got = torch.cat([(v.detach().cpu().double() - start[n]).reshape(-1) for n, v in model.rgb_pooler.named_parameters()])
    d_want = torch.cat([(state[n]["p"] - start[n].reshape(state[n]["p"].shape)).reshape(-1) for n in names])
    cos = (d_got @ d_want / (d_got.norm() * d_want.norm())).item()
    print("cos", cos, "norm ratio", d_got.norm().item() / d_want.norm().item())
    assert cos > 0.97, cos
    assert abs(d_got.norm().item() / d_want.norm().item() - 1.0) < 3e-2
    # the 2-D weights with the largest gradients (the output projection) agree closely element by element
    i = names.index("out_proj.weight")
    a = dict(model.rgb_pooler.named_parameters())["out_proj.weight"].detach().cpu().double() - start["out_proj.weight"]
    w = state["out_proj.weight"]["p"] - start["out_proj.weight"]
    assert ((a - w).norm() / w.norm()).item() < 0.15, ((a - w).norm() / w.norm()).item()
    assert math.isfinite(sum(got_losses)) and i >= 0
