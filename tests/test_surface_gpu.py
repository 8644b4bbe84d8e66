"""The reference's stage-1 call sequence on a real (synthetic-content) corpus directory, through the `lhrs.*` names only:
build_model -> build_loader (CaptionDatasetVQA over <root>/RSICD_Image + RSICD.json, DataLoader workers decode, the device resizes) ->
prepare_for_training -> build_optimizer -> initialize -> EpochBasedTrainer -> train -> auto_resume_helper -> resume -> FINAL.pt."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dataset_cases as DC  # noqa: E402


def _config(tmp_path, **kw):
    from lhrs.CustomTrainer.utils import ConfigDict
    c = ConfigDict(dict(stage=1, batch_size=2, workers=1, data_path=str(tmp_path / "corpus"), prompt_template="plain", output=str(tmp_path / "out"),
                        accelerator="gpu", enable_amp=True, wandb=False, gpus=0, local_rank=0, rank=0, world_size=1, is_distribute=False,
                        inf_sampler=False, optimizer="adanp", lr=2e-4, wd=0.0, max_grad_norm=0.3, epochs=2, llama_layers=1, seed=322,
                        bf16=True, fp16=False, accumulation_steps=1, tune_rgb_bk=False, tune_rgb_pooler=True, tune_im_start=False,
                        lora=dict(enable=False), schedule=dict(name="cosine", min_lr=0.0, warmup_epochs=2, warmup_method="linear", warmup_factor=0.1),
                        rgb_vision=dict(arch="vit_large", vit_name="openai/clip-vit-large-patch14"), text=dict(path="/nonexistent/Llama-2-7b-chat-hf"),
                        transform=dict(input_size=[224, 224]), log_period=1))
    c.update(kw)
    return c


@pytest.mark.timeout(1200)
def test_stage1_driver_on_a_corpus_directory_with_resume(tmp_path):
    import main_pretrain_stage1 as drv
    from lhrs.CustomTrainer.utils import auto_resume_helper
    DC.build_case(str(tmp_path / "corpus"), "rsicd")
    os.makedirs(tmp_path / "out" / "checkpoints", exist_ok=True)
    cfg = _config(tmp_path)
    t = drv.main(cfg)
    eng = t.model
    assert eng.module.base_weights == {"rgb": "random", "text": "random"}            # announced fallback, not silent
    assert len(t.data_loader) == 3 and t.max_iters == 6 and eng.global_steps == 6     # 6 pictures / batch 2, drop_last, 2 epochs
    assert len(t.history) == 6 and all(torch.isfinite(torch.tensor(h["loss"])) for h in t.history)
    assert t.history[0]["lr"] < t.history[1]["lr"]                                    # linear warm-up of the cosine hook
    assert os.path.exists(tmp_path / "out" / "checkpoints" / "FINAL.pt")
    # a mid-run checkpoint -> auto_resume_helper finds it -> a fresh run resumes at that iteration
    t.save_checkpoint("iter_5")
    found = auto_resume_helper(str(tmp_path / "out"))
    assert found is not None and found.endswith("iter_5.pth")
    cfg2 = _config(tmp_path, auto_resume=True, epochs=3)
    t2 = drv.main(cfg2)
    assert cfg2.resume_path == found and t2.model.global_steps == 9 and len(t2.history) == 3


@pytest.mark.timeout(900)
def test_uint8_pictures_of_different_sizes_equal_the_float_path(tmp_path):
    """batch["rgb"] as the loader delivers it (a list of uint8 HWC pictures) gives the same loss as the float tensor the reference's
    CLIPImageProcessor would have produced (the device transform is bit-exact: tests/test_image_gpu.py)."""
    from lhrs.Dataset.build_transform import build_vlp_transform
    from lhrs.models import build_model
    cfg = _config(tmp_path)
    model = build_model(cfg, activate_modal=("rgb", "text"))
    model.prepare_for_training()
    g = torch.Generator().manual_seed(0)
    pics = [torch.randint(0, 256, (h, w, 3), generator=g, dtype=torch.uint8) for h, w in ((240, 320), (300, 224))]
    ids = torch.randint(3, 32000, (2, 10), generator=g)
    ids[:, 0], ids[:, 1] = 1, -200
    labels = ids.clone()
    labels[:, :2] = -100
    proc = build_vlp_transform(cfg, is_train=False)
    px = proc(pics, return_tensors="pt").pixel_values
    assert tuple(px.shape) == (2, 3, 224, 224) and px.dtype == torch.float32
    a = model(dict(rgb=pics, input_ids=ids, labels=labels, attention_mask=ids.ne(0)))["total_loss"].item()
    b = model(dict(rgb=px, input_ids=ids, labels=labels, attention_mask=ids.ne(0)))["total_loss"].item()
    assert a == b


@pytest.mark.timeout(1200)
def test_stage1_driver_on_rs5m_tar_shards_has_an_epoch_length(tmp_path):
    """Stage 1 on an RS5M path (`"RS5M" in config.data_path` -> the tar-shard pipeline, build_loader.py:110-160) through the trainer: the loader
    has a length (`Trainer.epoch_len` = len(data_loader): the reference's `with_epoch(num_worker_batches)`), every worker yields exactly its share
    of batches - walking its shards again when they run dry - and the epoch ends after exactly that many optimizer steps."""
    import main_pretrain_stage1 as drv
    from test_datasets_cpu import _make_rs5m_shards
    root = str(tmp_path / "RS5M")
    _make_rs5m_shards(root)                                           # 4 shards x 5 samples
    os.makedirs(tmp_path / "out" / "checkpoints", exist_ok=True)
    cfg = _config(tmp_path, data_path=root, workers=2, epochs=1, rs5m_num_samples=28)     # ceil(28 / 2) = 14 batches = 2 workers x 7: more than one pass over a worker's 10 samples
    t = drv.main(cfg)
    assert len(t.data_loader) == 14 and t.max_iters == 14 and t.model.global_steps == 14
    assert len(t.history) == 14 and all(torch.isfinite(torch.tensor(h["loss"])) for h in t.history)
