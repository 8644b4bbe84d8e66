"""`lhrs` - the reference's import surface on the gfx950 engine.

The reference's entry scripts (main_pretrain_stage{1,2,3}.py:13-23, cli_qa.py:10-22) import `lhrs.models`, `lhrs.CustomTrainer`,
`lhrs.CustomTrainer.utils`, `lhrs.Dataset.{build_loader,build_transform,conversation}`, `lhrs.optimizer`, `lhrs.utils`.  This package
carries exactly those module paths and names; every object is implemented in `lhrs_bot_amd/` (HIP kernels behind the C ABI of
include/lhrs_hip.h) and only re-exported here.  Unlike the reference's `lhrs/__init__.py` (which eagerly imports deepspeed, wandb,
webdataset, timm, torchvision through its sub-packages) importing `lhrs` pulls nothing but torch.
"""
