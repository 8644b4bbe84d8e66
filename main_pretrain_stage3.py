#!/usr/bin/env python
"""Stage-3 entry point.  The reference's main_pretrain_stage3.py (/root/reference main_pretrain_stage3.py:180-245) repeats stage 1's call
sequence with a different YAML (`stage`, `lora`, `bits`, `optimizer: adamw`, `betas`), a checkpoint period of 100 and - stage 3 - `IterBasedTrainer(max_iters=config.epochs)`; the shared driver in main_pretrain_stage1.py reads all of that from the config."""
import main_pretrain_stage1 as drv

if __name__ == "__main__":
    drv.run(3)
