#!/usr/bin/env python
"""Visual grounding evaluation (Acc@0.5) by generation on the gfx950 engine - the reference's `main_vg.py` call sequence over the `lhrs.*`
surface (/root/reference main_vg.py:146-313):

    python main_vg.py -c Config/multi_modal_eval.yaml --model-path <FINAL.pt dir> --data-path <image dir> --data-target <X_RSVG_test.json> \\
        --batch-size 8 --accelerator gpu

build_model -> VGEvalDataset + DataCollatorForVGSupervisedDataset (prompts LEFT-padded) -> greedy batched `model.generate`, 100 new tokens ->
per-rank json -> merged, de-duplicated by file name -> "[x1, y1, x2, y2]" parsing and IoU > 0.5 counting (`score_grounding`).
"""
import json
import logging
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lhrs.CustomTrainer.utils.distribute import is_main_process  # noqa: E402
from lhrs.Dataset import DataCollatorForVGSupervisedDataset, VGEvalDataset  # noqa: E402
from lhrs.Dataset.build_transform import build_vlp_transform  # noqa: E402
from lhrs_bot_amd.evaluation import eval_entry, eval_model, eval_parse_option, generation_weights, save_result, score_grounding  # noqa: E402

logger = logging.getLogger("train")


def parse_option(args=None):
    return eval_parse_option(args, data_target=True)


def main(config):
    model = eval_model(config)
    tokenizer = model.text.tokenizer
    dataset = VGEvalDataset(root=config.data_path, target=config.data_target, transform=build_vlp_transform(config, is_train=False), tokenizer=tokenizer)
    logger.info(f"Data Length: {len(dataset)}")
    data_loader = torch.utils.data.DataLoader(dataset, num_workers=int(config.workers), pin_memory=True, batch_size=int(config.batch_size), shuffle=False,
                                              collate_fn=DataCollatorForVGSupervisedDataset(tokenizer))
    preds = []
    with torch.no_grad():
        for image, input_ids, targets, file_name, attention_mask in data_loader:
            output_ids = model.generate(input_ids=input_ids, images=image, num_beams=1, attention_mask=attention_mask, do_sample=False, temperature=1.0,
                                        top_p=1.0, max_new_tokens=100, weights=generation_weights(config))
            outputs = [o.strip() for o in tokenizer.batch_decode(output_ids, skip_special_tokens=True)]
            preds += [dict(pred=p, target=t, filename=n) for p, t, n in zip(outputs, targets, file_name)]
    save_result(preds, config.output, "eval_save_file", "filename")
    if not is_main_process():
        return None
    with open(os.path.join(config.output, "eval_save_file.json")) as f:
        result = score_grounding(json.load(f))
    logger.info(f"Accuracy: {result['accuracy']}")
    logger.info(f"Fail Sample: {result['fail']}")
    logger.info(f"Accuracy With Fail Sample: {result['accuracy_with_fail']}")
    return dict(result, predictions=preds)


if __name__ == "__main__":
    eval_entry(main, parse_option())
