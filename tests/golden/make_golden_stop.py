"""tests/golden/stopping_criteria.json: the reference's KeywordsStoppingCriteria (lhrs/utils/eval_utils.py:24-56) driven with a toy
tokenizer over hand-made id sequences.  Build container only."""
import importlib.util
import json
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_eval_utils", "/root/reference/lhrs/utils/eval_utils.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

VOCAB = {1: "<s>", 5: "###", 6: "stop", 7: "hello", 8: "world", 9: "</s>"}


class Tok:
    bos_token_id = 1

    def __call__(self, text):
        inv = {v: k for k, v in VOCAB.items()}
        return type("E", (), {"input_ids": [1] + [inv[w] for w in text.split(" ") if w]})()

    def batch_decode(self, ids, skip_special_tokens=True):
        return [" ".join(VOCAB[int(t)] for t in row if not (skip_special_tokens and int(t) in (1, 9))) for row in ids]


cases = []
for prompt_len in (2, 6):
    crit = ref.KeywordsStoppingCriteria(["###", "stop"], Tok(), torch.zeros((1, prompt_len), dtype=torch.long))
    for out in ([7], [7, 8], [7, 5], [6, 7, 8], [7, 8, 7, 6], [7, 8, 7, 8, 7, 8, 7, 8], [5, 7, 8, 7, 8, 7, 8, 7], [7, 7, 7, 7, 7, 7, 6, 8, 8]):
        got = crit(torch.tensor([out]), None)
        cases.append({"prompt_len": prompt_len, "out": out, "stop": bool(got)})
json.dump({"keywords": ["###", "stop"], "vocab": {str(k): v for k, v in VOCAB.items()}, "cases": cases}, open(os.path.join(HERE, "stopping_criteria.json"), "w"))
print(cases)
