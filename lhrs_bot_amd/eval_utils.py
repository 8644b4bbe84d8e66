"""Host-side helpers of the generate callers: `KeywordsStoppingCriteria` of
/root/reference lhrs/utils/eval_utils.py:24-56 as cli_qa.py:165-186 passes it to `model.generate(stopping_criteria=[...])`.

`TextModal.generate` calls every criterion as `criterion(new_token_ids [B, n], logits)` after each token - like HF's loop when it
is started from `inputs_embeds`, the ids it sees are the NEW tokens only.  Restated with the reference's arithmetic, including its
`start_len` quirk: `offset = min(n_new - prompt_len, 3)` is negative until more tokens than the prompt length were generated, and a
negative offset makes the decoded window `output_ids[:, -offset:]` start at column |offset| instead of covering the last 3 tokens.
"""
from __future__ import annotations

from typing import List, Sequence

import torch


class KeywordsStoppingCriteria:
    def __init__(self, keywords: Sequence[str], tokenizer, input_ids: torch.Tensor):
        self.keywords = list(keywords)
        self.keyword_ids: List[torch.Tensor] = []
        for keyword in keywords:
            cur = tokenizer(keyword).input_ids
            if len(cur) > 1 and cur[0] == tokenizer.bos_token_id:
                cur = cur[1:]
            self.keyword_ids.append(torch.tensor(cur))
        self.tokenizer = tokenizer
        self.start_len = input_ids.shape[1]

    def __call__(self, output_ids: torch.Tensor, scores=None, **kwargs) -> bool:
        assert output_ids.shape[0] == 1, "Only support batch size 1 (yet)"
        offset = min(output_ids.shape[1] - self.start_len, 3)
        self.keyword_ids = [k.to(output_ids.device) for k in self.keyword_ids]
        for k in self.keyword_ids:
            # tensor truth value, as in the reference: a multi-token keyword raises "Boolean value of Tensor ... is ambiguous"
            if bool(output_ids[0, -k.shape[0]:] == k):
                return True
        outputs = self.tokenizer.batch_decode(output_ids[:, -offset:], skip_special_tokens=True)[0]
        return any(keyword in outputs for keyword in self.keywords)
