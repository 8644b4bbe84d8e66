"""tests/golden/splice_multi.npz: the multi-image and `tune_im_start` branches of the reference's splice, straight through the reference method.

`TextModal.prepare_inputs_for_multimodal` (lhrs/models/text_modal.py:296-526) walks every `<image>` placeholder (-200) of a sample and takes
`image_embedding[cur_image_idx]` for each - a running counter over the BATCH that a sample WITHOUT a placeholder also advances (:339) - so a
batch whose samples hold 0, 1 or several placeholders needs sum(max(1, k_b)) image slots.  With `tune_pooler and tune_im_start` (:353-387, off in
every shipped YAML) the same walk keeps `<im_start>` / `<im_end>` (the tokens either side of the placeholder) out of the detached text: the
INTEGER outputs are those of the plain walk with the placeholder's right neighbour moved before the label cut.

The method only touches `self.get_text_encoder().model.embed_tokens / .config.hidden_size`, `self.tune_pooler`, `self.tune_im_start`, so it is
called UNBOUND on a stand-in object holding a small embedding table (no 7B model needed).  Every output row is identified by what it copies
(token index, image slot row, or zero padding).  Build container only; the fixture is data."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_golden as MG  # noqa: E402

NI, DIM, VOCAB = 4, 16, 32000
PAD = -10 ** 9
AMB = -2 * 10 ** 9      # a token id that occurs more than once in its row (padding zeros): the row copies that id, whichever position


def main():
    MG.import_reference_models()
    import lhrs.models.text_modal as tm

    g = torch.Generator().manual_seed(23)
    emb = torch.nn.Embedding(VOCAB, DIM)
    with torch.no_grad():
        emb.weight.copy_(torch.randn(VOCAB, DIM, generator=g))
    enc = types.SimpleNamespace(model=types.SimpleNamespace(embed_tokens=emb), config=types.SimpleNamespace(hidden_size=DIM))

    def stand_in(tune_im_start):
        return types.SimpleNamespace(get_text_encoder=lambda: enc, tune_pooler=tune_im_start, tune_im_start=tune_im_start)

    cases = {}

    def run(name, ids, labels, mask, n_slots, tune_im_start=False):
        img = (torch.arange(n_slots * NI * DIM, dtype=torch.float32).reshape(n_slots, NI, DIM) + 1.0) * 1e-3 + 100.0   # no row equals an embedding row
        _, new_mask, _, embeds, new_labels = tm.TextModal.prepare_inputs_for_multimodal(
            stand_in(tune_im_start), input_ids=ids, attention_mask=mask, labels=labels, past_key_values=None, image_embedding=img)
        B, S = embeds.shape[:2]
        flat = img.reshape(-1, DIM)
        kind = torch.full((B, S), PAD, dtype=torch.int64)
        for b in range(B):
            for j in range(S):
                row = embeds[b, j].detach()
                hit = (flat == row).all(-1).nonzero()
                if hit.numel():
                    kind[b, j] = -(1 + int(hit[0]))                     # -(1 + slot * NI + k): row k of image slot `slot`
                elif row.abs().sum() == 0:
                    kind[b, j] = PAD
                else:
                    cand = [t for t in range(ids.shape[1]) if ids[b, t] >= 0 and torch.equal(emb.weight[ids[b, t]].detach(), row)]
                    kind[b, j] = cand[0] if len(cand) == 1 else AMB
        cases[name + "_ids"] = ids.numpy(); cases[name + "_labels"] = labels.numpy(); cases[name + "_mask"] = mask.numpy()
        cases[name + "_src"] = kind.numpy(); cases[name + "_new_labels"] = new_labels.numpy(); cases[name + "_new_mask"] = new_mask.numpy()
        cases[name + "_slots"] = np.array(n_slots); cases[name + "_tune_im_start"] = np.array(int(tune_im_start))
        print(name, "S =", S, "slots =", n_slots)

    def mk(B, T, img_pos, pad_from=None):
        ids = torch.stack([torch.randperm(30000, generator=g)[:T] + 3 for _ in range(B)])  # unique ids per row
        ids[:, 0] = 1
        for b, ps in enumerate(img_pos):
            for p in ps:
                ids[b, p] = -200
        if pad_from is not None:
            for b, pf_ in enumerate(pad_from):
                if pf_ is not None:
                    ids[b, pf_:] = 0
        labels = ids.clone()
        labels[:, :2] = -100
        labels[ids == 0] = -100
        labels[ids == -200] = -100
        return ids, labels, ids.ne(0)

    def slots(img_pos):
        return sum(max(1, len(ps)) for ps in img_pos)

    for name, B, T, pos, pad in [
        ("two_each", 2, 8, [[1, 4], [2, 6]], None),                       # same spliced length: the no-padding branch (:496-524)
        ("two_and_one", 2, 9, [[1, 5], [1]], [None, 6]),                  # ragged: right padding, mask = left ones + mask + right zeros
        ("three_none_one", 3, 10, [[1, 3, 7], [], [2]], [None, 7, None]),  # a sample without a placeholder still consumes a slot
        ("adjacent_and_last", 2, 7, [[2, 3], [1, 6]], None),              # neighbouring placeholders; a placeholder as the last token
        ("four_in_one", 1, 9, [[1, 2, 5, 8]], None),
    ]:
        run(name, *mk(B, T, pos, pad), n_slots=slots(pos))
    # tune_im_start: <im_start> <image> <im_end> - the placeholder always has both neighbours in the reference's prompts (cap_dataset.py:875-876)
    for name, B, T, pos, pad in [
        ("ims_one_each", 2, 8, [[2], [3]], None),
        ("ims_ragged", 3, 10, [[2], [], [2, 6]], [7, None, None]),
    ]:
        run(name, *mk(B, T, pos, pad), n_slots=slots(pos), tune_im_start=True)
    np.savez_compressed(os.path.join(HERE, "splice_multi.npz"), n_img_tokens=np.array(NI), **cases)
    print("splice_multi golden:", [k[:-4] for k in cases if k.endswith("_src")])


if __name__ == "__main__":
    main()
