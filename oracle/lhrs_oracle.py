"""TEST INFRASTRUCTURE ONLY - CPU restatement (fp32/fp64 PyTorch, autograd for gradients) of the LHRS-Bot hot path.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this module; the product
path (lhrs_bot_amd/) never does and fails loudly when liblhrs_hip.so is missing.

Each function cites the reference lines it restates.  The floating-point oracle is plain torch on CPU (the
tier's rule for a floating-point path); the integer part (`splice`) is a pure index computation and must match
bit-exactly.  The restatement is PINNED: tests/golden/*.npz hold outputs of the reference's own modules
(lhrs/models/common_arch.py:AttnPooler, lhrs/models/text_modal.py:TextModal.prepare_inputs_for_multimodal,
HF CLIPVisionModel / LlamaForCausalLM driven through lhrs/models/UniBind.py) imported in the build container
by tests/golden/make_golden.py on the same seeded parameters; tests/test_oracle_cpu.py checks this file against
them.  Caveat: the container has transformers 5.x, not the pinned 4.36.1 - same arithmetic.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

IGNORE_INDEX = -100        # lhrs/models/__init__.py:1-6
IMAGE_TOKEN_INDEX = -200

STAGE_NUM = (64, 48, 32)    # lhrs/models/common_arch.py:104
SPLIT_PART = (256, 256, 256)
VIT_TAPS = (7, 15, 22)      # lhrs/models/rgb_vision_modal.py:159-164 with 24 layers


# ------------------------------------------------------------------------------------------------- ViT
def vit_forward(p: Dict, rgb: torch.Tensor, heads: int = 16, taps=None) -> torch.Tensor:
    """VisionModal.encode (lhrs/models/rgb_vision_modal.py:166-179) over HF CLIPVisionModel:
    patch conv (no bias) -> [cls | patches] + pos -> pre_layrnorm -> pre-LN encoder layers (quick_gelu MLP);
    hidden_states[i] = output after i layers (index 0 = post-pre-LN); returns cat of the taps without CLS."""
    dim = p["cls"].numel()
    n_layers = len(p["layers"])
    if taps is None:
        taps = (n_layers // 3 - 1, n_layers // 3 * 2 - 1, n_layers - 2)
    B = rgb.shape[0]
    x = F.conv2d(rgb, p["patch_w"], stride=p["patch_w"].shape[-1]).flatten(2).transpose(1, 2)
    x = torch.cat([p["cls"].expand(B, 1, dim), x], 1) + p["pos"]
    x = F.layer_norm(x, (dim,), p["pre_ln_w"], p["pre_ln_b"], 1e-5)
    hs = [x]
    hd = dim // heads
    for L in p["layers"][: max(taps)]:
        h = F.layer_norm(x, (dim,), L["ln1_w"], L["ln1_b"], 1e-5)
        qkv = F.linear(h, L["qkv_w"], L["qkv_b"]).view(B, -1, 3, heads, hd)
        q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))
        a = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, -1) @ v
        x = x + F.linear(a.transpose(1, 2).reshape(B, -1, dim), L["o_w"], L["o_b"])
        h = F.layer_norm(x, (dim,), L["ln2_w"], L["ln2_b"], 1e-5)
        h = F.linear(h, L["fc1_w"], L["fc1_b"])
        h = h * torch.sigmoid(1.702 * h)
        x = x + F.linear(h, L["fc2_w"], L["fc2_b"])
        hs.append(x)
    return torch.cat([hs[t][:, 1:] for t in taps], 1)


# ------------------------------------------------------------------------------------------------- AttnPooler
def _mha(qn, kvn, L, heads):
    """nn.MultiheadAttention as used by ResidualAttentionBlock.attention (common_arch.py:302-313): packed in_proj,
    Q from rows [0:d], K from [d:2d], V from [2d:3d]; softmax(QK^T/sqrt(hd)) V; out_proj."""
    B, Lq, d = qn.shape
    hd = d // heads
    q = F.linear(qn, L["in_w"][:d], L["in_b"][:d]).view(B, Lq, heads, hd).transpose(1, 2)
    k = F.linear(kvn, L["in_w"][d:2 * d], L["in_b"][d:2 * d]).view(B, -1, heads, hd).transpose(1, 2)
    v = F.linear(kvn, L["in_w"][2 * d:], L["in_b"][2 * d:]).view(B, -1, heads, hd).transpose(1, 2)
    a = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, -1) @ v
    return F.linear(a.transpose(1, 2).reshape(B, Lq, d), L["out_w"], L["out_b"])


def pooler_forward(p: Dict, image_embs: torch.Tensor, heads: int = 16) -> torch.Tensor:
    """AttnPooler.forward (common_arch.py:134-173) + ResidualAttentionBlock.forward (:315-333).
    K/V input of EVERY layer is cat(initial queries of the group, image tokens of the group) (:160-166)."""
    B, _, d = image_embs.shape
    query = p["query"].expand(B, -1, -1)
    outs = []
    for q_g, img_g in zip(torch.split(query, STAGE_NUM, 1), torch.split(image_embs, SPLIT_PART, 1)):
        kv = torch.cat([q_g, img_g], 1)
        t = q_g
        for L in p["layers"]:
            kvn = F.layer_norm(kv, (d,), L["ln1kv_w"], L["ln1kv_b"], 1e-5)
            t = t + _mha(F.layer_norm(t, (d,), L["ln1_w"], L["ln1_b"], 1e-5), kvn, L, heads)
            h = F.gelu(F.linear(F.layer_norm(t, (d,), L["ln2_w"], L["ln2_b"], 1e-5), L["fc_w"], L["fc_b"]))
            t = t + F.linear(h, L["proj_w"], L["proj_b"])
        outs.append(t)
    return F.linear(torch.cat(outs, 1), p["out_proj_w"], p["out_proj_b"])


# ------------------------------------------------------------------------------------------------- splice (int)
def splice(ids: torch.Tensor, labels: Optional[torch.Tensor], mask: Optional[torch.Tensor], n_img_tokens: int):
    """Index plan of TextModal.prepare_inputs_for_multimodal (lhrs/models/text_modal.py:296-526), tune_im_start off.
    Returns (src, new_labels, new_mask): src[b, j] = token index into ids[b] (>= 0), -(1+k) for image row k of
    sample b, or -10**9 for right padding.  Restated literally, including the reference's mask rule
    `cat([True] * (new_len - T), mask)` (:511-524) and the cur_image_idx == batch index convention."""
    B, T = ids.shape
    PAD = -10 ** 9
    rows, lab_rows = [], []
    for b in range(B):
        cur = ids[b].tolist()
        lab = labels[b].tolist() if labels is not None else None
        src: List[int] = []
        nl: List[int] = []
        base = 0
        if IMAGE_TOKEN_INDEX not in cur:
            src = list(range(T))
            nl = list(lab) if lab is not None else []
        else:
            first = True
            while IMAGE_TOKEN_INDEX in cur:
                p = cur.index(IMAGE_TOKEN_INDEX)
                assert first, "one image per sample (reference indexes image_embedding by a running counter)"
                first = False
                src += [base + i for i in range(p)] + [-(1 + k) for k in range(n_img_tokens)]
                if lab is not None:
                    nl += lab[:p] + [IGNORE_INDEX] * n_img_tokens
                    lab = lab[p + 1:]
                cur = cur[p + 1:]
                base += p + 1
            src += [base + i for i in range(len(cur))]
            if lab is not None:
                nl += lab
        rows.append(src)
        lab_rows.append(nl)
    S = max(len(r) for r in rows)
    src_t = torch.full((B, S), PAD, dtype=torch.int64)
    new_labels = torch.full((B, S), IGNORE_INDEX, dtype=torch.int64) if labels is not None else None
    new_mask = torch.zeros((B, S), dtype=torch.bool) if mask is not None else None
    for b in range(B):
        n = len(rows[b])
        src_t[b, :n] = torch.tensor(rows[b], dtype=torch.int64)
        if labels is not None:
            new_labels[b, :n] = torch.tensor(lab_rows[b], dtype=torch.int64)
        if mask is not None:
            new_mask[b, : n - T] = True
            new_mask[b, n - T: n] = mask[b]
    return src_t, new_labels, new_mask


def splice_multi(ids: torch.Tensor, labels: Optional[torch.Tensor], mask: Optional[torch.Tensor], n_img_tokens: int, tune_im_start: bool = False):
    """The GENERAL walk of TextModal.prepare_inputs_for_multimodal (lhrs/models/text_modal.py:318-438): any number of `<image>` placeholders per
    sample, each taking `image_embedding[cur_image_idx]` with the running counter over the batch that a placeholder-free sample also advances (:339).
    Returns (src, new_labels, new_mask, n_slots); image rows are encoded GLOBALLY: src = -(1 + slot * n_img_tokens + k).
    tune_im_start (the `tune_pooler and tune_im_start` branch, :353-387): the same token map - tokens[:p-1] (detached), `<im_start>` = token p-1, image,
    `<im_end>` = token p+1, rest - but the LABEL kept for `<im_end>` is labels[p] (the placeholder's) and the walk resumes at p+2 (:382-386)."""
    B, T = ids.shape
    PAD = -10 ** 9
    rows, lab_rows = [], []
    slot = 0
    for b in range(B):
        cur = ids[b].tolist()
        lab = labels[b].tolist() if labels is not None else None
        src: List[int] = []
        nl: List[int] = []
        base = 0
        if IMAGE_TOKEN_INDEX not in cur:
            rows.append(list(range(T)))
            lab_rows.append(list(lab) if lab is not None else [])
            slot += 1
            continue
        while IMAGE_TOKEN_INDEX in cur:
            p = cur.index(IMAGE_TOKEN_INDEX)
            src += [base + i for i in range(p)] + [-(1 + slot * n_img_tokens + k) for k in range(n_img_tokens)]
            step = 1
            if tune_im_start:
                assert p + 1 < len(cur), "tune_im_start: the placeholder needs its <im_end> neighbour"
                src += [base + p + 1]
                step = 2
            if lab is not None:
                nl += lab[:p] + [IGNORE_INDEX] * n_img_tokens + (lab[p:p + 1] if tune_im_start else [])
                lab = lab[p + step:]
            cur = cur[p + step:]
            base += p + step
            slot += 1
        src += [base + i for i in range(len(cur))]
        if lab is not None:
            nl += lab
        rows.append(src)
        lab_rows.append(nl)
    S = max(len(r) for r in rows)
    src_t = torch.full((B, S), PAD, dtype=torch.int64)
    new_labels = torch.full((B, S), IGNORE_INDEX, dtype=torch.int64) if labels is not None else None
    new_mask = torch.zeros((B, S), dtype=torch.bool) if mask is not None else None
    for b in range(B):
        n = len(rows[b])
        src_t[b, :n] = torch.tensor(rows[b], dtype=torch.int64)
        if labels is not None:
            new_labels[b, :n] = torch.tensor(lab_rows[b], dtype=torch.int64)
        if mask is not None:
            new_mask[b, : n - T] = True
            new_mask[b, n - T: n] = mask[b]
    return src_t, new_labels, new_mask, slot


def splice_embeds_multi(src: torch.Tensor, ids: torch.Tensor, image: torch.Tensor, embed: torch.Tensor) -> torch.Tensor:
    """Embeddings for a `splice_multi` plan: image [n_slots, n_img_tokens, d] indexed by the global row code; differentiable in `image`."""
    B, S = src.shape
    flat = image.reshape(-1, image.shape[-1])
    tok = embed[torch.gather(ids.clamp(min=0), 1, src.clamp(min=0))].to(image.dtype)
    is_img = (src < 0) & (src > -10 ** 8)
    img_rows = flat[(-src - 1).clamp(0, flat.shape[0] - 1)]
    out = torch.where(is_img[..., None], img_rows, tok)
    return torch.where((src <= -10 ** 8)[..., None], torch.zeros_like(out), out)


def splice_embeds(src: torch.Tensor, ids: torch.Tensor, image: torch.Tensor, embed: torch.Tensor) -> torch.Tensor:
    B, S = src.shape
    out = torch.zeros(B, S, embed.shape[1], dtype=image.dtype)
    for b in range(B):
        for j in range(S):
            s = int(src[b, j])
            if s >= 0:
                out[b, j] = embed[int(ids[b, s])].to(image.dtype)
            elif s > -10 ** 8:
                out[b, j] = image[b, -s - 1]
    return out


# ------------------------------------------------------------------------------------------------- LLaMA
def rope_tables(S: int, D: int = 128, theta: float = 10000.0):
    inv = 1.0 / (theta ** (torch.arange(0, D, 2).float() / D))
    fr = torch.outer(torch.arange(S).float(), inv)
    return fr.cos(), fr.sin()  # [S, D/2]


def _rope(x, cos, sin):  # x [B,H,S,D]; rotate_half convention
    D = x.shape[-1]
    c = torch.cat([cos, cos], -1).to(x.dtype)
    s = torch.cat([sin, sin], -1).to(x.dtype)
    rot = torch.cat([-x[..., D // 2:], x[..., : D // 2]], -1)
    return x * c + rot * s


def _rms(x, w, eps):
    return w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps))


_GROUP_OF = {"q": "qkv", "k": "qkv", "v": "qkv", "o": "o", "gate": "gu", "up": "gu", "down": "down"}


def _lora(L, proj, x):
    """peft lora.Linear delta (peft 0.7.1 is absent here; pinned through the merged-weight identity against the reference LLaMA:
    tests/golden/make_golden_lora.py, test_lora_restatement_pinned_to_the_reference_llama_with_merged_weights): s * (dropout(x) A^T) B^T with s = lora_alpha / r; call site
    lhrs/models/text_modal.py:133-151.  L["lora"] = {"scale": s, proj: (A [r,in], B [out,r])}; optional L["lora"]["drop"] =
    {group: mask of x's shape holding 0 or 1/(1-p)} reproduces a given dropout draw (the engine's counter-based masks, one per fused
    group - peft itself draws one per wrapped nn.Linear)."""
    lo = L.get("lora")
    if not lo or proj not in lo:
        return 0.0
    A, B = lo[proj]
    mask = (lo.get("drop") or {}).get(_GROUP_OF[proj])
    if mask is not None:
        x = x * mask
    return lo["scale"] * F.linear(F.linear(x, A), B)


def llama_hidden(p: Dict, embeds: torch.Tensor, mask: Optional[torch.Tensor], heads: int = 32, eps: float = 1e-5,
                 collect=None) -> torch.Tensor:
    """HF LlamaModel.forward as called by TextModal.decode (lhrs/models/text_modal.py:258-294): per layer
    h += o_proj(attn(RoPE(q), RoPE(k), v)); h += down(silu(gate) * up); causal + key-padding mask; final RMSNorm."""
    B, S, d = embeds.shape
    hd = d // heads
    cos, sin = rope_tables(S, hd)
    neg = torch.finfo(embeds.dtype).min
    bias = torch.triu(torch.full((S, S), neg, dtype=embeds.dtype), 1)[None, None]
    if mask is not None:
        bias = bias.masked_fill(~mask.bool()[:, None, None, :], neg)
    x = embeds
    lin = p.get("_linear", F.linear)  # oracle/int8_oracle.py swaps in the LLM.int8 product for `bits: 8` base weights
    for L in p["layers"]:
        h = _rms(x, L["ln1_w"], eps)
        qkv = lin(h, L["qkv_w"])
        qkv = qkv + torch.cat([_lora(L, pr, h) + torch.zeros_like(qkv[..., :d]) for pr in ("q", "k", "v")], -1)
        qkv = qkv.view(B, S, 3, heads, hd)
        q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))
        q, k = _rope(q, cos, sin), _rope(k, cos, sin)
        a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd) + bias, -1) @ v
        a = a.transpose(1, 2).reshape(B, S, d)
        x = x + lin(a, L["o_w"]) + _lora(L, "o", a)
        h = _rms(x, L["ln2_w"], eps)
        ff = L["gu_w"].shape[0] // 2
        gu = lin(h, L["gu_w"])
        gu = gu + torch.cat([_lora(L, pr, h) + torch.zeros_like(gu[..., :ff]) for pr in ("gate", "up")], -1)
        act = F.silu(gu[..., :ff]) * gu[..., ff:]
        x = x + lin(act, L["down_w"]) + _lora(L, "down", act)
        if collect is not None:
            collect.append(x)
    return _rms(x, p["norm_w"], eps)


def causal_lm_loss(p: Dict, hidden: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """Shifted CE inside HF LlamaForCausalLM.forward: logits.float(), ignore_index=-100, mean over valid targets."""
    logits = F.linear(hidden, p["lm_head"]).float()
    return F.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]), labels[:, 1:].reshape(-1), ignore_index=IGNORE_INDEX)


# ------------------------------------------------------------------------------------------------- UniBind
def unibind_forward(P: Dict, batch: Dict, collect: Optional[Dict] = None) -> torch.Tensor:
    """UniBind.forward (lhrs/models/UniBind.py:178-199): rgb -> ViT taps -> AttnPooler -> splice -> LLaMA -> loss."""
    taps = vit_forward(P["vit"], batch["rgb"])
    img = pooler_forward(P["pooler"], taps)
    n_ph = (batch["input_ids"] == IMAGE_TOKEN_INDEX).sum(dim=1)
    if int(n_ph.max()) > 1 or img.shape[0] != batch["input_ids"].shape[0]:         # several placeholders in a sample: the general walk, global image slots
        src, labels, mask, n_slots = splice_multi(batch["input_ids"], batch["labels"], batch["attention_mask"], img.shape[1])
        assert n_slots <= img.shape[0], "the reference indexes image_embedding[cur_image_idx] past its end"
        embeds = splice_embeds_multi(src, batch["input_ids"], img, P["llama"]["embed"])
    else:
        src, labels, mask = splice(batch["input_ids"], batch["labels"], batch["attention_mask"], img.shape[1])
        B, S = src.shape
        emb = P["llama"]["embed"]
        tok = batch["input_ids"].clamp(min=0)
        gathered = emb[torch.gather(tok, 1, src.clamp(min=0))]                      # [B,S,d] token rows
        img_rows = img[torch.arange(B)[:, None], (-src - 1).clamp(0, img.shape[1] - 1)]
        is_img = (src < 0) & (src > -10 ** 8)
        is_pad = src <= -10 ** 8
        embeds = torch.where(is_img[..., None], img_rows, gathered)
        embeds = torch.where(is_pad[..., None], torch.zeros_like(embeds), embeds)
    hidden = llama_hidden(P["llama"], embeds, mask)
    loss = causal_lm_loss(P["llama"], hidden, labels)
    if collect is not None:
        collect.update(taps=taps, image=img, embeds=embeds, labels=labels, mask=mask, hidden=hidden)
    return loss


# ------------------------------------------------------------------------------------------------- generate (greedy)
@torch.no_grad()
def generate_logits(P: Dict, rgb: torch.Tensor, input_ids: torch.Tensor, forced_tokens: torch.Tensor,
                    attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """UniBind.generate / TextModal.generate (lhrs/models/UniBind.py:214-242, lhrs/models/text_modal.py:528-627) restated
    WITHOUT a KV cache: for step t the full sequence [spliced prompt | forced_tokens[:, :t]] is re-run and the logits of the
    last position are returned -> [B, n_new, V].  Greedy decoding = argmax of these logits (HF do_sample=False).
    attention_mask (batched evaluation, left-padded prompts: main_vqa.py:205-214) goes through the splice's mask rule and hides
    keys; positions stay arange(S) because CustomLlamaForCausalLM.prepare_inputs_for_generation (text_modal.py:36-60) passes no
    position_ids; generated positions are visible (HF generate appends ones)."""
    img = pooler_forward(P["pooler"], vit_forward(P["vit"], rgb))
    emb = P["llama"]["embed"]
    if int((input_ids == IMAGE_TOKEN_INDEX).sum(dim=1).max()) > 1 or img.shape[0] != input_ids.shape[0]:   # several placeholders in a prompt
        src, _, mask, _ = splice_multi(input_ids, None, attention_mask, img.shape[1])
        B, S0 = src.shape
        embeds = splice_embeds_multi(src, input_ids, img, emb)
    else:
        src, _, mask = splice(input_ids, None, attention_mask, img.shape[1])
        B, S0 = src.shape
        tok = emb[torch.gather(input_ids.clamp(min=0), 1, src.clamp(min=0))]
        img_rows = img[torch.arange(B)[:, None], (-src - 1).clamp(0, img.shape[1] - 1)]
        embeds = torch.where((src < 0)[..., None], img_rows, tok)
    if attention_mask is None:
        mask = None
    out = []
    for t in range(forced_tokens.shape[1] + 1):
        if t > 0:
            embeds = torch.cat([embeds, emb[forced_tokens[:, t - 1]][:, None]], 1)
            if mask is not None:
                mask = torch.cat([mask, torch.ones((B, 1), dtype=mask.dtype)], 1)
        if t == forced_tokens.shape[1]:
            break
        h = llama_hidden(P["llama"], embeds, mask)
        out.append(F.linear(h[:, -1], P["llama"]["lm_head"]).float())
    return torch.stack(out, 1)
