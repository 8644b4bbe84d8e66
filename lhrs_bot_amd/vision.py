"""VisionModal: CLIP ViT-L/14 forward on gfx950 (frozen in every shipped stage).

Mirrors /root/reference lhrs/models/rgb_vision_modal.py: `VisionModal.encode` (:166-179) returns
cat(hidden_states[7][:,1:], hidden_states[15][:,1:], hidden_states[22][:,1:]) with `extract_stage` (:159-164).
The reference computes all 24 layers + post_layernorm; layers after the last tap are dead code for this path and
are skipped here.  All arithmetic is liblhrs_hip.so; torch only owns the memory.
"""
from __future__ import annotations

import os
from typing import Dict, List

import torch

from . import kernels as hk


class VisionModal:
    EMBEDDING_DIM = {"vit_base": 768, "vit_large": 1024}  # rgb_vision_modal.py:125-128

    def __init__(self, config=None, device="cuda", layers=24, dim=1024, ff=4096, heads=16, patch=14, img=224):
        self.device = torch.device(device)
        hk.ensure_gemm_workspace(self.device)
        self.layers_n, self.dim, self.ff, self.heads, self.patch, self.img = layers, dim, ff, heads, patch, img
        self.n_patch = (img // patch) ** 2
        self.kp = (3 * patch * patch + 63) // 64 * 64  # im2col K padded to the GEMM's K % 64 rule (588 -> 640)
        self.extract_stage = [layers // 3 - 1, layers // 3 * 2 - 1, layers - 2]  # rgb_vision_modal.py:159-164
        self.p: Dict = {}
        self._src = None
        self._desc_cache: Dict[int, torch.Tensor] = {}

    # ------------------------------------------------------------------ parameters
    def load_params(self, p: Dict) -> None:
        """p: engine-layout dict of fp32 CPU tensors (oracle/params.py layout)."""
        dev, bf = self.device, torch.bfloat16
        pw = torch.zeros(self.dim, self.kp)
        pw[:, : 3 * self.patch ** 2] = p["patch_w"].reshape(self.dim, -1)
        self.p = {"patch_w": pw.to(dev, bf), "cls": p["cls"].to(dev, bf), "pos": p["pos"].to(dev, bf).contiguous(),
                  "pre_ln_w": p["pre_ln_w"].to(dev, bf), "pre_ln_b": p["pre_ln_b"].to(dev, bf),
                  "layers": [{k: v.to(dev, bf).contiguous() for k, v in L.items()} for L in p["layers"][: max(self.extract_stage)]]}
        self._src = p  # full host copy (all 24 layers) so that FINAL.pt["rgb_ckpt"] can be written back unchanged

    def init_random(self, seed: int = 0) -> None:
        g = torch.Generator(device=self.device).manual_seed(seed)
        dev, bf, d, ff = self.device, torch.bfloat16, self.dim, self.ff

        def rn(*shape, std=0.02, mean=0.0):
            return (torch.randn(*shape, device=dev, generator=g) * std + mean).to(bf)

        pw = torch.zeros(d, self.kp, device=dev, dtype=bf)
        pw[:, : 3 * self.patch ** 2] = rn(d, 3 * self.patch ** 2)
        self.p = {"patch_w": pw, "cls": rn(d), "pos": rn(self.n_patch + 1, d), "pre_ln_w": rn(d, std=0.05, mean=1.0),
                  "pre_ln_b": rn(d), "layers": []}
        for _ in range(max(self.extract_stage)):
            self.p["layers"].append({
                "ln1_w": rn(d, std=0.05, mean=1.0), "ln1_b": rn(d), "qkv_w": rn(3 * d, d), "qkv_b": rn(3 * d),
                "o_w": rn(d, d), "o_b": rn(d), "ln2_w": rn(d, std=0.05, mean=1.0), "ln2_b": rn(d),
                "fc1_w": rn(ff, d), "fc1_b": rn(ff), "fc2_w": rn(d, ff), "fc2_b": rn(d)})

    def export_params(self) -> Dict:
        """Engine-layout host copy (what was loaded; for random init: the 22 layers that exist)."""
        if getattr(self, "_src", None) is not None:
            return self._src
        P2 = 3 * self.patch ** 2
        out = {k: v.float().cpu() for k, v in self.p.items() if torch.is_tensor(v)}
        out["patch_w"] = out["patch_w"][:, :P2].reshape(self.dim, 3, self.patch, self.patch)
        out["layers"] = [{k: v.float().cpu() for k, v in L.items()} for L in self.p["layers"]]
        return out

    def load_state_dict(self, sd, strict: bool = False):
        """VisionModal.load_state_dict with the reference's key names (`encoder.vision_model.*`, UniBind.py:96-100)."""
        from .checkpoint import vit_from_hf
        self.load_params(vit_from_hf(sd))

    def _desc(self, B: int) -> torch.Tensor:
        if B not in self._desc_cache:
            n = self.n_patch + 1
            self._desc_cache[B] = hk.make_desc([(b * n, n, b * n, n) for b in range(B)], self.device)
        return self._desc_cache[B]

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def encode(self, rgb: torch.Tensor) -> torch.Tensor:
        """rgb [B,3,224,224] float -> [B, 3*256, 1024] bf16 (the three taps without CLS)."""
        p, d, H = self.p, self.dim, self.heads
        B = rgb.shape[0]
        side = int(round(self.n_patch ** 0.5)) * self.patch
        if rgb.dim() != 4 or rgb.shape[1] != 3 or rgb.shape[2] != side or rgb.shape[3] != side:
            raise ValueError(f"VisionModal.encode expects [B, 3, {side}, {side}] pixel_values (CLIPImageProcessor output), got {tuple(rgb.shape)}")
        n = self.n_patch + 1
        rgb = rgb.to(self.device, torch.float32)
        x = hk.gemm_nt(hk.patchify(rgb, self.patch, self.kp), p["patch_w"])
        x = hk.vit_assemble(x, p["cls"], p["pos"], B, self.n_patch, d)
        x = hk.layernorm_fwd(x, p["pre_ln_w"], p["pre_ln_b"])
        desc = self._desc(B)
        LT = hk.pad64(n)
        out = torch.empty((B, len(self.extract_stage) * self.n_patch, d), device=self.device, dtype=torch.bfloat16)
        o = torch.empty((B * n, d), device=self.device, dtype=torch.bfloat16)
        scale = (d // H) ** -0.5
        native = os.environ.get("LHRS_NATIVE_LAYER", "1") != "0"   # one library call per encoder layer (lhrs_vit_layer_forward: same launches, same order)
        if native:
            h, qkv = torch.empty_like(x), torch.empty((B * n, 3 * d), device=self.device, dtype=torch.bfloat16)
            f = torch.empty((B * n, self.ff), device=self.device, dtype=torch.bfloat16)
        for li, L in enumerate(p["layers"]):
            if native:
                hk.vit_layer_forward(x, L, desc, B, n, LT, H, self.ff, h, qkv, o, f)
            else:
                h = hk.layernorm_fwd(x, L["ln1_w"], L["ln1_b"])
                qkv = hk.gemm_nt(h, L["qkv_w"], bias=L["qkv_b"])
                hk.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, None, desc, B, H, d // H, n, n, LT, False, scale)
                x = hk.gemm_nt(o, L["o_w"], bias=L["o_b"], residual=x, out=x)
                h = hk.layernorm_fwd(x, L["ln2_w"], L["ln2_b"], out=h)
                f = hk.gemm_nt(h, L["fc1_w"], bias=L["fc1_b"], act=hk.ACT_QUICK_GELU)
                x = hk.gemm_nt(f, L["fc2_w"], bias=L["fc2_b"], residual=x, out=x)
            if li + 1 in self.extract_stage:  # hidden_states[li+1]; drop CLS, place tap g at rows [g*256, (g+1)*256)
                gi = self.extract_stage.index(li + 1)
                row_b = d * 2
                hk.copy_2d(out.data_ptr() + gi * self.n_patch * row_b, out.shape[1] * row_b, x.data_ptr() + row_b, n * row_b,
                           self.n_patch * row_b, B)
        return out

    def __call__(self, data):  # BaseModal.forward dispatch (lhrs/models/base_modal.py:52-64)
        return self.encode(data["rgb"] if isinstance(data, dict) else data)

    def parameters(self) -> List[torch.Tensor]:
        out = [v for k, v in self.p.items() if torch.is_tensor(v)]
        for L in self.p.get("layers", []):
            out += list(L.values())
        return out
