"""TEST INFRASTRUCTURE ONLY - restatement of the reference's `bits: 4` base weights: bitsandbytes 4-bit storage (nf4 / fp4, double_quant).

PARITY UNPINNED: bitsandbytes==0.41.3 (pyproject.toml of the reference) is absent from this image and from /root/reference; the call site is
lhrs/models/text_modal.py:91-107 (`BitsAndBytesConfig(load_in_4bit=config.bits == 4, bnb_4bit_compute_dtype=compute_dtype,
bnb_4bit_use_double_quant=config.double_quant, bnb_4bit_quant_type=config.quant_type)`; every shipped YAML carries `double_quant: True`,
`quant_type: nf4`).  What follows restates the published algorithm (Dettmers et al., "QLoRA: Efficient Finetuning of Quantized LLMs", 2023, §3
and `bitsandbytes.functional.quantize_4bit / dequantize_4bit / quantize_blockwise / create_dynamic_map` + csrc/kernels.cu `dQuantizeNF4`,
`dQuantizeFP4`, `dDequantizeNF4`, `dDequantizeFP4Tree`, `dQuantize<0>` of the 0.41 series), in numpy fp32:

  * the weight of ONE nn.Linear, flattened row-major, in blocks of 64: absmax_b = max |w|; code = Q(w * (1 / absmax_b)), Q = the comparison tree of
    the data type - NF4: 16 quantiles of N(0, 1) normalised to [-1, 1], decision thresholds at the midpoints; FP4: sign bit + 3 bits with the
    magnitudes {0, 1/192, 1/6, 1/4, 1/3, 1/2, 2/3, 1}; two codes per byte, the first element in the high nibble;
  * double_quant: offset = mean(absmax); absmax - offset in blocks of 256 -> nearest entry of the signed "dynamic" 8-bit table (a halving search
    with strict comparisons against the midpoint) + one fp32 absmax2 per block; dequantised: table[q] * absmax2 + offset;
  * every product (MatMul4Bit forward and backward): x . dequant(W)^T with dequant(W) = table4[code] * absmax_b cast to the compute dtype.
    There is no 4-bit arithmetic: a model whose 16-bit weights are replaced by `dequantize_4bit(quantize_4bit(W))` computes what the package does.
"""
from __future__ import annotations

import numpy as np

BLOCK = 64          # bnb_4bit blocksize (quantize_4bit default)
BLOCK2 = 256        # blocksize of the nested statistics quantisation

NF4_LEVEL = np.array([-1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453, -0.28444138169288635, -0.18477343022823334,
                      -0.09105003625154495, 0.0, 0.07958029955625534, 0.16093020141124725, 0.24611230194568634, 0.33791524171829224,
                      0.44070982933044434, 0.5626170039176941, 0.7229568362236023, 1.0], dtype=np.float32)
# dQuantizeNF4's decision values, ascending; the tree returns the number of them that x exceeds
NF4_THR = np.array([-0.8480964004993439, -0.6106329262256622, -0.4599952697753906, -0.33967943489551544, -0.23460740596055984,
                    -0.13791173323988914, -0.045525018125772476, 0.03979014977812767, 0.1202552504837513, 0.2035212516784668, 0.2920137718319893,
                    0.3893125355243683, 0.5016634166240692, 0.6427869200706482, 0.8614784181118011], dtype=np.float32)
# dDequantizeFP4Tree: value of the 3 magnitude bits (x absmax x sign); dQuantizeFP4: thresholds on |x| and the code each interval maps to
FP4_MAG = np.array([0.0, 5.208333333e-03, 0.66666667, 1.0, 0.33333333, 0.5, 0.16666667, 0.25], dtype=np.float32)
FP4_THR = np.array([0.00260417, 0.0859375, 0.20833333, 0.29166667, 0.4166667, 0.583333, 0.8333333], dtype=np.float32)
FP4_CODE = np.array([0b000, 0b001, 0b110, 0b111, 0b100, 0b101, 0b010, 0b011], dtype=np.uint8)
FP4_LEVEL = np.concatenate([FP4_MAG, -FP4_MAG]).astype(np.float32)


def dynamic_map() -> np.ndarray:
    """functional.create_dynamic_map(signed=True, max_exponent_bits=7, total_bits=8): 127 positive + 127 negative values + 0 + 1, sorted."""
    import torch
    data = []
    for i in range(7):
        boundaries = torch.linspace(0.1, 1, 2 ** i + 1)
        means = (boundaries[:-1] + boundaries[1:]) / 2.0
        data += ((10 ** (-6 + i)) * means).tolist()
        data += (-(10 ** (-6 + i)) * means).tolist()
    data.append(0)
    data.append(1.0)
    assert len(data) == 256
    data.sort()
    return np.asarray(data, dtype=np.float32)


def _q4(v: np.ndarray, quant_type: str) -> np.ndarray:
    """The comparison trees: strict `>` against every threshold (a NaN - all-zero block, 0 * inf - exceeds none)."""
    if quant_type == "nf4":
        return (v[..., None] > NF4_THR).sum(-1).astype(np.uint8)
    if quant_type == "fp4":
        sign = np.where(v < 0, 0b1000, 0).astype(np.uint8)
        c = (np.abs(v)[..., None] > FP4_THR).sum(-1)
        return FP4_CODE[c] + sign
    raise ValueError(quant_type)


def _q_dynamic(code: np.ndarray, x: np.ndarray) -> np.ndarray:
    """dQuantize<0>: pivot 127, steps 64..1 over the sorted table, then the nearer of the pivot and the neighbour bound on x's side."""
    n = x.shape[0]
    pivot = np.full(n, 127, dtype=np.int64)
    upper_pivot = np.full(n, 255, dtype=np.int64)
    lower_pivot = np.zeros(n, dtype=np.int64)
    lower = np.full(n, -1.0, dtype=np.float32)
    upper = np.full(n, 1.0, dtype=np.float32)
    val = code[pivot]
    with np.errstate(invalid="ignore"):
        i = 64
        while i > 0:
            gt = x > val
            lower_pivot = np.where(gt, pivot, lower_pivot); lower = np.where(gt, val, lower)
            upper_pivot = np.where(gt, upper_pivot, pivot); upper = np.where(gt, upper, val)
            pivot = np.where(gt, pivot + i, pivot - i)
            val = code[pivot]
            i >>= 1
        upper = np.where(upper_pivot == 255, code[255], upper)
        lower = np.where(lower_pivot == 0, code[0], lower)
        gt = x > val
        mid_up = ((upper + val) * np.float32(0.5)).astype(np.float32)
        mid_lo = ((lower + val) * np.float32(0.5)).astype(np.float32)
        out = np.where(gt, np.where(x > mid_up, upper_pivot, pivot), np.where(x < mid_lo, lower_pivot, pivot))
    return out.astype(np.uint8)


def quantize_4bit(w: np.ndarray, quant_type: str = "nf4", double_quant: bool = True) -> dict:
    """w: the fp32 VALUES of one Linear's weight (any shape, numel % 64 == 0) -> state dict with `packed` uint8 [numel / 2] and either
    `absmax` fp32 [numel / 64] or its nested form (`qabsmax` uint8, `absmax2` fp32 [ceil(nb / 256)], `offset` float)."""
    x = np.ascontiguousarray(w, dtype=np.float32).reshape(-1, BLOCK)
    absmax = np.abs(x).max(axis=1).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = (np.float32(1.0) / absmax).astype(np.float32)
        codes = _q4((x * inv[:, None]).astype(np.float32), quant_type).reshape(-1)
    st = dict(packed=((codes[0::2] << 4) | codes[1::2]).astype(np.uint8), quant_type=quant_type, shape=tuple(w.shape))
    if not double_quant:
        st["absmax"] = absmax
        return st
    offset = np.float32(absmax.astype(np.float64).mean())     # the package: fp32 mean on the device (reduction order unspecified)
    rest = (absmax - offset).astype(np.float32)
    nb = rest.shape[0]
    pad = (-nb) % BLOCK2
    r = np.concatenate([rest, np.zeros(pad, dtype=np.float32)]).reshape(-1, BLOCK2)
    absmax2 = np.abs(r).max(axis=1).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        v = (r * (np.float32(1.0) / absmax2).astype(np.float32)[:, None]).astype(np.float32).reshape(-1)[:nb]
    st.update(qabsmax=_q_dynamic(dynamic_map(), v), absmax2=absmax2, offset=float(offset))
    return st


def absmax_of(st: dict) -> np.ndarray:
    if "absmax" in st:
        return st["absmax"]
    code = dynamic_map()
    q = st["qabsmax"]
    blk = np.arange(q.shape[0]) // BLOCK2
    return (code[q] * st["absmax2"][blk] + np.float32(st["offset"])).astype(np.float32)


def dequantize_4bit(st: dict) -> np.ndarray:
    """-> fp32 array of the stored shape: table[code] * absmax (the caller casts to the compute dtype, one rounding)."""
    level = NF4_LEVEL if st["quant_type"] == "nf4" else FP4_LEVEL
    p = st["packed"]
    codes = np.empty(p.shape[0] * 2, dtype=np.uint8)
    codes[0::2], codes[1::2] = p >> 4, p & 15
    a = absmax_of(st)
    return (level[codes].reshape(-1, BLOCK) * a[:, None]).astype(np.float32).reshape(st["shape"])


def fake_quant_weight(w, quant_type: str = "nf4", double_quant: bool = True, parts: int = 1):
    """torch weight [N, K] made of `parts` row-concatenated reference Linears (q|k|v, gate|up) -> dequantize_4bit(quantize_4bit(.)) of each, rounded
    to bf16 (the compute dtype) and returned as fp32: what every product of the 4-bit model multiplies with."""
    import torch
    rows = w.shape[0] // parts
    out = []
    for i in range(parts):
        sub = w[i * rows:(i + 1) * rows].detach().float().numpy()
        out.append(torch.from_numpy(dequantize_4bit(quantize_4bit(sub, quant_type, double_quant))).to(torch.bfloat16).float())
    return torch.cat(out, 0)
