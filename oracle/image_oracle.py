"""CPU restatement of the image half of the data boundary.  TEST INFRASTRUCTURE ONLY.

The reference builds its training / evaluation transform as `CLIPImageProcessor.from_pretrained(vit_name)`
(/root/reference lhrs/Dataset/build_transform.py:43-45) and calls `.preprocess(image, return_tensors="pt")["pixel_values"]`
(lhrs/Dataset/cap_dataset.py, cli_qa.py).  The arithmetic lives in two third-party dependencies that are not in the tree:

  * transformers (pinned 4.36.1; 5.15.0 here - same pipeline): convert RGB -> resize shortest edge to 224 (long edge
    int(224 * long / short)), PIL BICUBIC -> center crop 224 (top = (h-224)//2, left = (w-224)//2) ->
    float32(float64(u8) * (1/255)) -> (x - mean) / std in float32, CLIP mean/std;
  * Pillow `Image.resize(..., BICUBIC)` on 8-bit images = libImaging/Resample.c: two separable passes (horizontal, then
    vertical, each rounding to uint8), per output sample a window [xmin, xmin+xmax) of the anti-aliased cubic (a = -0.5,
    support 2 * max(1, in/out)), coefficients normalised in double and fixed to 22 fractional bits, accumulator started at 2^21,
    result clip((acc >> 22), 0, 255).

Pinned bit-exactly (uint8 stage and float32 output) against Pillow + HF's own CLIPImageProcessor run in the build container:
tests/golden/clip_preprocess.npz (tests/golden/make_golden_image.py).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the whole axis (box = [0, in_size))."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)      # C cast: truncation toward zero
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    """one 8-bit resampling pass along `axis` of an [H, W, C] uint8 image"""
    in_size = img.shape[axis]
    bounds, kk = precompute_coeffs(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], dtype=np.uint8)
    for xx in range(out_size):
        xmin, xmax = bounds[xx]
        acc = np.tensordot(kk[xx, :xmax].astype(np.int64), src[xmin:xmin + xmax], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_u8(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """PIL Image.resize((out_w, out_h), BICUBIC) for an [H, W, 3] uint8 array (ImagingResample: horizontal pass, then vertical;
    a pass whose size does not change is skipped)."""
    if img.shape[1] != out_w:
        img = _pass(img, out_w, 1)
    if img.shape[0] != out_h:
        img = _pass(img, out_h, 0)
    return img


def resized_size(h: int, w: int, short: int = 224):
    """HF get_resize_output_image_size(size=short, default_to_square=False) -> (new_h, new_w)"""
    if h <= w:
        return short, int(short * w / h)
    return int(short * h / w), short


def normalise_lut():
    """float32 value of every (channel, byte): float32(float64(b) * (1/255)) -> (x - mean) / std in float32"""
    r = (np.arange(256, dtype=np.float64) * 0.00392156862745098).astype(np.float32)
    m = np.array(CLIP_MEAN, dtype=np.float32)
    s = np.array(CLIP_STD, dtype=np.float32)
    return ((r[None, :] - m[:, None]) / s[:, None]).astype(np.float32)  # [3, 256]


def clip_preprocess(img: np.ndarray, crop: int = 224):
    """[H, W, 3] uint8 -> (uint8 crop [crop, crop, 3], float32 pixel_values [3, crop, crop])"""
    h, w = img.shape[:2]
    nh, nw = resized_size(h, w, crop)
    r = resize_u8(img, nw, nh)
    top, left = (nh - crop) // 2, (nw - crop) // 2
    c = r[top:top + crop, left:left + crop]
    lut = normalise_lut()
    out = np.stack([lut[ch][c[:, :, ch]] for ch in range(3)], 0)
    return c, out


def cls_eval_preprocess(img: np.ndarray, short: int = 256, crop: int = 224):
    """The evaluation transform of the classification caller (/root/reference lhrs/Dataset/build_transform.py:27-40):
    torchvision `Resize(256, BICUBIC)` -> `CenterCrop(224)` -> `ToTensor()` -> `Normalize(IMAGENET mean, std)` on a PIL image.

    torchvision is absent from this image AND from /root/reference (pyproject pins torchvision==0.16.2): PARITY UNPINNED - restated from
    torchvision 0.16's functional API: `resize` of a PIL image = `img.resize((new_w, new_h), BICUBIC)` with the short edge -> 256 and the
    long edge int(256 * long / short) (an image whose short edge already is 256 is returned as is); `center_crop` takes
    top = int(round((h - 224) / 2.0)), left likewise (Python rounding: half to even); `to_tensor` = uint8 -> float32, `.div(255)`;
    `normalize` = `(x - mean) / std` in float32.  The resize itself is Pillow's (restated and pinned above).
    [H, W, 3] uint8 -> float32 [3, 224, 224]."""
    h, w = img.shape[:2]
    nh, nw = resized_size(h, w, short)
    r = resize_u8(img, nw, nh)
    top, left = int(round((nh - crop) / 2.0)), int(round((nw - crop) / 2.0))
    c = r[top:top + crop, left:left + crop].astype(np.float32)
    x = (c / np.float32(255.0)).astype(np.float32)
    mean = np.array((0.485, 0.456, 0.406), dtype=np.float32)
    std = np.array((0.229, 0.224, 0.225), dtype=np.float32)
    return np.ascontiguousarray(((x - mean) / std).astype(np.float32).transpose(2, 0, 1))
