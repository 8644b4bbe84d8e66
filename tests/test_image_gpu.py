"""lhrs_clip_preprocess (HIP) vs the Pillow/HF-pinned oracle: bit-exact float32 pixel_values."""
import hashlib
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from image_cases import CASES, make_image  # noqa: E402
from lhrs_bot_amd import kernels as hk  # noqa: E402
from lhrs_bot_amd.data import CLIPImageProcessorHIP  # noqa: E402
from oracle import image_oracle as IO  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_golden_cases_bit_exact():
    Z = np.load(os.path.join(GOLD, "clip_preprocess.npz"))
    for i, (h, w) in enumerate(CASES):
        img = make_image(100 + i, h, w)
        out = hk.clip_preprocess(torch.from_numpy(img).cuda()).cpu().numpy()
        assert hashlib.sha256(np.ascontiguousarray(out).tobytes()).hexdigest() == str(Z[f"f32_sha_{i}"]), (i, h, w)


def test_random_sizes_against_oracle_and_processor_surface():
    rng = np.random.default_rng(5)
    sizes = [(224, 225), (225, 224), (1, 1), (2, 900), (900, 2), (223, 223), (1500, 2000), (640, 640), (31, 57)]
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    proc = CLIPImageProcessorHIP()
    pv = proc.preprocess(imgs, return_tensors="pt")["pixel_values"]
    assert pv.shape == (len(sizes), 3, 224, 224) and pv.dtype == torch.float32 and pv.is_cuda
    for b, im in enumerate(imgs):
        _, want = IO.clip_preprocess(im)
        assert np.array_equal(pv[b].cpu().numpy(), want), sizes[b]
    # non-contiguous rows (a crop view of a larger device image) go through row_stride
    big = torch.from_numpy(rng.integers(0, 256, (400, 500, 3), dtype=np.uint8)).cuda()
    view = big[10:310, :, :]
    assert np.array_equal(hk.clip_preprocess(view).cpu().numpy(), IO.clip_preprocess(view.cpu().numpy())[1])
    with pytest.raises(ValueError):
        proc.preprocess(np.zeros((8, 8), dtype=np.uint8))
