"""Seeded input images shared by tests/golden/make_golden_image.py (which runs Pillow + HF on them) and the parity tests."""
import numpy as np

CASES = [(300, 400), (400, 300), (224, 224), (100, 150), (480, 640), (333, 500), (224, 600), (1024, 768), (57, 31)]


def make_image(seed, h, w):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if seed % 2:  # smooth + noise: exercises rounding at many different accumulator values
        yy, xx = np.mgrid[0:h, 0:w]
        base = ((np.sin(xx / 7.0)[..., None] * 60 + np.cos(yy / 5.0)[..., None] * 60 + 128) + (base % 24)).clip(0, 255).astype(np.uint8)
    return base
