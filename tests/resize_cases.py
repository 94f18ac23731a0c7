"""Deterministic source images for the texture-reduction parity cases (tests/golden/tier_k_resize.npz):
shared by oracle/gen_golden.py (which feeds them to the reference) and tests/test_host_cpu.py."""
import numpy as np


def resize_case_input(name):
    """Deterministic source images for the > 1024 reduction cases: (height, width, channels) uint8."""
    sizes = {"2048x16_half": (16, 2048), "1100x40": (40, 1100), "37x1300": (1300, 37), "1030x60": (60, 1030),
             "1500x90_rgba": (90, 1500), "3000x7": (7, 3000), "1025x33_grey": (33, 1025)}
    h, w = sizes[name]
    rng = np.random.default_rng(sum(map(ord, name)))
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 3 + yy) % 256, (yy * 11 + xx // 3) % 256, ((xx ^ yy) * 5) % 256], -1).astype(np.uint8)
    noise = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    mask = ((xx // 64 + yy // 8) % 2 == 0)[..., None]
    img = np.where(mask, img, noise)
    img[:, : w // 7] = 255 * ((xx[:, : w // 7, None] // 2) % 2)            # hard black / white stripes (ringing, saturation)
    if name.endswith("_rgba"):
        return np.dstack([img, ((xx * 7 + yy * 13) % 256).astype(np.uint8)])
    if name.endswith("_grey"):
        return img[..., 1]
    return img


RESIZE_CASES = ["2048x16_half", "1100x40", "37x1300", "1030x60", "1500x90_rgba", "3000x7", "1025x33_grey"]
