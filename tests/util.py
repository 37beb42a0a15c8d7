"""Shared helpers of the parity tests."""
import numpy as np


def rgb8(img):
    """RGB8 quantisation used by the parity gate: floor(clamp(c,0,1)*255 + 0.5)."""
    return np.floor(np.clip(np.nan_to_num(img, nan=0.0), 0.0, 1.0) * 255.0 + 0.5).astype(np.int32)


def psnr_rgb8(a, b):
    d = (rgb8(a) - rgb8(b)).astype(np.float64)
    mse = np.mean(d * d)
    return float("inf") if mse == 0 else 10.0 * np.log10(255.0 * 255.0 / mse)


def ulp_diff(a, b):
    """Distance in units of last place between float32 arrays (same-sign finite values)."""
    a = np.ascontiguousarray(a, dtype=np.float32); b = np.ascontiguousarray(b, dtype=np.float32)
    ia = a.view(np.int32).astype(np.int64); ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia); ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


def assert_image_parity(got_rgb, ref_rgb, min_frac=0.999, min_psnr=50.0):
    """north-star gate: RGB8 within +-1 LSB on >= 99.9 % of pixels and PSNR >= 50 dB."""
    d = np.abs(rgb8(got_rgb) - rgb8(ref_rgb)).max(axis=-1)
    frac = float((d <= 1).mean())
    p = psnr_rgb8(got_rgb, ref_rgb)
    assert frac >= min_frac, f"only {frac:.5f} of pixels within 1 LSB"
    assert p >= min_psnr, f"PSNR {p:.2f} dB"
    return frac, p
