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


class HipMem:
    """The few HIP runtime calls a test needs for device buffers of its own (plain hipMalloc / hipMemcpy)."""

    def __init__(self):
        import ctypes as C
        self.C = C
        try:
            self.lib = C.CDLL("libamdhip64.so")
        except OSError:
            self.lib = C.CDLL("/opt/rocm/lib/libamdhip64.so")
        self.lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.lib.hipFree.argtypes = [C.c_void_p]
        self.lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.lib.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.lib.hipDeviceSynchronize.argtypes = []

    def malloc(self, n):
        p = self.C.c_void_p()
        assert self.lib.hipMalloc(self.C.byref(p), n) == 0
        return p.value

    def free(self, p):
        assert self.lib.hipFree(self.C.c_void_p(p)) == 0

    def h2d(self, dst, arr):
        arr = np.ascontiguousarray(arr)
        assert self.lib.hipMemcpy(self.C.c_void_p(dst), self.C.c_void_p(arr.ctypes.data), arr.nbytes, 1) == 0

    def d2h(self, arr, src):
        assert self.lib.hipMemcpy(self.C.c_void_p(arr.ctypes.data), self.C.c_void_p(src), arr.nbytes, 2) == 0

    def memset(self, p, v, n):
        assert self.lib.hipMemset(self.C.c_void_p(p), v, n) == 0

    def sync(self):
        assert self.lib.hipDeviceSynchronize() == 0

    def mem_free(self):
        """Free device memory in bytes (hipMemGetInfo)."""
        free, total = self.C.c_size_t(), self.C.c_size_t()
        assert self.lib.hipMemGetInfo(self.C.byref(free), self.C.byref(total)) == 0
        return free.value
