"""Python binding of the C-ABI HIP executor (include/hip_executor.h).

`HipExecutor` plays the role of the reference's `OpenClExecutor` (opencl_executor.h:12-95):
constructed from a scene, it owns the device copy of the geometry and answers closest-hit /
any-hit / trace / render requests.  There is no CPU fallback: if the library is missing or a
call fails, a RuntimeError is raised (the reference throws std::runtime_error,
opencl_executor.h:67-72).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi
from ._abi import HIPEXEC_MODE_CL, HIPEXEC_MODE_CPU, HIPEXEC_TILE_ROWS
from .scene import FlatScene


class HipExecutorError(RuntimeError):
    pass


def n_tiles(height: int, tile_first: int = 0, tile_step: int = 1) -> int:
    total = (height + HIPEXEC_TILE_ROWS - 1) // HIPEXEC_TILE_ROWS
    if tile_first >= total:
        return 0
    return (total - tile_first + tile_step - 1) // tile_step


class HipExecutor:
    def __init__(self, scene: FlatScene, device: int = -1, _clone_of: "HipExecutor | None" = None):
        self._lib = _abi.load_library()
        self.scene = scene
        self._h = C.c_void_p()
        if _clone_of is not None:
            self._check(self._lib.hipexec_clone(C.byref(self._h), _clone_of._h))
            return
        desc, keep = scene.to_desc()
        self._check(self._lib.hipexec_create(C.byref(self._h), C.byref(desc), int(device)))
        del keep

    def clone(self) -> "HipExecutor":
        """A further executor of the same scene that shares this one's device copy of it (hipexec_clone)."""
        return HipExecutor(self.scene, _clone_of=self)

    def set_camera(self, cam_pos, cam_mat):
        """Camera of later calls: position [3] and matrix [3][3] (rows = JSON transform rows)."""
        p = np.ascontiguousarray(cam_pos, dtype=np.float32).reshape(3)
        m = np.ascontiguousarray(cam_mat, dtype=np.float32).reshape(9)
        self._check(self._lib.hipexec_set_camera(self._h, p.ctypes.data_as(C.POINTER(C.c_float)), m.ctypes.data_as(C.POINTER(C.c_float))))

    def optimize(self) -> float:
        """Rebuilds the scene's hierarchy with the surface-area topology (hipexec_optimize), for every executor that
        shares the scene; same results, fewer node visits.  Returns the milliseconds the rebuild took (0.0 when there
        was nothing to do)."""
        ms = C.c_float(0.0)
        self._check(self._lib.hipexec_optimize(self._h, C.byref(ms)))
        return float(ms.value)

    # -- plumbing -------------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            msg = self._lib.hipexec_last_error().decode("utf-8", "replace")
            raise HipExecutorError(f"hip_executor error {rc}: {msg}")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.hipexec_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_stream(self, stream_ptr: int | None):
        self._check(self._lib.hipexec_set_stream(self._h, C.c_void_p(stream_ptr or 0)))

    def sync(self):
        self._check(self._lib.hipexec_sync(self._h))

    def set_lanes(self, lanes: int):
        """Concurrent lanes a frame is spread over (default 2); 1 for executors that alternate frames (FramesInFlight)."""
        self._check(self._lib.hipexec_set_lanes(self._h, int(lanes)))

    # -- frame ----------------------------------------------------------------------------
    @staticmethod
    def _cfg(width, height, mode, max_depth, tone, tile_first, tile_step, time_kernels, algo=None, bilinear=False, refract=False, ao=False):
        # time_kernels: False / True (every kernel class) / an iterable of class indices (only those classes)
        flags = 0
        if time_kernels is True:
            flags = _abi.HIPEXEC_FLAG_TIME_KERNELS
        elif time_kernels:
            flags = _abi.HIPEXEC_FLAG_TIME_KERNELS
            for k in time_kernels:
                flags |= 1 << (_abi.HIPEXEC_FLAG_TIME_CLASS_SHIFT + int(k))
        if bilinear:
            flags |= _abi.HIPEXEC_FLAG_BILINEAR
        if refract:
            flags |= _abi.HIPEXEC_FLAG_REFRACT
        if ao:
            flags |= _abi.HIPEXEC_FLAG_AO
        if algo is not None:
            flags |= {"scan": _abi.HIPEXEC_FLAG_SCAN, "bvh": _abi.HIPEXEC_FLAG_BVH}[algo]
        return _abi.RenderCfg(int(width), int(height), int(mode), int(max_depth), int(bool(tone)), int(tile_first),
                              int(tile_step), flags)

    def set_algorithm(self, algo: str):
        """Algorithm of calls that do not choose one: "bvh" (default) or "scan" (index-order scans)."""
        self._check(self._lib.hipexec_set_algorithm(
            self._h, {"scan": _abi.HIPEXEC_ALGO_SCAN, "bvh": _abi.HIPEXEC_ALGO_BVH}[algo]))

    def render(self, width, height, mode=HIPEXEC_MODE_CPU, max_depth=0, tone=True, tile_first=0, tile_step=1,
               want_ids=True, time_kernels=False, out=None, algo=None, bilinear=False, refract=False, ao=False):
        """Render to host memory.  -> dict(rgb [H][W][3] float32, ids [H][W][spp] int32 | None, stats)."""
        spp = 1 if mode == HIPEXEC_MODE_CL else 4
        rgb = out if out is not None else np.zeros((height, width, 3), np.float32)
        assert rgb.dtype == np.float32 and rgb.flags.c_contiguous and rgb.shape == (height, width, 3)
        ids = np.full((height, width, spp), -3, np.int32) if want_ids else None
        st = _abi.Stats()
        cfg = self._cfg(width, height, mode, max_depth, tone, tile_first, tile_step, time_kernels, algo, bilinear, refract, ao)
        self._check(self._lib.hipexec_render(self._h, C.byref(cfg), rgb.ctypes.data,
                                             ids.ctypes.data if ids is not None else None, C.byref(st)))
        return {"rgb": rgb, "ids": ids, "stats": st}

    def render_rect(self, width, height, x, y, w, h, out, mode=HIPEXEC_MODE_CPU, max_depth=0, tone=True, ids=None, algo=None):
        """Render the pixels [x, x+w) x [y, y+h) of the width x height frame into `out` ([height][width][3] float32, the
        whole frame's image; the rest is left untouched).  -> Stats"""
        assert out.dtype == np.float32 and out.flags.c_contiguous and out.shape == (height, width, 3)
        st = _abi.Stats()
        cfg = self._cfg(width, height, mode, max_depth, tone, 0, 1, False, algo)
        self._check(self._lib.hipexec_render_rect(self._h, C.byref(cfg), int(x), int(y), int(w), int(h), out.ctypes.data,
                                                  ids.ctypes.data if ids is not None else None, C.byref(st)))
        return st

    def render_progressive(self, width, height, rows_per_step=48, on_rows=None, mode=HIPEXEC_MODE_CPU, max_depth=0,
                           tone=True, out=None, algo=None):
        """Render to host memory in steps of `rows_per_step` rows; `on_rows(first_row, row_count)` is called
        after each step with those rows already in the image.  -> dict(rgb, stats)."""
        rgb = out if out is not None else np.zeros((height, width, 3), np.float32)
        assert rgb.dtype == np.float32 and rgb.flags.c_contiguous and rgb.shape == (height, width, 3)
        st = _abi.Stats()
        cfg = self._cfg(width, height, mode, max_depth, tone, 0, 1, False, algo)
        cb = _abi.ROWS_CB((lambda user, y0, rows: on_rows(int(y0), int(rows))) if on_rows else (lambda user, y0, rows: None))
        self._check(self._lib.hipexec_render_progressive(self._h, C.byref(cfg), int(rows_per_step), rgb.ctypes.data,
                                                         C.cast(cb, C.c_void_p), None, C.byref(st)))
        return {"rgb": rgb, "stats": st}

    def render_device(self, width, height, d_out_rgba: int, d_out_ids: int | None = None, mode=HIPEXEC_MODE_CPU,
                      max_depth=0, tone=True, tile_first=0, tile_step=1, time_kernels=False, stats=None, algo=None):
        """Render into device memory (raw device pointers, e.g. torch.Tensor.data_ptr()).
        Asynchronous; call sync() before reading `stats`."""
        cfg = self._cfg(width, height, mode, max_depth, tone, tile_first, tile_step, time_kernels, algo)
        self._check(self._lib.hipexec_render_device(self._h, C.byref(cfg), C.c_void_p(d_out_rgba),
                                                    C.c_void_p(d_out_ids or 0),
                                                    C.byref(stats) if stats is not None else None))

    # -- batches (ray_tracer_cl.h:29-54) --------------------------------------------------
    @staticmethod
    def _rays(rays6):
        r = np.ascontiguousarray(rays6, dtype=np.float32).reshape(-1, 6)
        return r

    def closest_hits(self, rays6, mode=HIPEXEC_MODE_CL):
        r = self._rays(rays6)
        out = np.zeros(r.shape[0], dtype=_abi.HIT_DTYPE)
        self._check(self._lib.hipexec_closest_hits(self._h, int(mode), r.ctypes.data, r.shape[0], out.ctypes.data))
        return out

    def any_hits(self, rays6, mode=HIPEXEC_MODE_CL, hits=None):
        r = self._rays(rays6)
        if hits is None:
            hits = np.zeros(r.shape[0], dtype=np.int8)
        assert hits.dtype == np.int8 and hits.flags.c_contiguous and hits.shape[0] == r.shape[0]
        self._check(self._lib.hipexec_any_hits(self._h, int(mode), r.ctypes.data, r.shape[0], hits.ctypes.data))
        return hits

    def any_hits_static(self, d_rays6: int, n: int, d_hits: int, mode=HIPEXEC_MODE_CL):
        """any_hits on device pointers through the library's fixed work memory: allocates nothing, asynchronous
        on the executor's stream (sync() before reading d_hits), at most _abi.STATIC_MAX_RAYS rays."""
        self._check(self._lib.hipexec_any_hits_static(self._h, int(mode), C.c_void_p(d_rays6), int(n), C.c_void_p(d_hits)))

    def trace_rays(self, rays6, mode=HIPEXEC_MODE_CL, max_depth=0, colours=None, want_stats=False):
        r = self._rays(rays6)
        if colours is None:
            colours = np.zeros((r.shape[0], 3), np.float32)
        assert colours.dtype == np.float32 and colours.flags.c_contiguous and colours.shape == (r.shape[0], 3)
        st = _abi.Stats()
        self._check(self._lib.hipexec_trace_rays(self._h, int(mode), int(max_depth), r.ctypes.data, r.shape[0],
                                                 colours.ctypes.data, C.byref(st)))
        return (colours, st) if want_stats else colours


def trim(device: int = -1):
    """Hands what the library's pool of device memory, pinned memory and streams holds for `device` (-1: all) back to the
    driver (hipexec_destroy keeps them for the next executor)."""
    lib = _abi.load_library()
    rc = lib.hipexec_trim(int(device))
    if rc != 0:
        raise RuntimeError(f"hipexec_trim: {rc} ({lib.hipexec_last_error().decode()})")


class FramesInFlight:
    """Throughput mode: `depth` executors of one scene (one uploaded, the others its clones: they share the scene's device
    memory), one lane each, take device-resident frames in turn.

    submit() enqueues a frame (asynchronous) on the next executor and returns the slot it used; collect(slot) waits
    for that frame and returns its Stats.  A slot must be collected before submit() reaches it again; submit() does it
    itself if the caller has not.  With depth 2 the thinly populated deep recursion levels of one frame run beside the
    camera and first-bounce launches of the next.  The frames are the ones a single executor renders, bit for bit.
    Unless `optimize` is false the scene's hierarchy is rebuilt once with the surface-area topology (optimize_ms)."""

    def __init__(self, scene: FlatScene, device: int = -1, depth: int = 2, lanes: int | None = None, optimize: bool = True):
        first = HipExecutor(scene, device=device)
        # a scene that is kept for many frames gets the better tree (hipexec_optimize, once for all the clones)
        self.optimize_ms = first.optimize() if optimize else 0.0
        self.ex = [first] + [first.clone() for _ in range(depth - 1)]      # one device copy of the scene for all of them
        for e in self.ex:
            e.set_lanes(lanes if lanes else (1 if depth > 1 else 2))
        self.stats = [_abi.Stats() for _ in range(depth)]
        self.busy = [False] * depth
        self.next = 0

    def submit(self, width, height, d_out_rgba: int, d_out_ids: int | None = None, **kw) -> int:
        k = self.next
        self.next = (k + 1) % len(self.ex)
        if self.busy[k]:
            self.collect(k)
        self.ex[k].render_device(width, height, d_out_rgba, d_out_ids, stats=self.stats[k], **kw)
        self.busy[k] = True
        return k

    def collect(self, slot: int):
        if self.busy[slot]:
            self.ex[slot].sync()
            self.busy[slot] = False
        return self.stats[slot]

    def drain(self):
        for k in range(len(self.ex)):
            self.collect(k)

    def close(self):
        for e in self.ex:
            e.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
