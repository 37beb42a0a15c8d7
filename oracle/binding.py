"""ctypes binding of the CPU oracle (oracle/librt_oracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py, never by
the raytracer_amd package.  Parity is unpinned (see rt_oracle.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

from raytracer_amd import _abi

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_DIR, "librt_oracle.so")
_lib = None


def build():
    subprocess.run(["make", "-C", _DIR, "-s"], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        fp = C.POINTER(C.c_float)
        L.orc_triangle_hit.restype = C.c_int
        L.orc_triangle_hit.argtypes = [C.c_int, fp, fp, fp, fp, fp, fp, fp]
        L.orc_sphere_hit.restype = C.c_int
        L.orc_sphere_hit.argtypes = [C.c_int, fp, C.c_float, fp, fp, fp, fp, fp]
        L.orc_tex_bilinear.restype = None
        L.orc_tex_bilinear.argtypes = [C.POINTER(_abi.SceneDesc), C.c_int, C.c_float, C.c_float, fp]
        L.orc_tex_get.restype = None
        L.orc_tex_get.argtypes = [C.POINTER(_abi.SceneDesc), C.c_int, C.c_uint, C.c_uint, fp]
        L.orc_closest_hits.restype = None
        L.orc_closest_hits.argtypes = [C.POINTER(_abi.SceneDesc), C.c_int, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_any_hits.restype = None
        L.orc_any_hits.argtypes = [C.POINTER(_abi.SceneDesc), C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(_abi.Stats)]
        L.orc_trace_rays.restype = None
        L.orc_trace_rays.argtypes = [C.POINTER(_abi.SceneDesc), C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(_abi.Stats)]
        L.orc_render_region.restype = None
        L.orc_render_region.argtypes = [C.POINTER(_abi.SceneDesc), C.POINTER(_abi.RenderCfg), C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(_abi.Stats), C.c_int]
        L.orc_render_blocks.restype = None
        L.orc_render_blocks.argtypes = [C.POINTER(_abi.SceneDesc), C.POINTER(_abi.RenderCfg), C.c_int, C.c_void_p,
                                        C.POINTER(_abi.Stats), C.c_int]
        _lib = L
    return _lib


def _f3(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def triangle_hit(mode, p, e1, e2, o, d):
    """-> (isHit, t, u, v, point)"""
    keep = [_f3(x) for x in (p, e1, e2, o, d)]
    tuv = np.zeros(3, np.float32); pt = np.zeros(3, np.float32)
    h = lib().orc_triangle_hit(mode, *[k[1] for k in keep], tuv.ctypes.data_as(C.POINTER(C.c_float)),
                               pt.ctypes.data_as(C.POINTER(C.c_float)))
    return bool(h), tuv[0], tuv[1], tuv[2], pt


def sphere_hit(mode, c, r, o, d):
    """-> (isHit, t, point, norm)"""
    keep = [_f3(x) for x in (c, o, d)]
    t = C.c_float(0); pt = np.zeros(3, np.float32); n = np.zeros(3, np.float32)
    h = lib().orc_sphere_hit(mode, keep[0][1], C.c_float(float(np.float32(r))), keep[1][1], keep[2][1], C.byref(t),
                             pt.ctypes.data_as(C.POINTER(C.c_float)), n.ctypes.data_as(C.POINTER(C.c_float)))
    return bool(h), np.float32(t.value), pt, n


class Oracle:
    """CPU oracle bound to one flattened scene."""

    def __init__(self, flat_scene):
        self.scene = flat_scene
        self.desc, self._keep = flat_scene.to_desc()

    def tex_get(self, tex, x, y):
        out = np.zeros(4, np.float32)
        lib().orc_tex_get(C.byref(self.desc), int(tex), C.c_uint(int(x) & 0xFFFFFFFF), C.c_uint(int(y) & 0xFFFFFFFF),
                          out.ctypes.data_as(C.POINTER(C.c_float)))
        return out

    def tex_bilinear(self, tex, ux, uy):
        out = np.zeros(4, np.float32)
        lib().orc_tex_bilinear(C.byref(self.desc), int(tex), C.c_float(ux), C.c_float(uy), out.ctypes.data_as(C.POINTER(C.c_float)))
        return out

    def closest_hits(self, rays6, mode=_abi.HIPEXEC_MODE_CPU):
        rays = np.ascontiguousarray(rays6, dtype=np.float32).reshape(-1, 6)
        out = np.zeros(rays.shape[0], dtype=_abi.HIT_DTYPE)
        lib().orc_closest_hits(C.byref(self.desc), mode, rays.ctypes.data, rays.shape[0], out.ctypes.data)
        return out

    def any_hits(self, rays6, mode=_abi.HIPEXEC_MODE_CPU, hits=None, stats=None):
        rays = np.ascontiguousarray(rays6, dtype=np.float32).reshape(-1, 6)
        if hits is None:
            hits = np.zeros(rays.shape[0], dtype=np.int8)
        lib().orc_any_hits(C.byref(self.desc), mode, rays.ctypes.data, rays.shape[0], hits.ctypes.data,
                           C.byref(stats) if stats is not None else None)
        return hits

    def trace_rays(self, rays6, mode=_abi.HIPEXEC_MODE_CPU, max_depth=0, colours=None, stats=None):
        rays = np.ascontiguousarray(rays6, dtype=np.float32).reshape(-1, 6)
        if colours is None:
            colours = np.zeros((rays.shape[0], 3), np.float32)
        lib().orc_trace_rays(C.byref(self.desc), mode, max_depth, rays.ctypes.data, rays.shape[0], colours.ctypes.data,
                             C.byref(stats) if stats is not None else None)
        return colours

    def render_region(self, width, height, x, y, w, h, mode=_abi.HIPEXEC_MODE_CPU, max_depth=0, tone=1, n_threads=1,
                      bilinear=False, refract=False, ao=False):
        """-> dict(rgb [h][w][3], linear [h][w][3], ids [h][w][spp], stats)"""
        cfg = _abi.RenderCfg(width, height, mode, max_depth, tone, 0, 1,
                             (_abi.HIPEXEC_FLAG_BILINEAR if bilinear else 0) | (_abi.HIPEXEC_FLAG_REFRACT if refract else 0) | (_abi.HIPEXEC_FLAG_AO if ao else 0))
        spp = 1 if mode == _abi.HIPEXEC_MODE_CL else 4
        rgb = np.zeros((h, w, 3), np.float32); lin = np.zeros((h, w, 3), np.float32)
        ids = np.full((h, w, spp), -2, np.int32)
        st = _abi.Stats()
        lib().orc_render_region(C.byref(self.desc), C.byref(cfg), x, y, w, h, rgb.ctypes.data, lin.ctypes.data,
                                ids.ctypes.data, C.byref(st), n_threads)
        return {"rgb": rgb, "linear": lin, "ids": ids, "stats": st}

    def render_blocks(self, width, height, blocks_xywh, mode=_abi.HIPEXEC_MODE_CPU, max_depth=0, n_threads=1):
        """Trace the pixels of a list of (x, y, w, h) rectangles, colours discarded -> Stats."""
        cfg = _abi.RenderCfg(width, height, mode, max_depth, 1, 0, 1, 0)
        blk = np.ascontiguousarray(blocks_xywh, dtype=np.int32).reshape(-1, 4)
        st = _abi.Stats()
        lib().orc_render_blocks(C.byref(self.desc), C.byref(cfg), blk.shape[0], blk.ctypes.data, C.byref(st), n_threads)
        return st

    def render(self, width, height, **kw):
        return self.render_region(width, height, 0, 0, width, height, **kw)
