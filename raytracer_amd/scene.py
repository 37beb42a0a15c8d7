"""Scene model at the boundary: the flattened (structure-of-arrays) form of the reference's
`Scene` (scene.h:143-153) and a loader for the reference's `.scene` JSON files that
reproduces loadScene (scene.cpp:15-135) including its quirks.

Everything is numpy float32 evaluated in the reference's operation order so that the arrays
hold the same bits scene.cpp would leave in its structs.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from dataclasses import dataclass, field

import numpy as np

from . import _abi

f32 = np.float32


def _f32(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    if shape is not None:
        a = a.reshape(shape)
    return a


def cross_f32(x, y):
    """glm::cross on float32 rows: (x.y*y.z - y.y*x.z, x.z*y.x - y.z*x.x, x.x*y.y - y.x*x.y)."""
    x = _f32(x); y = _f32(y)
    return np.stack([x[..., 1] * y[..., 2] - y[..., 1] * x[..., 2],
                     x[..., 2] * y[..., 0] - y[..., 2] * x[..., 0],
                     x[..., 0] * y[..., 1] - y[..., 0] * x[..., 1]], axis=-1).astype(np.float32)


def dot_f32(a, b):
    a = _f32(a); b = _f32(b)
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def normalize_f32(v):
    """glm::normalize: v * (1 / sqrt(dot(v,v))), each step rounded to float32.  A zero vector (the cross product of a
    zero-area triangle's edges, scene.cpp:91) gives 0 * inf = NaN exactly as the reference's loader stores it; such a
    triangle has det = 0 for every ray and is never hit, so that normal is never read by the tracer."""
    v = _f32(v)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv = (f32(1.0) / np.sqrt(dot_f32(v, v))).astype(np.float32)
        return (v * inv[..., None]).astype(np.float32)


@dataclass
class FlatScene:
    """Flattened scene; field meanings are those of hipexec_scene_desc."""
    tri_p: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.float32))
    tri_e1: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.float32))
    tri_e2: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.float32))
    tri_norm: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.float32))
    tri_uv: np.ndarray = field(default_factory=lambda: np.zeros((0, 6), np.float32))
    tri_mat: np.ndarray = field(default_factory=lambda: np.zeros((0,), np.int32))
    sph_center: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.float32))
    sph_radius: np.ndarray = field(default_factory=lambda: np.zeros((0,), np.float32))
    sph_mat: np.ndarray = field(default_factory=lambda: np.zeros((0,), np.int32))
    mat_color: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.float32))
    mat_kd: np.ndarray = field(default_factory=lambda: np.zeros((0,), np.float32))
    mat_ks: np.ndarray = field(default_factory=lambda: np.zeros((0,), np.float32))
    mat_hard: np.ndarray = field(default_factory=lambda: np.zeros((0,), np.float32))
    mat_kr: np.ndarray = field(default_factory=lambda: np.zeros((0,), np.float32))
    mat_tex: np.ndarray = field(default_factory=lambda: np.zeros((0,), np.int32))
    # refractionFactor / refractionHardness (scene.h:28-29): never loaded by the reference; None = all zero.  Only the
    # optional, non-reference refraction mode reads them (hardness = relative index of refraction).
    mat_kt: np.ndarray | None = None
    mat_ior: np.ndarray | None = None
    textures: list = field(default_factory=list)       # list of uint8 arrays [h][w][4], rows top-down
    lamp_pos: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.float32))
    lamp_intensity: np.ndarray = field(default_factory=lambda: np.zeros((0,), np.float32))
    lamp_distance: np.ndarray = field(default_factory=lambda: np.zeros((0,), np.float32))
    cam_pos: np.ndarray = field(default_factory=lambda: np.zeros((3,), np.float32))
    cam_mat: np.ndarray = field(default_factory=lambda: np.eye(3, dtype=np.float32))   # JSON transform rows
    horizon: np.ndarray = field(default_factory=lambda: np.full((3,), 0.3, np.float32))
    ambient: np.ndarray = field(default_factory=lambda: np.zeros((3,), np.float32))
    ambient_factor: float = 0.0                # worldAmbientFactor (scene.cpp:30): only the optional AO mode reads it

    # ------------------------------------------------------------------------------------
    @property
    def n_tri(self): return int(np.size(self.tri_p) // 3)

    @property
    def n_sph(self): return int(np.size(self.sph_radius))

    @property
    def n_lamp(self): return int(np.size(self.lamp_intensity))

    def normalised(self) -> "FlatScene":
        """Coerce dtypes/shapes/contiguity in place and return self."""
        n = np.size(np.asarray(self.tri_p, dtype=np.float32)) // 3
        self.tri_p = _f32(self.tri_p, (n, 3)); self.tri_e1 = _f32(self.tri_e1, (n, 3))
        self.tri_e2 = _f32(self.tri_e2, (n, 3)); self.tri_norm = _f32(self.tri_norm, (n, 3))
        if self.tri_uv is None or np.size(self.tri_uv) == 0:
            self.tri_uv = np.zeros((n, 6), np.float32)
        self.tri_uv = _f32(self.tri_uv, (n, 6))
        self.tri_mat = np.ascontiguousarray(self.tri_mat, dtype=np.int32).reshape(n)
        ns = np.size(self.sph_radius)
        self.sph_center = _f32(self.sph_center, (ns, 3)); self.sph_radius = _f32(self.sph_radius, (ns,))
        self.sph_mat = np.ascontiguousarray(self.sph_mat, dtype=np.int32).reshape(ns)
        nm = np.size(self.mat_kd)
        self.mat_color = _f32(self.mat_color, (nm, 3))
        for k in ("mat_kd", "mat_ks", "mat_hard", "mat_kr"):
            setattr(self, k, _f32(getattr(self, k), (nm,)))
        self.mat_tex = np.ascontiguousarray(self.mat_tex, dtype=np.int32).reshape(nm)
        for k in ("mat_kt", "mat_ior"):
            if getattr(self, k) is not None:
                setattr(self, k, _f32(getattr(self, k), (nm,)))
        nl = np.size(self.lamp_intensity)
        self.lamp_pos = _f32(self.lamp_pos, (nl, 3))
        self.lamp_intensity = _f32(self.lamp_intensity, (nl,)); self.lamp_distance = _f32(self.lamp_distance, (nl,))
        self.cam_pos = _f32(self.cam_pos, (3,)); self.cam_mat = _f32(self.cam_mat, (3, 3))
        self.horizon = _f32(self.horizon, (3,)); self.ambient = _f32(self.ambient, (3,))
        self.textures = [np.ascontiguousarray(t, dtype=np.uint8) for t in self.textures]
        for t in self.textures:
            assert t.ndim == 3 and t.shape[2] == 4, "textures are [h][w][4] RGBA8"
        return self

    def to_desc(self):
        """Build a hipexec_scene_desc.  Returns (desc, keepalive); keep `keepalive` referenced
        for as long as the descriptor is in use."""
        self.normalised()
        d = _abi.SceneDesc()
        keep = []

        def fp(a):
            keep.append(a)
            return a.ctypes.data_as(C.POINTER(C.c_float))

        def ip(a):
            keep.append(a)
            return a.ctypes.data_as(C.POINTER(C.c_int32))

        d.n_tri = self.n_tri
        d.tri_p, d.tri_e1, d.tri_e2, d.tri_norm = fp(self.tri_p), fp(self.tri_e1), fp(self.tri_e2), fp(self.tri_norm)
        d.tri_uv, d.tri_mat = fp(self.tri_uv), ip(self.tri_mat)
        d.n_sph = self.n_sph
        d.sph_center, d.sph_radius, d.sph_mat = fp(self.sph_center), fp(self.sph_radius), ip(self.sph_mat)
        d.n_mat = int(self.mat_kd.shape[0])
        d.mat_color, d.mat_kd, d.mat_ks = fp(self.mat_color), fp(self.mat_kd), fp(self.mat_ks)
        d.mat_hard, d.mat_kr, d.mat_tex = fp(self.mat_hard), fp(self.mat_kr), ip(self.mat_tex)
        if self.mat_kt is not None:
            d.mat_kt = fp(self.mat_kt)
        if self.mat_ior is not None:
            d.mat_ior = fp(self.mat_ior)
        d.n_tex = len(self.textures)
        offs, ws, hs, blobs, off = [], [], [], [], 0
        for t in self.textures:
            offs.append(off); hs.append(t.shape[0]); ws.append(t.shape[1])
            blobs.append(t.reshape(-1)); off += t.size
        atlas = np.concatenate(blobs) if blobs else np.zeros((4,), np.uint8)
        a_off = np.asarray(offs if offs else [0], dtype=np.uint64)
        a_w = np.asarray(ws if ws else [0], dtype=np.uint32)
        a_h = np.asarray(hs if hs else [0], dtype=np.uint32)
        keep += [atlas, a_off, a_w, a_h]
        d.tex_rgba = atlas.ctypes.data_as(C.POINTER(C.c_uint8))
        d.tex_offset = a_off.ctypes.data_as(C.POINTER(C.c_uint64))
        d.tex_w = a_w.ctypes.data_as(C.POINTER(C.c_uint32))
        d.tex_h = a_h.ctypes.data_as(C.POINTER(C.c_uint32))
        d.n_lamp = self.n_lamp
        d.lamp_pos, d.lamp_intensity, d.lamp_distance = fp(self.lamp_pos), fp(self.lamp_intensity), fp(self.lamp_distance)
        d.cam_pos[:] = [float(x) for x in self.cam_pos]
        d.cam_mat[:] = [float(x) for x in self.cam_mat.reshape(-1)]
        d.horizon[:] = [float(x) for x in self.horizon]
        d.ambient[:] = [float(x) for x in self.ambient]
        d.ambient_factor = float(np.float32(self.ambient_factor))
        return d, keep


def assemble_triangles(s: FlatScene, verts, idx, uv_raw, has_uv, tmat, tex_scale):
    """Vectorised body of the face loop of loadScene (scene.cpp:61-94): p, e1 = v1 - p,
    e2 = v2 - p, UVs relative to the first corner and pre-multiplied by
    (texW * scaleX, texH * scaleY) for textured materials, norm = normalize(cross(e1, e2)).
    `s.textures` and `s.mat_tex` must already be set; tex_scale is [n_mat][2]."""
    verts = _f32(verts, (-1, 3))
    idx = np.asarray(idx, dtype=np.int64).reshape(-1, 3)
    n = idx.shape[0]
    p = verts[idx[:, 0]]
    e1 = (verts[idx[:, 1]] - p).astype(np.float32)                  # scene.cpp:62-64
    e2 = (verts[idx[:, 2]] - p).astype(np.float32)
    uv = np.zeros((n, 6), np.float32)
    if n:
        uv_raw = _f32(uv_raw, (n, 3, 2))
        hu = np.asarray(has_uv, bool).reshape(n)
        uv[hu, 0:2] = uv_raw[hu, 0]                                 # scene.cpp:68-75
        uv[hu, 2:4] = uv_raw[hu, 1] - uv_raw[hu, 0]
        uv[hu, 4:6] = uv_raw[hu, 2] - uv_raw[hu, 0]
        tmat = np.asarray(tmat, dtype=np.int32).reshape(n)
        mtex = np.asarray(s.mat_tex, dtype=np.int32).reshape(-1)
        for mi in range(mtex.shape[0]):                             # scene.cpp:77-86
            if mtex[mi] < 0:
                continue
            t = s.textures[mtex[mi]]
            sx = f32(t.shape[1]) * f32(tex_scale[mi][0])
            sy = f32(t.shape[0]) * f32(tex_scale[mi][1])
            sel = tmat == mi
            uv[np.ix_(sel, [0, 2, 4])] *= sx
            uv[np.ix_(sel, [1, 3, 5])] *= sy
    s.tri_p, s.tri_e1, s.tri_e2, s.tri_uv = p, e1, e2, uv
    s.tri_mat = np.asarray(tmat, dtype=np.int32).reshape(n)
    s.tri_norm = normalize_f32(cross_f32(e1, e2)) if n else np.zeros((0, 3), np.float32)   # scene.cpp:91
    return s


# ----------------------------------------------------------------------------------------
def decode_png_rgba8(path: str) -> np.ndarray:
    """Role of lodepng::decode in tex_image::createImage (tex_image.h:25-31): RGBA8, rows top-down."""
    from PIL import Image
    with Image.open(path) as im:
        return np.ascontiguousarray(np.asarray(im.convert("RGBA"), dtype=np.uint8))


def scene_from_json(doc: dict, base_dir: str = ".", texture_loader=decode_png_rgba8) -> FlatScene:
    """loadScene (scene.cpp:15-135) on an already parsed JSON document."""
    s = FlatScene()
    world = doc.get("world")
    if world is not None:                                           # scene.cpp:24-35
        s.ambient = _f32(world["ambientColor"])
        s.horizon = _f32(world["horizonColor"])
        s.ambient_factor = float(np.float32(world.get("ambientFactor", 0.0)))
    else:
        s.ambient = _f32([0.0, 0.0, 0.0]); s.horizon = _f32([0.3, 0.3, 0.3])

    mats = doc.get("materials") or []                               # scene.cpp:37-51
    col, kd, ks, hard, kr, mtex, scale = [], [], [], [], [], [], []
    for m in mats:
        col.append(m["diffusiveColor"]); kd.append(m["diffusiveFactor"]); ks.append(m["specularFactor"])
        hard.append(m["specularHardness"]); kr.append(m["reflectionFactor"])
        if m.get("imagePath") is not None:
            p = m["imagePath"]
            if not os.path.isabs(p):
                p = os.path.join(base_dir, p)
            s.textures.append(texture_loader(p))
            mtex.append(len(s.textures) - 1)
            scale.append((f32(m["scaleX"]), f32(m["scaleY"])))
        else:
            mtex.append(-1); scale.append((f32(1), f32(1)))
    s.mat_color, s.mat_kd, s.mat_ks, s.mat_hard, s.mat_kr = col, kd, ks, hard, kr
    s.mat_tex = np.asarray(mtex, dtype=np.int32)

    verts = _f32(doc.get("vertices") or [], (-1, 3))                # scene.cpp:53-59
    faces = doc.get("faces") or []
    n = len(faces)
    idx = np.asarray([f["vertices"] for f in faces], dtype=np.int64).reshape(n, 3)
    uv_raw = np.zeros((n, 3, 2), np.float32)
    has_uv = np.zeros((n,), bool)
    tmat = np.full((n,), -1, np.int32)
    for i, f in enumerate(faces):
        if f.get("uv") is not None:
            uv_raw[i] = _f32(f["uv"], (3, 2)); has_uv[i] = True
        if f.get("material") is not None:
            tmat[i] = int(f["material"])
    tex_scale = np.asarray(scale, dtype=np.float32).reshape(-1, 2)
    assemble_triangles(s, verts, idx, uv_raw, has_uv, tmat, tex_scale)

    sph = doc.get("spheres") or []                                  # scene.cpp:96-109: centre parsed into int
    s.sph_center = _f32([[int(c) for c in sp["center"]] for sp in sph], (-1, 3))
    s.sph_radius = _f32([sp["radius"] for sp in sph])
    s.sph_mat = np.asarray([(-1 if sp.get("material") is None else int(sp["material"])) for sp in sph], dtype=np.int32)

    lamps = doc.get("lamps") or []                                  # scene.cpp:111-119: pos parsed into int
    s.lamp_pos = _f32([[int(c) for c in lp["pos"]] for lp in lamps], (-1, 3))
    s.lamp_intensity = _f32([lp["intensity"] for lp in lamps])
    s.lamp_distance = _f32([lp["distance"] for lp in lamps])

    s.cam_pos = _f32(doc["translate"])                              # scene.cpp:121-134
    s.cam_mat = _f32(doc["transform"], (3, 3))
    return s.normalised()


def load_scene(path: str) -> FlatScene:
    """loadScene(Scene&, path) (scene.cpp:15-22)."""
    with open(path, "r") as fh:
        doc = json.load(fh)
    return scene_from_json(doc, base_dir=os.path.dirname(os.path.abspath(path)))
