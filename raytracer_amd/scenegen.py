"""Procedural scenes in the reference's `.scene` schema (scene.cpp:15-135).

The reference ships no sample scene and its exporter (scene_exporter.py) needs Blender, so
the benchmark and test scenes are generated here.  A `SceneSpec` holds the JSON-level
description (vertices, faces, materials with in-memory textures, integer-or-fractional
sphere/lamp coordinates, camera) and can be emitted either as a JSON document
(`to_json_doc`, textures written as PNG files) or directly as a `FlatScene` (`to_flat`,
vectorised, same arithmetic as the loader).
"""
from __future__ import annotations

import json
import math
import os
from dataclasses import dataclass, field

import numpy as np

from .scene import FlatScene, assemble_triangles, scene_from_json, _f32


@dataclass
class SceneSpec:
    vertices: np.ndarray                       # [nv][3] float32
    faces: np.ndarray                          # [nf][3] int
    face_uv: np.ndarray | None = None          # [nf][3][2] float32 or None
    face_has_uv: np.ndarray | None = None      # [nf] bool
    face_mat: np.ndarray | None = None         # [nf] int, -1 = null
    materials: list = field(default_factory=list)   # dicts with the JSON keys; "texture": uint8 [h][w][4] or None
    spheres: list = field(default_factory=list)     # dicts center/radius/material (as written to JSON)
    lamps: list = field(default_factory=list)       # dicts pos/intensity/distance
    transform: np.ndarray = field(default_factory=lambda: np.eye(3, dtype=np.float32))
    translate: np.ndarray = field(default_factory=lambda: np.zeros(3, np.float32))
    world: dict | None = None

    def to_flat(self) -> FlatScene:
        s = FlatScene()
        if self.world is not None:
            s.ambient = _f32(self.world["ambientColor"]); s.horizon = _f32(self.world["horizonColor"])
            s.ambient_factor = float(np.float32(self.world.get("ambientFactor", 0.0)))
        else:
            s.ambient = _f32([0, 0, 0]); s.horizon = _f32([0.3, 0.3, 0.3])
        scale = []
        mtex = []
        for m in self.materials:
            if m.get("texture") is not None:
                s.textures.append(np.ascontiguousarray(m["texture"], dtype=np.uint8))
                mtex.append(len(s.textures) - 1)
                scale.append((np.float32(m["scaleX"]), np.float32(m["scaleY"])))
            else:
                mtex.append(-1); scale.append((np.float32(1), np.float32(1)))
        s.mat_color = [m["diffusiveColor"] for m in self.materials]
        s.mat_kd = [m["diffusiveFactor"] for m in self.materials]
        s.mat_ks = [m["specularFactor"] for m in self.materials]
        s.mat_hard = [m["specularHardness"] for m in self.materials]
        s.mat_kr = [m["reflectionFactor"] for m in self.materials]
        if any(m.get("refractionFactor", 0.0) for m in self.materials):
            s.mat_kt = _f32([m.get("refractionFactor", 0.0) for m in self.materials])
            s.mat_ior = _f32([m.get("refractionHardness", 1.0) for m in self.materials])
        s.mat_tex = np.asarray(mtex, dtype=np.int32)
        nf = self.faces.shape[0]
        uv = self.face_uv if self.face_uv is not None else np.zeros((nf, 3, 2), np.float32)
        has = self.face_has_uv if self.face_has_uv is not None else np.full((nf,), self.face_uv is not None)
        fm = self.face_mat if self.face_mat is not None else np.full((nf,), -1, np.int32)
        assemble_triangles(s, self.vertices, self.faces, uv, has, fm,
                           np.asarray(scale, dtype=np.float32).reshape(-1, 2))
        s.sph_center = _f32([[int(c) for c in sp["center"]] for sp in self.spheres], (-1, 3))
        s.sph_radius = _f32([sp["radius"] for sp in self.spheres])
        s.sph_mat = np.asarray([(-1 if sp.get("material") is None else int(sp["material"])) for sp in self.spheres], dtype=np.int32)
        s.lamp_pos = _f32([[int(c) for c in lp["pos"]] for lp in self.lamps], (-1, 3))
        s.lamp_intensity = _f32([lp["intensity"] for lp in self.lamps])
        s.lamp_distance = _f32([lp["distance"] for lp in self.lamps])
        s.cam_pos = _f32(self.translate); s.cam_mat = _f32(self.transform, (3, 3))
        return s.normalised()

    def to_json_doc(self, texture_dir: str | None = None) -> dict:
        """JSON document in the reference schema.  Textured materials need `texture_dir`
        (PNG files are written there and referenced by absolute path like the exporter does)."""
        mats = []
        for i, m in enumerate(self.materials):
            d = {k: (list(map(float, m[k])) if k == "diffusiveColor" else float(m[k]))
                 for k in ("diffusiveColor", "diffusiveFactor", "specularFactor", "specularHardness", "reflectionFactor")}
            if m.get("texture") is not None:
                assert texture_dir is not None, "textured scene needs a texture_dir"
                from PIL import Image
                os.makedirs(texture_dir, exist_ok=True)
                p = os.path.abspath(os.path.join(texture_dir, f"tex{i}.png"))
                Image.fromarray(np.asarray(m["texture"], dtype=np.uint8), "RGBA").save(p)
                d["imagePath"] = p; d["scaleX"] = float(m["scaleX"]); d["scaleY"] = float(m["scaleY"])
            else:
                d["imagePath"] = None
            mats.append(d)
        nf = self.faces.shape[0]
        faces = []
        for i in range(nf):
            has = (self.face_uv is not None) and (self.face_has_uv is None or bool(self.face_has_uv[i]))
            fm = None if self.face_mat is None or int(self.face_mat[i]) < 0 else int(self.face_mat[i])
            faces.append({"vertices": [int(v) for v in self.faces[i]], "material": fm,
                          "uv": ([[float(a) for a in c] for c in self.face_uv[i]] if has else None)})
        return {
            "world": self.world,
            "materials": mats,
            "vertices": [[float(a) for a in v] for v in np.asarray(self.vertices, dtype=np.float32)],
            "faces": faces,
            "spheres": self.spheres, "lamps": self.lamps,
            "transform": [[float(a) for a in r] for r in np.asarray(self.transform, dtype=np.float32).reshape(3, 3)],
            "translate": [float(a) for a in np.asarray(self.translate, dtype=np.float32)],
        }

    def write(self, path: str):
        doc = self.to_json_doc(texture_dir=os.path.dirname(os.path.abspath(path)))
        with open(path, "w") as fh:
            json.dump(doc, fh)
        return path


# ----------------------------------------------------------------------------------------
def material(color, kd=0.8, ks=0.5, hard=50.0, kr=0.0, texture=None, scale=(1.0, 1.0), kt=0.0, ior=1.0):
    """kt / ior: refractionFactor / refractionHardness (scene.h:28-29).  The reference never loads them, so they are not
    written to the JSON document; to_flat() passes them on for the optional refraction mode."""
    return {"diffusiveColor": [float(c) for c in color], "diffusiveFactor": float(kd), "specularFactor": float(ks),
            "specularHardness": float(hard), "reflectionFactor": float(kr),
            "texture": texture, "scaleX": float(scale[0]), "scaleY": float(scale[1]),
            "refractionFactor": float(kt), "refractionHardness": float(ior)}


def camera(pos, pitch_deg=0.0, yaw_deg=0.0):
    """Camera looking down -Z, +Y up, rotated by yaw about Y then pitch about X.  Returns
    (transform rows, translate) as the exporter would write them: world = transform @ cam."""
    p, y = math.radians(pitch_deg), math.radians(yaw_deg)
    rx = np.array([[1, 0, 0], [0, math.cos(p), -math.sin(p)], [0, math.sin(p), math.cos(p)]])
    ry = np.array([[math.cos(y), 0, math.sin(y)], [0, 1, 0], [-math.sin(y), 0, math.cos(y)]])
    return (ry @ rx).astype(np.float32), np.asarray(pos, dtype=np.float32)


def checker_texture(size, rng, cells=16, alpha_lo=128):
    """Square RGBA8 checker + noise, alpha in {alpha_lo, 255}."""
    yy, xx = np.mgrid[0:size, 0:size]
    chk = ((xx * cells // size) + (yy * cells // size)) & 1
    base = np.where(chk[..., None] == 1, np.array([220, 200, 60]), np.array([40, 90, 200])).astype(np.int32)
    noise = rng.integers(-30, 31, size=(size, size, 3))
    rgb = np.clip(base + noise, 0, 255).astype(np.uint8)
    a = np.where(((xx // max(1, size // 8)) + (yy // max(1, size // 4))) & 1, 255, alpha_lo).astype(np.uint8)
    return np.ascontiguousarray(np.concatenate([rgb, a[..., None]], axis=-1))


def grid_mesh(nx, nz, x0, x1, z0, z1, height_fn=None, uv_tiles=1.0):
    """Heightfield of nx*nz cells = 2*nx*nz triangles with normals facing +Y, per-corner UVs."""
    xs = np.linspace(x0, x1, nx + 1, dtype=np.float32)
    zs = np.linspace(z0, z1, nz + 1, dtype=np.float32)
    X, Z = np.meshgrid(xs, zs, indexing="xy")            # [nz+1][nx+1]
    Y = np.zeros_like(X) if height_fn is None else height_fn(X, Z).astype(np.float32)
    verts = np.stack([X, Y, Z], axis=-1).reshape(-1, 3).astype(np.float32)
    U = ((X - x0) / (x1 - x0) * uv_tiles).astype(np.float32)
    Vv = ((Z - z0) / (z1 - z0) * uv_tiles).astype(np.float32)
    uvv = np.stack([U, Vv], axis=-1).reshape(-1, 2)
    iz, ix = np.mgrid[0:nz, 0:nx]
    v00 = (iz * (nx + 1) + ix).reshape(-1)
    v10 = v00 + 1
    v01 = v00 + (nx + 1)
    v11 = v01 + 1
    # (v00, v01, v10): e1 = +z, e2 = +x -> cross(e1,e2) = +y
    f1 = np.stack([v00, v01, v10], axis=-1)
    f2 = np.stack([v10, v01, v11], axis=-1)
    faces = np.stack([f1, f2], axis=1).reshape(-1, 3)
    return verts, faces, uvv[faces].astype(np.float32)


def box_mesh(lo, hi):
    lo = np.asarray(lo, np.float32); hi = np.asarray(hi, np.float32)
    c = np.array([[lo[0], lo[1], lo[2]], [hi[0], lo[1], lo[2]], [hi[0], hi[1], lo[2]], [lo[0], hi[1], lo[2]],
                  [lo[0], lo[1], hi[2]], [hi[0], lo[1], hi[2]], [hi[0], hi[1], hi[2]], [lo[0], hi[1], hi[2]]], np.float32)
    q = [(0, 3, 2, 1), (4, 5, 6, 7), (0, 1, 5, 4), (2, 3, 7, 6), (1, 2, 6, 5), (0, 4, 7, 3)]
    faces = []
    for a, b, cc, d in q:
        faces += [(a, b, cc), (a, cc, d)]
    return c, np.asarray(faces, dtype=np.int64)


def _merge(meshes):
    """meshes: list of (verts, faces, uv or None, mat index)."""
    V, F, UV, HAS, M, off = [], [], [], [], [], 0
    for verts, faces, uv, mi in meshes:
        V.append(verts); F.append(faces + off); off += verts.shape[0]
        nf = faces.shape[0]
        UV.append(uv if uv is not None else np.zeros((nf, 3, 2), np.float32))
        HAS.append(np.full((nf,), uv is not None)); M.append(np.full((nf,), mi, np.int32))
    return (np.concatenate(V).astype(np.float32), np.concatenate(F), np.concatenate(UV).astype(np.float32),
            np.concatenate(HAS), np.concatenate(M))


# ----------------------------------------------------------------------------------------
def scene_c1(seed=1, plane_cells=22, fractional=False) -> SceneSpec:
    """Config 1: ground plane + box (~1k triangles), 8 spheres, 1 lamp, grey horizon."""
    rng = np.random.default_rng(seed)
    gv, gf, _ = grid_mesh(plane_cells, plane_cells, -11, 11, -16, 6)
    bv, bf = box_mesh((-6, 0, -9), (-4, 2, -7))
    mats = [material((0.7, 0.7, 0.65), kd=0.8, ks=0.2, hard=20), material((0.8, 0.2, 0.2), kd=0.9, ks=0.6, hard=60)]
    sph = []
    for k in range(8):
        mats.append(material(rng.uniform(0.2, 0.9, 3), kd=0.8, ks=0.7, hard=float(rng.integers(10, 120))))
        r = float(np.float32(rng.uniform(0.5, 1.5)))
        cx, cz = int(rng.integers(-7, 8)), int(rng.integers(-12, 2))
        c = [cx + (0.7 if fractional else 0), (1 if r <= 1 else 2) + (0.2 if fractional else 0), cz - (0.9 if fractional else 0)]
        sph.append({"center": c, "radius": r, "material": 2 + k})
    V, F, UV, HAS, M = _merge([(gv, gf, None, 0), (bv, bf, None, 1)])
    T, pos = camera((0, 5, 14), pitch_deg=-17)
    lamp = [{"pos": [4 + (0.6 if fractional else 0), 9, 3], "intensity": 1.0, "distance": 25.0}]
    return SceneSpec(V, F, None, None, M, mats, sph, lamp, T, pos,
                     {"ambientColor": [0.05, 0.05, 0.06], "horizonColor": [0.45, 0.5, 0.6], "ambientFactor": 0.0})


def scene_c2(seed=2, cells=224, tex_size=1024) -> SceneSpec:
    """Config 2/3: textured heightfield (2*cells^2 triangles, 100 352 at cells=224) with a
    reflective ground (kr 0.2), 4 mirror spheres (kr 0.5), 2 lamps."""
    rng = np.random.default_rng(seed)
    ph = rng.uniform(0, 6.28, 4)

    def hf(X, Z):
        return (0.35 * np.sin(0.6 * X + ph[0]) * np.cos(0.5 * Z + ph[1]) + 0.15 * np.sin(1.7 * X + ph[2]) * np.sin(1.3 * Z + ph[3]))

    gv, gf, guv = grid_mesh(cells, cells, -12, 12, -20, 4, hf, uv_tiles=6.0)
    tex = checker_texture(tex_size, rng)
    mats = [material((0.6, 0.6, 0.6), kd=0.8, ks=0.3, hard=30, kr=0.2, texture=tex, scale=(1.0, 1.0))]
    sph = []
    for k, (cx, cz) in enumerate([(-5, -8), (-1, -11), (3, -7), (6, -13)]):
        mats.append(material(rng.uniform(0.3, 0.9, 3), kd=0.6, ks=0.8, hard=80, kr=0.5))
        sph.append({"center": [cx, 2, cz], "radius": 1.5, "material": 1 + k})
    T, pos = camera((0, 6, 12), pitch_deg=-20)
    lamps = [{"pos": [6, 12, 2], "intensity": 1.0, "distance": 30.0},
             {"pos": [-8, 9, -4], "intensity": 0.6, "distance": 25.0}]
    return SceneSpec(gv, gf, guv, np.full((gf.shape[0],), True), np.zeros((gf.shape[0],), np.int32), mats, sph, lamps,
                     T, pos, {"ambientColor": [0.04, 0.04, 0.05], "horizonColor": [0.5, 0.6, 0.8], "ambientFactor": 0.0})


def scene_c4(seed=4, cells=71, n_tex=8, tex_size=2048) -> SceneSpec:
    """Config 4: ~10k triangles in n_tex material groups, each with its own tex_size^2 texture
    (UV scale 4, wrapping), 1 lamp."""
    rng = np.random.default_rng(seed)
    gv, gf, guv = grid_mesh(cells, cells, -12, 12, -20, 4, lambda X, Z: 0.2 * np.sin(0.9 * X) * np.cos(0.7 * Z), uv_tiles=1.0)
    mats = [material(rng.uniform(0.3, 0.9, 3), kd=0.85, ks=0.3, hard=40,
                     texture=checker_texture(tex_size, rng, cells=8 + 4 * k), scale=(4.0, 4.0)) for k in range(n_tex)]
    cell = np.arange(gf.shape[0]) // 2
    fm = ((cell % cells) * n_tex // cells).astype(np.int32)
    T, pos = camera((0, 6, 12), pitch_deg=-20)
    lamps = [{"pos": [5, 11, 3], "intensity": 1.0, "distance": 30.0}]
    return SceneSpec(gv, gf, guv, np.full((gf.shape[0],), True), fm, mats, [], lamps, T, pos,
                     {"ambientColor": [0.05, 0.05, 0.05], "horizonColor": [0.4, 0.5, 0.7], "ambientFactor": 0.0})


def scene_c5(seed=5, cells=100, n_lamps=16, n_spheres=16) -> SceneSpec:
    """Config 5: 20k-triangle heightfield + 16 spheres, 16 lamps on a ring (shadow dominated)."""
    rng = np.random.default_rng(seed)
    gv, gf, _ = grid_mesh(cells, cells, -12, 12, -20, 4, lambda X, Z: 0.3 * np.sin(0.8 * X) * np.sin(0.6 * Z))
    mats = [material((0.75, 0.75, 0.7), kd=0.8, ks=0.2, hard=25)]
    sph = []
    for k in range(n_spheres):
        mats.append(material(rng.uniform(0.2, 0.9, 3), kd=0.8, ks=0.6, hard=60))
        sph.append({"center": [int(rng.integers(-9, 10)), int(rng.integers(1, 4)), int(rng.integers(-16, 1))],
                    "radius": float(np.float32(rng.uniform(0.6, 1.3))), "material": 1 + k})
    lamps = []
    for k in range(n_lamps):
        a = 2 * math.pi * k / n_lamps
        lamps.append({"pos": [round(10 * math.cos(a)), 10, round(-8 + 10 * math.sin(a))], "intensity": 0.25, "distance": 30.0})
    T, pos = camera((0, 6, 12), pitch_deg=-20)
    return SceneSpec(gv, gf, None, None, np.zeros((gf.shape[0],), np.int32), mats, sph, lamps, T, pos,
                     {"ambientColor": [0.03, 0.03, 0.03], "horizonColor": [0.35, 0.4, 0.5], "ambientFactor": 0.0})


def scene_mirror_box(seed=7) -> SceneSpec:
    """Small closed room with mirror walls and two spheres: deep reflection chains."""
    rng = np.random.default_rng(seed)
    bv, bf = box_mesh((-4, 0, -12), (4, 6, 2))
    bf = bf[:, ::-1].copy()                      # face inward
    mats = [material((0.8, 0.8, 0.85), kd=0.5, ks=0.4, hard=40, kr=0.6),
            material((0.9, 0.3, 0.2), kd=0.7, ks=0.9, hard=90, kr=0.4),
            material(rng.uniform(0.2, 0.9, 3), kd=0.7, ks=0.5, hard=30, kr=0.0)]
    sph = [{"center": [-1, 1, -6], "radius": 1.0, "material": 1}, {"center": [2, 2, -8], "radius": 1.25, "material": 2}]
    lamps = [{"pos": [0, 5, -3], "intensity": 1.0, "distance": 45.0}, {"pos": [-3, 4, -9], "intensity": 0.7, "distance": 35.0}]
    T, pos = camera((0, 3, 1), pitch_deg=-8, yaw_deg=4)
    return SceneSpec(bv, bf, None, None, np.zeros((bf.shape[0],), np.int32), mats, sph, lamps, T, pos,
                     {"ambientColor": [0.02, 0.02, 0.02], "horizonColor": [0.3, 0.3, 0.3], "ambientFactor": 0.0})


def scene_glass(seed=11) -> SceneSpec:
    """Mirror-box variant for the optional refraction mode: two glass spheres (one in front of the other), a glass
    slab of two triangles, a mirror floor; rays enter and leave the glass and meet total internal reflection."""
    rng = np.random.default_rng(seed)
    bv, bf = box_mesh((-5, 0, -14), (5, 7, 3))
    bf = bf[:, ::-1].copy()
    sv = np.array([[-2.5, 0.5, -5.5], [2.5, 0.5, -5.5], [2.5, 4.5, -6.5], [-2.5, 4.5, -6.5]], np.float32)     # tilted pane
    sf = np.array([[0, 1, 2], [0, 2, 3]], np.int64)
    mats = [material((0.8, 0.8, 0.85), kd=0.6, ks=0.3, hard=30, kr=0.3),
            material((0.9, 0.95, 1.0), kd=0.1, ks=0.8, hard=120, kr=0.2, kt=0.85, ior=1.5),
            material((0.6, 0.9, 0.7), kd=0.2, ks=0.6, hard=80, kt=0.7, ior=1.33),
            material(rng.uniform(0.2, 0.9, 3), kd=0.8, ks=0.4, hard=25)]
    V, F, UV, HAS, M = _merge([(bv, bf, None, 0), (sv, sf, None, 2)])
    sph = [{"center": [-1, 2, -8], "radius": 1.5, "material": 1}, {"center": [2, 1, -4], "radius": 1.0, "material": 1},
           {"center": [1, 2, -11], "radius": 1.75, "material": 3}]
    lamps = [{"pos": [0, 6, -2], "intensity": 1.0, "distance": 45.0}, {"pos": [-4, 5, -10], "intensity": 0.6, "distance": 35.0}]
    T, pos = camera((0, 3, 2), pitch_deg=-6, yaw_deg=3)
    return SceneSpec(V, F, None, None, M, mats, sph, lamps, T, pos,
                     {"ambientColor": [0.03, 0.03, 0.03], "horizonColor": [0.3, 0.35, 0.45], "ambientFactor": 0.0})


def scene_ties(seed=8) -> SceneSpec:
    """Adversarial ties: coplanar duplicate triangles (lowest index must win), a sphere tangent
    to a triangle plane, a face with null material, a face without UVs on a textured material,
    no world block, fractional sphere / lamp coordinates (truncated by the loader)."""
    rng = np.random.default_rng(seed)
    gv, gf, guv = grid_mesh(4, 4, -4, 4, -10, -2, None, uv_tiles=2.0)
    tex = checker_texture(64, rng, cells=8)
    mats = [material((0.5, 0.6, 0.7), kd=0.9, ks=0.4, hard=32, texture=tex, scale=(1.0, 1.0)),
            material((0.9, 0.9, 0.1), kd=0.8, ks=0.1, hard=8)]
    V, F, UV, HAS, M = _merge([(gv, gf, guv, 0), (gv, gf, guv, 1)])     # second copy coplanar, other material
    HAS[3] = False                                                     # textured material, no uv
    M[5] = -1                                                          # null material
    sph = [{"center": [0.7, 1.2, -5.9], "radius": 1.0, "material": 1},    # -> (0,1,-5): tangent to y=0 plane
           {"center": [2, 1, -7], "radius": 1.0, "material": None}]
    lamps = [{"pos": [1.9, 6.2, -2.5], "intensity": 1.0, "distance": 15.0}]
    T, pos = camera((0, 4, 4), pitch_deg=-25)
    return SceneSpec(V, F, UV, HAS, M, mats, sph, lamps, T, pos, None)


def scene_soup(seed=9, n_tri=1500, extent=10.0) -> SceneSpec:
    """Unordered triangle soup: sizes from 1e-3 to the whole scene, slivers, a few giant
    triangles, shuffled so that consecutive indices share nothing spatially.  Stresses any
    bound computed over index ranges."""
    rng = np.random.default_rng(seed)
    c = rng.uniform([-extent, 0.0, -2.0 * extent], [extent, 0.6 * extent, 0.0], size=(n_tri, 3))
    size = 10.0 ** rng.uniform(-3.0, 0.3, size=(n_tri, 1, 1))
    size[rng.integers(0, n_tri, 6)] = extent                              # giants
    off = rng.normal(size=(n_tri, 3, 3)) * size
    off[::7, 2] = off[::7, 1] * 1.01 + off[::7, 0] * 1e-2               # slivers
    V = (c[:, None, :] + off).reshape(-1, 3).astype(np.float32)
    F = np.arange(3 * n_tri, dtype=np.int64).reshape(n_tri, 3)
    F = F[rng.permutation(n_tri)]
    mats = [material(rng.uniform(0.2, 0.9, 3), kd=0.8, ks=0.4, hard=20, kr=0.3),
            material(rng.uniform(0.2, 0.9, 3), kd=0.6, ks=0.8, hard=60, kr=0.0)]
    fm = rng.integers(0, 2, n_tri).astype(np.int32)
    sph = [{"center": [int(x) for x in rng.integers(-6, 6, 3) + np.array([0, 4, -10])], "radius": float(rng.uniform(0.5, 2.0)),
            "material": int(i % 2)} for i in range(5)]
    lamps = [{"pos": [3, 9, 2], "intensity": 1.0, "distance": 40.0}, {"pos": [-6, 5, -14], "intensity": 0.6, "distance": 30.0}]
    T, pos = camera((0, 3, 9), pitch_deg=-6, yaw_deg=3)
    return SceneSpec(V, F, None, None, fm, mats, sph, lamps, T, pos,
                     {"ambientColor": [0.05, 0.05, 0.05], "horizonColor": [0.2, 0.3, 0.4], "ambientFactor": 0.0})


def scene_city(seed=12, cells=96, n_boxes=1500, yaw=False) -> SceneSpec:
    """Not a benchmark config: a ground grid with boxes of random size standing on it (walls, roofs and ground mix their
    orientations in every neighbourhood, as a modelled scene does and the terrains of configs 2-5 do not), 2 lamps.
    `yaw`: the boxes are turned by random angles, so the walls' normals fill the horizontal directions."""
    rng = np.random.default_rng(seed)
    gv, gf, _ = grid_mesh(cells, cells, -24, 24, -40, 8)
    mats = [material((0.6, 0.6, 0.55), kd=0.8, ks=0.2, hard=20, kr=0.1), material((0.7, 0.5, 0.3), kd=0.9, ks=0.4, hard=40)]
    meshes = [(gv, gf, None, 0)]
    for _ in range(n_boxes):
        x, z = rng.uniform(-23, 22), rng.uniform(-39, 7)
        w, d, hgt = rng.uniform(0.2, 1.0), rng.uniform(0.2, 1.0), rng.uniform(0.3, 3.0)
        bv, bf = box_mesh((x, 0, z), (x + w, hgt, z + d))
        if yaw:                                     # every box turned about its own vertical axis
            a = rng.uniform(0, np.pi)
            c, sn = np.float32(np.cos(a)), np.float32(np.sin(a))
            cx, cz = np.float32(x + 0.5 * w), np.float32(z + 0.5 * d)
            px, pz = bv[:, 0] - cx, bv[:, 2] - cz
            bv = np.stack([cx + c * px - sn * pz, bv[:, 1], cz + sn * px + c * pz], axis=-1).astype(np.float32)
        meshes.append((bv, bf, None, 1))
    V, F, UV, HAS, M = _merge(meshes)
    T, pos = camera((0, 9, 16), pitch_deg=-22)
    lamps = [{"pos": [10, 14, 6], "intensity": 1.0, "distance": 45.0}, {"pos": [-14, 10, -20], "intensity": 0.8, "distance": 45.0}]
    return SceneSpec(V, F, None, None, M, mats, [], lamps, T, pos,
                     {"ambientColor": [0.05, 0.05, 0.06], "horizonColor": [0.45, 0.5, 0.6], "ambientFactor": 0.0})


def small_variant(name: str) -> SceneSpec:
    """Reduced-size versions of the benchmark scenes for parity tests."""
    return {
        "c1": lambda: scene_c1(plane_cells=8),
        "c1_frac": lambda: scene_c1(plane_cells=8, fractional=True),
        "c2": lambda: scene_c2(cells=24, tex_size=64),
        "c4": lambda: scene_c4(cells=12, n_tex=4, tex_size=128),
        "c5": lambda: scene_c5(cells=12, n_lamps=16, n_spheres=6),
        "mirror": scene_mirror_box,
        "glass": scene_glass,
        "ties": scene_ties,
        "soup": scene_soup,
        "city": lambda: scene_city(cells=16, n_boxes=40),
        "city_yaw": lambda: scene_city(cells=16, n_boxes=40, yaw=True),
    }[name]()


SMALL_SCENES = ("c1", "c1_frac", "c2", "c4", "c5", "mirror", "ties", "city")
