"""Known-answer tests of the CPU oracle (oracle/rt_oracle.c).

The reference ships no tests or vectors, so these answers are derived by hand from its
sources (file:line cited per case).  They pin the oracle's thresholds, edge rules, tie rules
and documented quirks; they are not outputs of the reference itself (parity unpinned).
"""
import numpy as np
import pytest

from raytracer_amd import _abi
from raytracer_amd.scene import FlatScene
from oracle import binding as orc

CPU, CL = _abi.HIPEXEC_MODE_CPU, _abi.HIPEXEC_MODE_CL
f32 = np.float32
P, E1, E2 = [0, 0, 0], [1, 0, 0], [0, 1, 0]          # unit right triangle in z = 0, u along x, v along y


def tri(mode, o, d):
    return orc.triangle_hit(mode, P, E1, E2, o, d)


def test_triangle_centre_hit():
    for mode in (CPU, CL):
        hit, t, u, v, pt = tri(mode, [0.25, 0.25, 1], [0, 0, -1])
        assert hit and t == 1 and u == f32(0.25) and v == f32(0.25)
        assert np.array_equal(pt, np.array([0.25, 0.25, 0], f32))


def test_triangle_unnormalised_direction_scales_t_only():
    hit, t, u, v, _ = tri(CPU, [0.25, 0.5, 2], [0, 0, -4])
    assert hit and t == f32(0.5) and u == f32(0.25) and v == f32(0.5)


@pytest.mark.parametrize("o,cpu_hit", [([0.0, 0.25, 1], True),      # u == 0
                                       ([0.25, 0.0, 1], True),      # v == 0
                                       ([0.5, 0.5, 1], True),       # u + v == 1
                                       ([0.0, 0.0, 1], True),       # vertex
                                       ([0.75, 0.5, 1], False),     # u + v > 1
                                       ([-0.01, 0.2, 1], False)])   # u < 0
def test_triangle_edges_inclusive_cpu_strict_cl(o, cpu_hit):
    # ray_tracer.cpp:323 rejects u<0 || v<0 || u+v>1 (edges inclusive); cl_kernels.c:147 needs u>0 && v>0 && u+v<1
    assert tri(CPU, o, [0, 0, -1])[0] == cpu_hit
    assert tri(CL, o, [0, 0, -1])[0] is False


def test_triangle_behind_and_parallel():
    for mode in (CPU, CL):
        assert not tri(mode, [0.25, 0.25, 1], [0, 0, 1])[0]        # t = -1
        assert not tri(mode, [0.25, 0.25, 1], [1, 0, 0])[0]        # det = 0


def test_triangle_t_threshold():
    eps_f = f32(0.0001)                                            # largest float below 1e-4
    up = np.nextafter(eps_f, f32(1))
    # ray_tracer.cpp:315  t < EPS (double compare): t = 0.0001f is rejected, the next float is accepted
    assert not tri(CPU, [0.25, 0.25, eps_f], [0, 0, -1])[0]
    assert tri(CPU, [0.25, 0.25, up], [0, 0, -1])[0]
    # cl_kernels.c:146  t > 0.001f
    assert not tri(CL, [0.25, 0.25, f32(0.001)], [0, 0, -1])[0]
    assert tri(CL, [0.25, 0.25, np.nextafter(f32(0.001), f32(1))], [0, 0, -1])[0]


def test_triangle_det_threshold():
    eps_f = f32(0.0001)
    up = np.nextafter(eps_f, f32(1))
    # ray_tracer.cpp:309  fabsf(det) < EPS; det = -d.z here
    assert not tri(CPU, [0.25, 0.25, 1], [0, 0, -eps_f])[0]
    hit, t, u, v, _ = tri(CPU, [0.25, 0.25, 1], [0, 0, -up])
    assert hit and t == f32(1) / up
    # cl_kernels.c:145  det > 0.001f || det < -0.001f
    assert not tri(CL, [0.25, 0.25, 1], [0, 0, -f32(0.001)])[0]
    assert tri(CL, [0.25, 0.25, 1], [0, 0, -np.nextafter(f32(0.001), f32(1))])[0]


def test_triangle_back_face_hits_too():
    hit, t, u, v, _ = tri(CPU, [0.25, 0.25, -2], [0, 0, 1])        # det < 0: no culling in the reference
    assert hit and t == 2 and u == f32(0.25) and v == f32(0.25)


def test_sphere_two_roots_and_inside():
    for mode in (CPU, CL):
        hit, t, pt, n = orc.sphere_hit(mode, [0, 0, 0], 1.0, [0, 0, 3], [0, 0, -1])
        assert hit and t == 2 and np.array_equal(pt, [0, 0, 1]) and np.array_equal(n, [0, 0, 1])
        hit, t, pt, n = orc.sphere_hit(mode, [0, 0, 0], 1.0, [0, 0, 0], [0, 0, -1])     # origin inside: far root
        assert hit and t == 1 and np.array_equal(pt, [0, 0, -1])
        assert not orc.sphere_hit(mode, [0, 0, 0], 1.0, [2, 0, 3], [0, 0, -1])[0]
        assert not orc.sphere_hit(mode, [0, 0, 0], 1.0, [0, 0, 3], [0, 0, 1])[0]        # both roots negative


def test_sphere_unnormalised_direction():
    hit, t, pt, n = orc.sphere_hit(CPU, [0, 0, 0], 1.0, [0, 0, 3], [0, 0, -2])          # a = 4
    assert hit and t == 1 and np.array_equal(pt, [0, 0, 1])


def test_sphere_normal_is_divided_by_radius():
    hit, t, pt, n = orc.sphere_hit(CPU, [0, 0, 0], 2.0, [0, 0, 5], [0, 0, -1])
    assert hit and t == 3 and np.array_equal(n, [0, 0, 1])


def test_sphere_tangent_branch_quirk():
    # D == 0.  ray_tracer.cpp:345-347 takes t = -b/2*a (multiplies by a): exact for a = 1 ...
    hit, t, _, _ = orc.sphere_hit(CPU, [0, 0, 0], 1.0, [1, 0, 3], [0, 0, -1])
    assert hit and t == 3
    # ... and 4x too large for |d| = 2 (true t is 1.5): b = -12, a = 4 -> (12/2)*4 = 24
    hit, t, _, _ = orc.sphere_hit(CPU, [0, 0, 0], 1.0, [1, 0, 3], [0, 0, -2])
    assert hit and t == 24
    # cl_kernels.c:214 needs D > -0.0 for that branch, so D == 0 goes through sqrt: t = 12/8
    hit, t, _, _ = orc.sphere_hit(CL, [0, 0, 0], 1.0, [1, 0, 3], [0, 0, -2])
    assert hit and t == f32(1.5)


def _scene(tris, spheres=(), mats=1, lamps=()):
    s = FlatScene()
    s.tri_p = [t[0] for t in tris]; s.tri_e1 = [t[1] for t in tris]; s.tri_e2 = [t[2] for t in tris]
    s.tri_norm = [[0, 0, 1]] * len(tris); s.tri_mat = [t[3] for t in tris]
    s.sph_center = [sp[0] for sp in spheres]; s.sph_radius = [sp[1] for sp in spheres]; s.sph_mat = [sp[2] for sp in spheres]
    s.mat_color = [[0.1 * (i + 1), 0.2, 0.3] for i in range(mats)]
    s.mat_kd = [0.5] * mats; s.mat_ks = [0.25] * mats; s.mat_hard = [2.0] * mats; s.mat_kr = [0.0] * mats
    s.mat_tex = [-1] * mats
    s.lamp_pos = [l[0] for l in lamps]; s.lamp_intensity = [l[1] for l in lamps]; s.lamp_distance = [l[2] for l in lamps]
    return s.normalised()


BIG = ([-10, -10, 0], [20, 0, 0], [0, 20, 0])


def test_closest_lowest_index_wins_ties():
    # ray_tracer.cpp:214 / opencl_executor.cpp:242: strict '<' keeps the first of equal t
    s = _scene([(*BIG, 0), (*BIG, 1), ([-10, -10, -1], [20, 0, 0], [0, 20, 0], 1)], mats=2)
    o = orc.Oracle(s)
    for mode in (CPU, CL):
        h = o.closest_hits([[-0.5, -0.25, 3, 0, 0, -1]], mode)[0]
        assert h["id"] == 0 and h["mat"] == 0 and h["t"] == 3
        assert np.array_equal(h["color"], s.mat_color[0])


def test_sphere_triangle_tie_rule_differs_by_mode():
    # sphere of radius 1 centred at z = -1 touches the plane z = 0 at the origin: both t = 3
    s = _scene([(*BIG, 0)], spheres=[([-1, -1, -1], 1.0, 1)], mats=2)
    o = orc.Oracle(s)
    assert o.closest_hits([[-1, -1, 3, 0, 0, -1]], CPU)[0]["id"] == 1      # ray_tracer.cpp:230: sphere wins
    assert o.closest_hits([[-1, -1, 3, 0, 0, -1]], CL)[0]["id"] == 0       # ray_tracer_cl.cpp:206: triangle wins


def test_null_material_is_default_black():
    s = _scene([(*BIG, -1)], mats=1)
    h = orc.Oracle(s).closest_hits([[-0.5, -0.5, 1, 0, 0, -1]], CPU)[0]
    assert h["id"] == 0 and h["mat"] == -1 and np.array_equal(h["color"], [0, 0, 0])


def test_any_hit_is_unbounded_and_counts_reference_order():
    # occluder beyond the "lamp" still shadows: ray_tracer.cpp:282-294 has no upper bound on t
    s = _scene([([5, 5, 50], [1, 0, 0], [0, 1, 0], 0), (*BIG, 0)], spheres=[([-0.5, -0.5, 30], 1.0, 0)])
    o = orc.Oracle(s)
    st = _abi.Stats()
    rays = [[-0.5, -0.5, 1, 0, 0, 1],    # misses tri 0, misses tri 1 (behind), hits the sphere at z=29
            [-0.5, -0.5, -1, 0, 0, 1],   # hits tri 1 (index 1) first in scan order
            [100, 100, 1, 0, 0, 1]]      # hits nothing
    hits = o.any_hits(rays, CPU, stats=st)
    assert list(hits) == [1, 1, 0]
    assert st.rays_shadow == 3
    assert st.tests_tri_any == 2 + 2 + 2 and st.tests_sph_any == 1 + 0 + 1
    pre = np.array([0, 0, 1], np.int8)   # OR-ed into the caller's array (ray_tracer_cl.cpp:255,260)
    assert list(o.any_hits(rays, CL, hits=pre)) == [1, 1, 1]


def test_texture_fetch_flip_wrap_and_float_to_unsigned():
    s = _scene([(*BIG, 0)], mats=1)
    tex = np.zeros((4, 4, 4), np.uint8)
    for r in range(4):
        for c in range(4):
            tex[r, c] = (10 * r, 10 * c, 7, 255)
    s.textures = [tex]; s.mat_tex = [0]
    s.tri_uv = [[0, 0, 20, 0, 0, 20]]            # uv = (20*(u), 20*(v)) texel units
    o = orc.Oracle(s.normalised())
    # tex_image.h:37: row (H-1-y), column x of the top-down image
    assert np.array_equal(o.tex_get(0, 1, 0) * 255, [30, 10, 7, 255])
    assert np.array_equal(o.tex_get(0, 2, 3) * 255, [0, 20, 7, 255])
    assert np.array_equal(o.tex_get(0, 5, 6) * 255, [10, 10, 7, 255])       # wrap: x=5%4=1, y=6%4=2 -> row 1
    # (unsigned) of a negative float as x86-64 evaluates it: -3.7 -> 0xFFFFFFFD -> % 4 = 1
    assert np.array_equal(o.tex_get(0, 0xFFFFFFFD, 0) * 255, [30, 10, 7, 255])


def test_trace_cpu_single_lamp_hand_computed():
    """One lit point, colour evaluated by hand in float32 in the reference's order (ray_tracer.cpp:166-187)."""
    s = _scene([(*BIG, 0)], mats=1, lamps=[([-1, -1, 4], 2.0, 3.0)])
    s.ambient = [0.01, 0.02, 0.03]; s.horizon = [0.5, 0.5, 0.5]
    o = orc.Oracle(s.normalised())
    d = np.array([0, 0, -1], f32)
    col = o.trace_rays([[-1, -1, 2, *d]], CPU, max_depth=0)[0]
    color = np.array(s.mat_color[0], f32)
    toLamp = np.array([0, 0, 4], f32)
    diff = f32(2.0) * f32(3.0) * f32(4.0) / f32(16.0)                       # I*dist*|N.L| / L.L
    exp = np.array(s.ambient, f32) + (color * f32(0.5)) * diff
    refl = toLamp - (f32(2.0) * np.array([0, 0, 1], f32)) * f32(4.0)        # (0,0,-4) -> normalised (0,0,-1)
    refl = refl * (f32(1.0) / np.sqrt(f32(16.0)))
    sd = f32(refl @ d)                                                      # 1
    phong = (f32(2.0) * f32(3.0) / f32(16.0)) * f32(np.power(sd, f32(2.0)))
    exp = exp + (color * f32(0.25)) * phong
    assert np.array_equal(col, exp.astype(f32))
    # a miss returns the horizon colour (ray_tracer.cpp:199)
    assert np.array_equal(o.trace_rays([[100, 100, 2, 0, 0, -1]], CPU)[0], [0.5, 0.5, 0.5])


def test_trace_facing_rules_differ_by_mode():
    # lamp exactly in the plane of the triangle: dotWithLamp == 0.
    s = _scene([(*BIG, 0)], mats=1, lamps=[([5, 0, 0], 1.0, 10.0)])
    o = orc.Oracle(s)
    st_cpu, st_cl = _abi.Stats(), _abi.Stats()
    o.trace_rays([[-1, -1, 2, 0, 0, -1]], CPU, stats=st_cpu)
    o.trace_rays([[-1, -1, 2, 0, 0, -1]], CL, stats=st_cl)
    # ray_tracer.cpp:176 skips only equal strict signs -> a shadow ray is cast; ray_tracer_cl.cpp:143
    # (dir<0 && lamp>=0) also casts one
    assert st_cpu.rays_shadow == 1 and st_cl.rays_shadow == 1
    # ray direction parallel to the surface normal plane (dotWithDir == 0): CPU casts, CL never lights
    s2 = _scene([(*BIG, 0)], spheres=[([0, 0, 1], 1.0, 0)], mats=1, lamps=[([0, 0, 9], 1.0, 10.0)])
    o2 = orc.Oracle(s2)
    st_cpu, st_cl = _abi.Stats(), _abi.Stats()
    ray = [[-5, 0, 2, 1, 0, 0]]       # grazes the sphere top at (0,0,2): normal (0,0,1), d.N == 0
    o2.trace_rays(ray, CPU, stats=st_cpu); o2.trace_rays(ray, CL, stats=st_cl)
    assert st_cpu.rays_shadow == 1 and st_cl.rays_shadow == 0


def test_reflection_recursion_depth_and_threshold():
    # two facing mirrors (kr = 0.5): depth-limited recursion, ray_tracer.cpp:191-195
    s = _scene([([-10, -10, 0], [20, 0, 0], [0, 20, 0], 0), ([-10, -10, 4], [20, 0, 0], [0, 20, 0], 0)], mats=1)
    s.mat_kr = [0.5]; s.ambient = [0.25, 0.25, 0.25]
    o = orc.Oracle(s.normalised())
    for depth in range(0, 5):
        st = _abi.Stats()
        c = o.trace_rays([[-1, -1, 2, 0, 0, -1]], CPU, max_depth=depth, stats=st)[0]
        assert st.rays_primary == 1 and st.rays_bounce == depth
        exp = f32(0)
        for _ in range(depth + 1):
            exp = f32(0.25) + exp * f32(0.5)
        assert np.array_equal(c, [exp] * 3)
    s.mat_kr = [0.0001]                      # kr > EPS is false for 0.0001f (config.h:20 is a double)
    st = _abi.Stats()
    orc.Oracle(s.normalised()).trace_rays([[-1, -1, 2, 0, 0, -1]], CPU, max_depth=3, stats=st)
    assert st.rays_bounce == 0


def test_render_region_matches_full_frame_and_threads():
    from raytracer_amd import scenegen
    fs = scenegen.small_variant("c2").to_flat()
    o = orc.Oracle(fs)
    full = o.render(40, 24, max_depth=2, n_threads=1)
    par = o.render(40, 24, max_depth=2, n_threads=4)
    assert np.array_equal(full["rgb"].view(np.int32), par["rgb"].view(np.int32))
    assert np.array_equal(full["ids"], par["ids"]) and full["stats"].rays == par["stats"].rays
    sub = o.render_region(40, 24, 8, 16, 24, 8, max_depth=2)         # like renderScene(..., x, y), ray_tracer.cpp:15-23
    assert np.array_equal(sub["rgb"].view(np.int32), full["rgb"][16:24, 8:32].view(np.int32))
    assert np.array_equal(sub["ids"], full["ids"][16:24, 8:32])
    # pixel = sum of 4 sub-samples / 4 then powf(c/2.2, 0.3) (ray_tracer.cpp:42-51)
    lin = full["linear"]
    tone = np.power((lin / f32(2.2)).astype(np.float64), 0.3).astype(f32)
    assert np.max(np.abs(tone - full["rgb"])) <= 2e-7


def test_bilinear_mode_is_the_nearest_lookup_at_texel_centres_and_blends_between():
    """HIPEXEC_FLAG_BILINEAR is not a reference mode; its definition: texel i is centred at i + 0.5,
    the four neighbours come from tex_image::get (same wrap, same vertical flip)."""
    from raytracer_amd.scene import FlatScene
    fs = FlatScene()
    tex = np.zeros((4, 4, 4), np.uint8)
    tex[..., 0] = np.arange(16).reshape(4, 4) * 16          # red ramp
    tex[..., 3] = 255
    fs.textures = [tex]
    o = orc.Oracle(fs.normalised())
    for x in range(4):
        for y in range(4):
            assert np.array_equal(o.tex_bilinear(0, x + 0.5, y + 0.5), o.tex_get(0, x, y))
    a, b = o.tex_get(0, 1, 2), o.tex_get(0, 2, 2)
    mid = o.tex_bilinear(0, 2.0, 2.5)                       # half way between texel 1 and texel 2 of row 2
    assert np.allclose(mid, 0.5 * (a + b), rtol=0, atol=1e-7)
    q = o.tex_bilinear(0, 1.75, 2.25)
    c = [o.tex_get(0, 1, 1), o.tex_get(0, 2, 1), o.tex_get(0, 1, 2), o.tex_get(0, 2, 2)]
    want = (c[0] * 0.75 + c[1] * 0.25) * 0.25 + (c[2] * 0.75 + c[3] * 0.25) * 0.75
    assert np.allclose(q, want, rtol=0, atol=1e-6)


# ---- optional refraction mode (HIPEXEC_FLAG_REFRACT; NOT reference behaviour: defined in include/hip_executor.h) ----
def _pane_scene(flip, ior, target):
    """Camera at the origin looking down -Z; a large pane through (0,0,-5) tilted 45 degrees about X, made of a material
    with refractionFactor 0.5; one more quad (`target`: (y0, y1, z) of a wall facing the camera, or (x.., "ceiling")).
    No lamps: every hit contributes the ambient colour, every miss the horizon colour."""
    from raytracer_amd.scenegen import SceneSpec, material
    s = np.float32(20.0)
    # pane: points p + a*(1,0,0) + b*(0,1,-1): normal ~ (0, 1, 1)/sqrt2 (towards the camera) for the first winding
    quad = np.array([[-s, -s, -5 + s], [s, -s, -5 + s], [s, s, -5 - s], [-s, s, -5 - s]], np.float32)
    faces = [[0, 1, 2], [0, 2, 3]]
    if flip:
        faces = [[0, 2, 1], [0, 3, 2]]
    verts = [quad]
    if target[0] == "wall":
        _, y0, y1, z = target
        verts.append(np.array([[-3, y0, z], [3, y0, z], [3, y1, z], [-3, y1, z]], np.float32))
    else:
        _, y = target
        verts.append(np.array([[-3, y, -8], [3, y, -8], [3, y, -2], [-3, y, -2]], np.float32))
    faces += [[4, 5, 6], [4, 6, 7]]
    mats = [material((0, 0, 0), kd=0, ks=0, hard=1, kt=0.5, ior=ior), material((0, 0, 0), kd=0, ks=0, hard=1)]
    return SceneSpec(np.concatenate(verts), np.asarray(faces), None, None, np.array([0, 0, 1, 1], np.int32), mats, [], [],
                     np.eye(3, dtype=np.float32), np.zeros(3, np.float32),
                     {"ambientColor": [0.1, 0.2, 0.3], "horizonColor": [0.5, 0.6, 0.8], "ambientFactor": 0.0}).to_flat()


def _centre(fs, refract, depth=1):
    from oracle.binding import Oracle
    r = Oracle(fs).render_region(64, 64, 32, 32, 1, 1, max_depth=depth, tone=0, refract=refract)
    return r["linear"][0, 0], r["ids"][0, 0]


def test_refraction_bends_towards_the_normal_on_entering():
    """45 degrees onto n = 1.5: sin(t) = sin(45)/1.5, the ray leaves the straight line by 16.87 degrees: from (0,0,-5) it
    reaches z = -15 at y = -10*tan(16.87 deg) = -3.03.  A wall covering y in [-6, -1.5] there is hit by the bent ray only."""
    a, hz = np.float32([0.1, 0.2, 0.3]), np.float32([0.5, 0.6, 0.8])
    bent = _pane_scene(False, 1.5, ("wall", -6.0, -1.5, -15.0))
    lin, ids = _centre(bent, True)
    assert np.isin(ids, (0, 1)).all()                                           # one of the pane's two triangles
    assert np.allclose(lin, a + np.float32(0.5) * a, rtol=1e-6)                 # pane (ambient) + 0.5 * wall (ambient)
    lin, _ = _centre(_pane_scene(False, 1.0, ("wall", -6.0, -1.5, -15.0)), True)
    assert np.allclose(lin, a + np.float32(0.5) * hz, rtol=1e-6)                # index 1: straight on, misses the wall
    lin, _ = _centre(bent, False)
    assert np.allclose(lin, a, rtol=1e-6)                                       # mode off: reflectionFactor is 0, the path ends
    lin, _ = _centre(_pane_scene(False, 0.0, ("wall", -6.0, -1.5, -15.0)), True)
    assert np.allclose(lin, a + np.float32(0.5) * hz, rtol=1e-6)                # index <= 0 counts as 1
    # the offset itself: a wall that ends 0.2 above / 0.2 below y = -3.03 separates hit from miss (the four sub-samples spread by 0.06)
    y = -10.0 * np.tan(np.deg2rad(45.0) - np.arcsin(np.sin(np.deg2rad(45.0)) / 1.5))
    assert abs(y + 3.03) < 0.01
    lin, _ = _centre(_pane_scene(False, 1.5, ("wall", -6.0, y + 0.2, -15.0)), True)
    assert np.allclose(lin, a + np.float32(0.5) * a, rtol=1e-6)
    lin, _ = _centre(_pane_scene(False, 1.5, ("wall", -6.0, y - 0.2, -15.0)), True)
    assert np.allclose(lin, a + np.float32(0.5) * hz, rtol=1e-6)


def test_total_internal_reflection_takes_the_mirrored_ray():
    """Leaving n = 1.5 at 45 degrees (critical angle 41.8): k = 1 - 2.25 * 0.5 < 0, the secondary ray is the mirrored one,
    (0,0,-1) - 2N(d.N) = (0,1,0) straight up from (0,0,-5); a ceiling patch above is hit.  At n = 1.3 (critical angle
    50.3) the ray leaves instead, bent away from the normal, and misses the patch."""
    a, hz = np.float32([0.1, 0.2, 0.3]), np.float32([0.5, 0.6, 0.8])
    lin, _ = _centre(_pane_scene(True, 1.5, ("ceiling", 8.0)), True)
    assert np.allclose(lin, a + np.float32(0.5) * a, rtol=1e-6)
    lin, _ = _centre(_pane_scene(True, 1.3, ("ceiling", 8.0)), True)
    assert np.allclose(lin, a + np.float32(0.5) * hz, rtol=1e-6)


# ---- optional ambient occlusion (HIPEXEC_FLAG_AO: ENABLE_AO of ray_tracer.cpp:167-169,406-435 with a counter-based generator) ----
def _ao_scene(closed):
    from raytracer_amd.scenegen import SceneSpec, material, box_mesh
    if closed:
        v, f = box_mesh((-3, -3, -12), (3, 3, 2))
        f = f[:, ::-1].copy()                       # a closed room around the camera: every occlusion ray hits a wall
    else:
        v = np.array([[-50, -2, 50], [50, -2, 50], [50, -2, -50], [-50, -2, -50]], np.float32)      # a floor below the camera
        f = np.array([[0, 1, 2], [0, 2, 3]], np.int64)
    mats = [material((0.5, 0.25, 1.0), kd=0, ks=0, hard=1)]
    from raytracer_amd.scenegen import camera
    T, pos = camera((0, 0, 0), pitch_deg=-30)
    return SceneSpec(v, f, None, None, np.zeros(f.shape[0], np.int32), mats, [], [], T, pos,
                     {"ambientColor": [0.1, 0.2, 0.3], "horizonColor": [0.5, 0.6, 0.8], "ambientFactor": 0.75}).to_flat()


def test_ambient_occlusion_open_floor_and_closed_room():
    """No lamps.  Above an open floor every occlusion ray is turned into the hemisphere of the normal that faces the viewer
    and escapes: ao = 30 * ambientFactor / 30, the pixel is ambient + colour * ambientFactor.  Inside a closed room every
    occlusion ray is blocked: ao = 0, the pixel is the ambient colour.  Without the flag: ambient in both."""
    from oracle.binding import Oracle
    a, col, f = np.float32([0.1, 0.2, 0.3]), np.float32([0.5, 0.25, 1.0]), np.float32(0.75)
    for closed, want in ((False, a + col * f), (True, a)):
        fs = _ao_scene(closed)
        assert abs(fs.ambient_factor - 0.75) < 1e-7
        r = Oracle(fs).render_region(32, 32, 16, 4, 1, 1, max_depth=0, tone=0, ao=True)
        assert (r["ids"] >= 0).all()
        assert np.allclose(r["linear"][0, 0], want, rtol=2e-6), (closed, r["linear"][0, 0])
        assert r["stats"].rays_shadow == 4 * 30                       # 30 occlusion rays per sub-sample, counted as shadow rays
        off = Oracle(fs).render_region(32, 32, 16, 4, 1, 1, max_depth=0, tone=0)
        assert np.allclose(off["linear"][0, 0], a, rtol=1e-6) and off["stats"].rays_shadow == 0
