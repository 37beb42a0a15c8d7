"""Independent numpy restatement of the reference tracer (CPU mode, ray_tracer.cpp), written
separately from oracle/rt_oracle.c as a cross-check of the oracle.  float32 elementwise numpy
operations round once per operation and never fuse, so the arithmetic is the reference's.
Vectorised over primitives per ray; meant for a few thousand rays."""
import numpy as np

f32 = np.float32
EPS = 0.0001          # double, config.h:20


def dot(a, b):
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def cross(x, y):
    return np.stack([x[..., 1] * y[..., 2] - y[..., 1] * x[..., 2],
                     x[..., 2] * y[..., 0] - y[..., 2] * x[..., 0],
                     x[..., 0] * y[..., 1] - y[..., 0] * x[..., 1]], axis=-1)


def normalize(v):
    return v * (f32(1.0) / np.sqrt(dot(v, v)))


class NumpyTracer:
    def __init__(self, fs):
        self.s = fs

    # ray_tracer.cpp:300-329, all triangles at once -> (hit mask, t, u, v)
    def tri_hits(self, o, d):
        s = self.s
        with np.errstate(all="ignore"):
            q = o[None, :] - s.tri_p
            det = dot(d[None, :], cross(s.tri_e2, s.tri_e1))
            ok = ~(np.abs(det).astype(np.float64) < EPS)
            dett = dot(q, cross(s.tri_e1, s.tri_e2))
            t = dett / det
            ok &= ~(t.astype(np.float64) < EPS)
            detu = dot(d[None, :], cross(s.tri_e2, q))
            detv = dot(d[None, :], cross(q, s.tri_e1))
            u = detu / det
            v = detv / det
            ok &= ~((u < 0) | (v < 0) | ((v + u) > 1))
        return ok, t, u, v

    # ray_tracer.cpp:332-373
    def sph_hits(self, o, d):
        s = self.s
        n = s.sph_radius.shape[0]
        ok = np.zeros(n, bool); tt = np.zeros(n, f32)
        for i in range(n):
            c, r = s.sph_center[i], s.sph_radius[i]
            v = o - c
            a = dot(d, d); b = f32(2) * dot(v, d); cc = dot(v, v) - r * r
            D = b * b - f32(4) * a * cc
            if D < 0:
                continue
            if float(np.abs(D)) < EPS:
                t = -b / f32(2) * a
                if float(t) > EPS:
                    ok[i] = True; tt[i] = t
                continue
            sq = np.sqrt(D)
            t1 = (-b + sq) / (f32(2) * a); t2 = (-b - sq) / (f32(2) * a)
            if float(t1) > EPS and float(t2) > EPS:
                ok[i] = True; tt[i] = t1 if t1 < t2 else t2
            elif float(t1) > EPS:
                ok[i] = True; tt[i] = t1
            elif float(t2) > EPS:
                ok[i] = True; tt[i] = t2
        return ok, tt

    def material(self, mi):
        s = self.s
        if mi < 0:
            return np.zeros(3, f32), f32(0), f32(0), f32(0), f32(0), -1
        return s.mat_color[mi], s.mat_kd[mi], s.mat_ks[mi], s.mat_hard[mi], s.mat_kr[mi], int(s.mat_tex[mi])

    def tex_get(self, tex, x, y):                       # tex_image.h:34-44 for square textures
        img = self.s.textures[tex]
        H, W = img.shape[:2]
        rx = x % W if x >= W else x
        ry = y % H if y >= W else y
        return img[H - 1 - ry, rx].astype(f32) / f32(255.0)

    # ray_tracer.cpp:204-273 -> dict or None
    def closest(self, o, d):
        s = self.s
        best_t = None
        if s.n_tri:
            ok, t, u, v = self.tri_hits(o, d)
            ok = ok & (t > 0)
            if ok.any():
                tm = np.where(ok, t, np.inf)
                i = int(np.argmin(tm))                  # first index of the minimum: strict '<' scan
                best_t = ("tri", i, t[i], u[i], v[i])
        best_s = None
        if s.n_sph:
            oks, ts = self.sph_hits(o, d)
            oks = oks & (ts > 0)
            if oks.any():
                tm = np.where(oks, ts, np.inf)
                j = int(np.argmin(tm))
                best_s = ("sph", j, ts[j])
        if best_t is not None and (best_s is None or best_t[2] < best_s[2]):
            _, i, t, u, v = best_t
            color, kd, ks, hard, kr, tex = self.material(int(s.tri_mat[i]))
            if tex >= 0:
                uv = s.tri_uv[i]
                ux = uv[0] + uv[2] * u + uv[4] * v
                uy = uv[1] + uv[3] * u + uv[5] * v
                rgba = self.tex_get(tex, int(np.int64(np.trunc(ux))) & 0xFFFFFFFF, int(np.int64(np.trunc(uy))) & 0xFFFFFFFF)
                color = rgba[3] * rgba[:3] + (f32(1) - rgba[3]) * color
            return dict(id=i, t=t, point=o + t * d, norm=s.tri_norm[i], color=color, kd=kd, ks=ks, hard=hard, kr=kr)
        if best_s is not None:
            _, j, t = best_s
            color, kd, ks, hard, kr, _ = self.material(int(s.sph_mat[j]))
            pt = o + t * d
            return dict(id=s.n_tri + j, t=t, point=pt, norm=(pt - s.sph_center[j]) / s.sph_radius[j], color=color,
                        kd=kd, ks=ks, hard=hard, kr=kr)
        return None

    # ray_tracer.cpp:276-297
    def any_hit(self, o, d):
        if self.s.n_tri and self.tri_hits(o, d)[0].any():
            return True
        return bool(self.s.n_sph and self.sph_hits(o, d)[0].any())

    # ray_tracer.cpp:156-201
    def trace(self, o, d, depth, max_depth):
        s = self.s
        h = self.closest(o, d)
        if h is None:
            return s.horizon.copy(), -1
        ret = s.ambient.copy()
        N = h["norm"]
        for l in range(s.n_lamp):
            L = s.lamp_pos[l] - h["point"]
            a = dot(L, N); b = dot(d, N)
            if (b < 0 and a < 0) or (b > 0 and a > 0):
                continue
            if self.any_hit(h["point"], L):
                continue
            diff = s.lamp_intensity[l] * s.lamp_distance[l] * np.abs(dot(N, L)) / dot(L, L)
            ret = ret + (h["color"] * h["kd"]) * diff
            R = normalize(L - (f32(2.0) * N) * dot(L, N))
            sd = dot(R, d)
            phong = f32(0)
            if sd > 0:
                phong = (h["hard"] * s.lamp_distance[l] / dot(L, L)) * f32(np.power(np.float64(sd), np.float64(h["hard"])))
            ret = ret + (h["color"] * h["ks"]) * phong
        if max_depth > 0 and float(h["kr"]) > EPS and depth < max_depth:
            rd = d - (f32(2.0) * N) * dot(d, N)
            sub, _ = self.trace(h["point"], rd, depth + 1, max_depth)
            ret = ret + sub * h["kr"]
        return ret.astype(f32), h["id"]

    # ray_tracer.cpp:15-54 -> (linear [H][W][3], ids [H][W][4])
    def render(self, W, H, max_depth):
        s = self.s
        camH = f32(0.5); camW = f32(W) * (camH / f32(H))
        dh = camH / f32(H); dw = camW / f32(W)
        M = s.cam_mat
        vm = lambda v: np.array([(M[i, 0] * v[0] + M[i, 1] * v[1]) + M[i, 2] * v[2] for i in range(3)], f32)
        lin = np.zeros((H, W, 3), f32); ids = np.zeros((H, W, 4), np.int32)
        wdx = vm(np.array([dw / f32(2), 0, 0], f32)); wdy = vm(np.array([0, dh / f32(2), 0], f32))
        for i in range(H):
            for j in range(W):
                rx = -(camW / f32(2)) + f32(j) * dw
                ry = -(camH / f32(2)) + f32(i) * dh
                wd = vm(np.array([rx, ry, -1], f32))
                dirs = [wd, wd + wdx, wd + wdy, (wd + wdx) + wdy]
                c = None
                for r, dd in enumerate(dirs):
                    col, hid = self.trace(s.cam_pos, normalize(dd), 0, max_depth)
                    c = col if c is None else (c + col)
                    ids[i, j, r] = hid
                lin[i, j] = c / f32(4.0)
        return lin, ids
