"""Host model of the round-off bound of ``csrc/contains.hip`` (contains_binned_kernel).

Emulates, in numpy fp32 (fma = exact product and sum in fp64, rounded once to fp32 - inputs are fp32, so the fp64
product is exact and the double rounding of the sum is the only liberty), the kernel's per-triangle setup, its per-pair
test and its inflated projected box, on adversarial pairs: points placed ON the borders of the projected triangle (the
three edges, the vertices, their continuations) displaced by 0 .. a few ulp, slivers with Q up to 1e5, triangles far from
the origin, tiny triangles near the parallel threshold.  Claim checked: a pair that passes the fp32 test lies inside the
box (so skipping pairs outside the box never loses a hit).  Also reports how tight the margin is: the largest observed
excursion outside the UN-inflated box, as a fraction of the margin.

``python tools/contains_bound_sim.py [n_triangles]``; imported by tests/test_contains_bound_host.py.
"""
import sys

import numpy as np

F32 = np.float32
RAY = np.array([0.4395064455, 0.617598629942, 0.652231566745], dtype=F32)
AX = np.array([-0.814752659671, 0.579808678409, 0.0], dtype=F32)
AY = np.array([-0.378169522731, -0.531407403727, 0.758019777672], dtype=F32)
TOL = F32(0.0000001)
EPS = F32(5.9604645e-8)
QMAX = F32(1.0e4)


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F32)


def dot3(ax, ay, az, bx, by, bz):
    return fma(az, bz, fma(ay, by, (ax * bx).astype(F32)))


def tri_setup(A, B, C):
    """A, B, C [T,3] fp32 -> dict of the kernel's per-triangle constants (fp32, same operation order)."""
    e1 = (B - A).astype(F32)
    e2 = (C - A).astype(F32)
    rx, ry, rz = RAY
    px = (ry * e2[:, 2] - rz * e2[:, 1]).astype(F32)
    py = (rz * e2[:, 0] - rx * e2[:, 2]).astype(F32)
    pz = (rx * e2[:, 1] - ry * e2[:, 0]).astype(F32)
    det = ((e1[:, 0] * px + e1[:, 1] * py).astype(F32) + e1[:, 2] * pz).astype(F32)
    D = (det + F32(0.1) * TOL).astype(F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = (F32(1.0) / D).astype(F32)
    inv = np.where(np.abs(det) < TOL, F32(np.nan), inv)
    wx = (e1[:, 1] * rz - e1[:, 2] * ry).astype(F32)
    wy = (e1[:, 2] * rx - e1[:, 0] * rz).astype(F32)
    wz = (e1[:, 0] * ry - e1[:, 1] * rx).astype(F32)
    nx = (e1[:, 1] * e2[:, 2] - e1[:, 2] * e2[:, 1]).astype(F32)
    ny = (e1[:, 2] * e2[:, 0] - e1[:, 0] * e2[:, 2]).astype(F32)
    nz = (e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]).astype(F32)
    PU = np.stack([px * inv, py * inv, pz * inv], 1).astype(F32)
    PW = np.stack([wx * inv, wy * inv, wz * inv], 1).astype(F32)
    PN = np.stack([nx * inv, ny * inv, nz * inv], 1).astype(F32)
    return dict(A=A, e1=e1, e2=e2, det=det, D=D, PU=PU, PW=PW, PN=PN)


def ray_hit(o, s, need_t=True):
    """o [T,K,3] points per triangle -> bool [T,K] of the kernel's fp32 test (need_t=False: the u / v / u+v part only,
    which is all the box argument uses)."""
    A = s["A"][:, None]
    t = (o - A).astype(F32)
    tx, ty, tz = t[..., 0], t[..., 1], t[..., 2]
    def d(P):
        return dot3(tx, ty, tz, np.broadcast_to(P[:, None, 0], tx.shape), np.broadcast_to(P[:, None, 1], tx.shape),
                    np.broadcast_to(P[:, None, 2], tx.shape))
    with np.errstate(invalid="ignore"):
        u, v, tt = d(s["PU"]), d(s["PW"]), d(s["PN"])
        ok = (u > 0) & (u < 1) & (v > 0) & ((u + v).astype(F32) < 1)
        if need_t:
            ok &= tt >= TOL
    return ok


def proj(v):
    x = dot3(v[..., 0], v[..., 1], v[..., 2], *[np.broadcast_to(c, v[..., 0].shape) for c in AX])
    y = dot3(v[..., 0], v[..., 1], v[..., 2], *[np.broadcast_to(c, v[..., 0].shape) for c in AY])
    return x, y


def tri_box(s, tile_centre, tile_radius, cmax):
    """The kernel's inflated projected box per triangle: (boxed [T] bool, xlo, xhi, ylo, yhi, margin, raw box)."""
    e1, e2, A = s["e1"], s["e2"], s["A"]
    e3 = (e2 - e1).astype(F32)
    sq = lambda v: dot3(v[:, 0], v[:, 1], v[:, 2], v[:, 0], v[:, 1], v[:, 2])
    l1, l2, l3 = sq(e1), sq(e2), sq(e3)
    Lm = np.sqrt(np.maximum(l1, l2)).astype(F32)
    L3 = np.sqrt(np.maximum(np.maximum(l1, l2), l3)).astype(F32)
    absD = np.abs(s["D"])
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        Q = (Lm * L3 / absD).astype(F32)
        dv = (A - tile_centre[None]).astype(F32)
        tmax = (F32(1.001) * np.sqrt(sq(dv)).astype(F32) + tile_radius).astype(F32)
        call = (cmax + np.abs(A).sum(1).astype(F32) + L3).astype(F32)
        m = (F32(128.0) * EPS * F32(1.01) * Q * (tmax + L3) + F32(64.0) * EPS * call).astype(F32)
        ff = (F32(1.0) + F32(1.2e-8) / absD).astype(F32)
        gax, gay = proj(A)
        g1x, g1y = proj(e1)
        g2x, g2y = proj(e2)
        g1x, g1y, g2x, g2y = (ff * g1x).astype(F32), (ff * g1y).astype(F32), (ff * g2x).astype(F32), (ff * g2y).astype(F32)
        z = np.zeros_like(gax)
        rxlo = (gax + np.minimum(z, np.minimum(g1x, g2x))).astype(F32)
        rxhi = (gax + np.maximum(z, np.maximum(g1x, g2x))).astype(F32)
        rylo = (gay + np.minimum(z, np.minimum(g1y, g2y))).astype(F32)
        ryhi = (gay + np.maximum(z, np.maximum(g1y, g2y))).astype(F32)
        xlo, xhi, ylo, yhi = (rxlo - m).astype(F32), (rxhi + m).astype(F32), (rylo - m).astype(F32), (ryhi + m).astype(F32)
    fin = lambda a: np.abs(a) <= np.finfo(F32).max
    boxed = (Q <= QMAX) & fin(xlo) & fin(xhi) & fin(ylo) & fin(yhi)
    return boxed, xlo, xhi, ylo, yhi, m, (rxlo, rxhi, rylo, ryhi)


def adversarial_scene(rng, T, K):
    """T triangles x K points each, fp32.  Triangle families: well shaped, slivers (log-uniform aspect), nearly edge-on to
    the ray, tiny (det near tol), far from the origin.  Points: on the triangle's projected border lines (edge lines at
    barycentric 0 / f), displaced along the ray by a random amount and off the border by 0 .. +-16 ulp-ish steps."""
    scale = 10.0 ** rng.uniform(-3, 2.5, size=(T, 1))           # metres .. hundreds of mm
    centre = rng.normal(0, 1, size=(T, 3)) * scale * 10.0 ** rng.uniform(-1, 2, size=(T, 1))
    A = centre
    d1 = rng.normal(0, 1, size=(T, 3))
    d2 = rng.normal(0, 1, size=(T, 3))
    fam = rng.randint(0, 5, size=T)
    sliver = 10.0 ** rng.uniform(-6, 0, size=(T, 1))
    d2 = np.where((fam == 1)[:, None], d1 * rng.uniform(0.2, 2, size=(T, 1)) + d2 * sliver, d2)   # sliver
    ray = RAY.astype(np.float64)
    edge_on = 10.0 ** rng.uniform(-6, 0, size=(T, 1))
    d2 = np.where((fam == 2)[:, None], ray[None] * rng.uniform(0.2, 2, size=(T, 1)) + d2 * edge_on, d2)  # contains the ray
    tiny = np.where(fam == 3, 10.0 ** rng.uniform(-4.5, -2.5, size=T), 1.0)[:, None]
    B = A + d1 * scale * tiny
    C = A + d2 * scale * tiny
    A, B, C = A.astype(F32), B.astype(F32), C.astype(F32)
    # points on border lines in exact (fp64) barycentrics of the fp32 triangle
    A64, e1, e2 = A.astype(np.float64), B.astype(np.float64) - A.astype(np.float64), C.astype(np.float64) - A.astype(np.float64)
    kind = rng.randint(0, 6, size=(T, K))
    a = rng.uniform(-0.5, 1.5, size=(T, K))
    b = rng.uniform(-0.5, 1.5, size=(T, K))
    det = np.einsum("ti,ti->t", e1, np.cross(ray[None], e2))
    f = np.where(np.abs(det) > 0, (det + 1e-8) / np.where(det == 0, 1, det), 1.0)[:, None]
    a = np.where(kind == 0, 0.0, a)                       # on u = 0
    b = np.where(kind == 1, 0.0, b)                       # on v = 0
    b = np.where(kind == 2, f - a, b)                     # on u + v = f
    a = np.where(kind == 3, rng.choice([0.0, 1.0], size=(T, K)) * f, a)   # at / near vertices
    b = np.where(kind == 3, np.where(a == 0, rng.choice([0.0, 1.0], size=(T, K)) * f, 0.0), b)
    # kind 4: interior, kind 5: anywhere
    a = np.where(kind == 4, rng.uniform(0, 0.5, size=(T, K)) * f, a)
    b = np.where(kind == 4, rng.uniform(0, 0.5, size=(T, K)) * f, b)
    nudge = rng.choice([0, 0, 1, -1, 3, -3, 16, -16, 300, -300], size=(T, K, 2)) * 2.0 ** -24
    a = a + nudge[..., 0] * np.maximum(np.abs(a), 1e-3)
    b = b + nudge[..., 1] * np.maximum(np.abs(b), 1e-3)
    s_along = rng.normal(0, 1, size=(T, K, 1)) * scale[:, None] * 10.0 ** rng.uniform(-1, 1.5, size=(T, K, 1))
    s_along = -np.abs(s_along)  # in front of the triangle along the ray: the t test passes, the u / v tests decide
    o = A64[:, None] + a[..., None] * e1[:, None] + b[..., None] * e2[:, None] + s_along * ray[None, None]
    return A, B, C, o.astype(F32)


def check(seed=0, T=20000, K=64, need_t=False):
    rng = np.random.RandomState(seed)
    A, B, C, o = adversarial_scene(rng, T, K)
    s = tri_setup(A, B, C)
    hit = ray_hit(o, s, need_t=need_t)
    # tile = this triangle's K points (the kernel's tile bounds, computed as the kernel does)
    lo, hi = o.min(1), o.max(1)
    c = (F32(0.5) * (lo + hi)).astype(F32)
    r = np.maximum(hi - c, c - lo).astype(F32)
    rad = (F32(1.001) * np.sqrt((r * r).sum(1)).astype(F32)).astype(F32)
    cm = (F32(1.75) * np.maximum(np.abs(lo), np.abs(hi)).max(1)).astype(F32)
    # per-triangle tile centre: evaluate the box with each triangle's own tile
    e = dict(s)
    boxed = np.zeros(T, bool)
    xlo = np.zeros(T, F32); xhi = np.zeros(T, F32); ylo = np.zeros(T, F32); yhi = np.zeros(T, F32); m = np.zeros(T, F32)
    raw = [np.zeros(T, F32) for _ in range(4)]
    # tri_box takes one tile centre; vectorise by shifting nothing: call per chunk of identical semantics
    dv = (A - c).astype(F32)
    sq = lambda v: dot3(v[:, 0], v[:, 1], v[:, 2], v[:, 0], v[:, 1], v[:, 2])
    # re-implement tri_box's tile-dependent pieces with per-triangle tiles
    boxed, xlo, xhi, ylo, yhi, m, raw = _tri_box_per_tile(s, dv, rad, cm)
    gx, gy = proj(o)
    inside = (gx >= xlo[:, None]) & (gx <= xhi[:, None]) & (gy >= ylo[:, None]) & (gy <= yhi[:, None])
    live = ~np.isnan(s["PU"][:, 0])
    lost = hit & ~inside & boxed[:, None] & live[:, None]
    # tightness: excursion of hitting points outside the raw box, in units of the margin
    with np.errstate(invalid="ignore", divide="ignore"):
        exc = np.maximum.reduce([raw[0][:, None] - gx, gx - raw[1][:, None], raw[2][:, None] - gy, gy - raw[3][:, None]])
        frac = np.where(hit & boxed[:, None] & live[:, None], exc / m[:, None], -np.inf)
    return dict(pairs=int(T * K), hits=int(hit.sum()), boxed=int(boxed.sum()), live=int(live.sum()), lost=int(lost.sum()),
                worst_margin_fraction=float(np.max(frac)))


def _tri_box_per_tile(s, dv, rad, cm):
    e1, e2, A = s["e1"], s["e2"], s["A"]
    e3 = (e2 - e1).astype(F32)
    sq = lambda v: dot3(v[:, 0], v[:, 1], v[:, 2], v[:, 0], v[:, 1], v[:, 2])
    l1, l2, l3 = sq(e1), sq(e2), sq(e3)
    Lm = np.sqrt(np.maximum(l1, l2)).astype(F32)
    L3 = np.sqrt(np.maximum(np.maximum(l1, l2), l3)).astype(F32)
    absD = np.abs(s["D"])
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        Q = (Lm * L3 / absD).astype(F32)
        tmax = (F32(1.001) * np.sqrt(sq(dv)).astype(F32) + rad).astype(F32)
        call = (cm + np.abs(A).sum(1).astype(F32) + L3).astype(F32)
        m = (F32(128.0) * EPS * F32(1.01) * Q * (tmax + L3) + F32(64.0) * EPS * call).astype(F32)
        ff = (F32(1.0) + F32(1.2e-8) / absD).astype(F32)
        gax, gay = proj(A)
        g1x, g1y = proj(e1)
        g2x, g2y = proj(e2)
        g1x, g1y, g2x, g2y = (ff * g1x).astype(F32), (ff * g1y).astype(F32), (ff * g2x).astype(F32), (ff * g2y).astype(F32)
        z = np.zeros_like(gax)
        rxlo = (gax + np.minimum(z, np.minimum(g1x, g2x))).astype(F32)
        rxhi = (gax + np.maximum(z, np.maximum(g1x, g2x))).astype(F32)
        rylo = (gay + np.minimum(z, np.minimum(g1y, g2y))).astype(F32)
        ryhi = (gay + np.maximum(z, np.maximum(g1y, g2y))).astype(F32)
        xlo, xhi, ylo, yhi = (rxlo - m).astype(F32), (rxhi + m).astype(F32), (rylo - m).astype(F32), (ryhi + m).astype(F32)
    fin = lambda a: np.abs(a) <= np.finfo(F32).max
    boxed = (Q <= QMAX) & fin(xlo) & fin(xhi) & fin(ylo) & fin(yhi)
    return boxed, xlo, xhi, ylo, yhi, m, (rxlo, rxhi, rylo, ryhi)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    tot = dict(pairs=0, hits=0, boxed=0, live=0, lost=0, worst_margin_fraction=-np.inf)
    for seed in range(8):
        r = check(seed, T=n)
        for k in ("pairs", "hits", "boxed", "live", "lost"):
            tot[k] += r[k]
        tot["worst_margin_fraction"] = max(tot["worst_margin_fraction"], r["worst_margin_fraction"])
        print(seed, r, flush=True)
    print("total", tot)
