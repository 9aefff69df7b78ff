"""Independent numpy-float32 transcription of the reference's sampling code, straight from the Ada text (NOT from oracle/ or csrc/):
  vector_math.adb:64-82 (normalize, length, reflect), :175-326 (GetPerpendicular, MapSampleToCosineDist[Fixed]),
  generic_vector_math.adb:19-35,110-122 (min, max, dot, cross), lights.adb:42-255, materials.adb:18-99,197-410.
Every operation is a numpy float32 scalar operation in the source's order, so each intermediate is rounded to binary32 exactly as
`Float` arithmetic is.  sin / cos / "**" are taken as correctly rounded (float64 libm rounded once to float32) -- the contract of
ART-M1, checked against mpmath in tests/test_oracle_kat.py.  TEST INFRASTRUCTURE ONLY."""
import numpy as np

f = np.float32
M_PI = f(np.pi)            # vector_math.ads:19
INV_PI = f(1.0 / np.pi)    # vector_math.ads:20
EPS_DIV = f(1.0e-20)
EPS_COS = f(1.0e-6)
INFINITY = np.finfo(np.float32).max


def V(x, y, z):
    return (f(x), f(y), f(z))


def amin(a, b):
    return a if a < b else b


def amax(a, b):
    return a if a >= b else b


def clamp(x, a, b):           # generic_vector_math.adb:60-63
    return amin(amax(x, a), b)


def add(a, b):
    return (a[0] + b[0], a[1] + b[1], a[2] + b[2])


def sub(a, b):
    return (a[0] - b[0], a[1] - b[1], a[2] - b[2])


def scale(k, a):              # "*"(k, vector3) and "*"(vector3, k): (k*a.x, k*a.y, k*a.z)
    return (k * a[0], k * a[1], k * a[2])


def dot(a, b):
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def cross(a, b):
    return (a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1])


def sqrt(x):
    return f(np.sqrt(x))


def sin(x):
    return f(np.sin(np.float64(x)))


def cos(x):
    return f(np.cos(np.float64(x)))


def power(x, y):              # Ada "**" on Float: the RM special cases, else correctly rounded
    if y == 0:
        return f(1)
    if x == 0:
        return f(0)
    if x == 1:
        return f(1)
    if y == 1:
        return x
    if y == 2:
        return x * x
    if y == f(0.5):
        return sqrt(x)
    return f(np.power(np.float64(x), np.float64(y)))


def normalize(a):
    l_inv = f(1) / sqrt(dot(a, a))
    return (l_inv * a[0], l_inv * a[1], l_inv * a[2])


def length(a):
    return sqrt(dot(a, a))


def reflect(d, n):            # normalize((normal * dot(dir, normal) * (-2.0)) + dir)
    return normalize(add(scale(f(-2), scale(dot(d, n), n)), d))


def get_perpendicular(a):
    xp, yp, zp = abs(a[0]), abs(a[1]), abs(a[2])
    e = f(1.0e-5)
    if xp <= yp + e and xp <= zp + e:
        least = V(1, 0, 0)
    elif yp < xp + e and yp <= zp + e:
        least = V(0, 1, 0)
    else:
        least = V(0, 0, 1)
    return normalize(cross(a, least))


def _frame_tail(deviation, direction, normal):
    ny = direction
    nx = get_perpendicular(ny)
    nz = normalize(cross(nx, ny))
    ny, nz = nz, ny
    res = add(add(scale(deviation[0], nx), scale(deviation[1], ny)), scale(deviation[2], nz))
    inv_sign = f(1) if dot(direction, normal) >= 0 else f(-1)
    if inv_sign * dot(res, normal) < 0:
        nx = normalize(cross(normal, direction))
        nz = normalize(cross(nx, ny))
        if dot(nz, res) < 0:
            nz = scale(f(-1), nz)
        res = reflect(scale(f(-1), res), nz)
        if dot(res, normal) < 0:
            res = direction
    return res


def map_sample_to_cosine_dist(r1, r2, direction, normal, pw):
    sin_phi = sin(f(2) * r1 * M_PI)
    cos_phi = cos(f(2) * r1 * M_PI)
    cos_theta = power(f(1) - r2, f(1) / (pw + f(1)))
    sin_theta = sqrt(f(1) - cos_theta * cos_theta)
    return _frame_tail((sin_theta * cos_phi, sin_theta * sin_phi, cos_theta), direction, normal)


def map_sample_to_cosine_dist_fixed(r1, r2, direction, normal, pw):
    h = sqrt(f(1) - power(r1, f(2) / (pw + f(1))))
    dev = (h * cos(f(2) * M_PI * r2), h * sin(f(2) * M_PI * r2), power(r1, f(1) / (pw + f(1))))
    return _frame_tail(dev, direction, normal)


# ------------------------------------------------------------------------------------------------ lights.adb
def pdf_a_to_w(pdf_a, dist, cos_there):
    return pdf_a * dist * dist / amax(cos_there, EPS_DIV)


def area_light_sample(l, r1, r2, p):
    pos = (l["boxMin"][0] + r1 * (l["boxMax"][0] - l["boxMin"][0]), l["boxMin"][1], l["boxMin"][2] + r2 * (l["boxMax"][2] - l["boxMin"][2]))
    ray_dir = sub(pos, p)
    d = length(ray_dir)
    ray_dir = scale(f(1) / d, ray_dir)
    cos_theta = amax(dot(ray_dir, scale(f(-1), l["normal"])), f(0))
    return dict(pos=pos, dir=l["normal"], pdf=pdf_a_to_w(f(1) / l["surfaceArea"], d, cos_theta), intensity=l["intensity"])


def area_light_eval_pdf(l, p, ray_dir, hit_dist):
    return pdf_a_to_w(f(1) / l["surfaceArea"], hit_dist, amax(dot(ray_dir, scale(f(-1), l["normal"])), f(0)))


def distance_squared(a, b):
    d = sub(b, a)
    return dot(d, d)


def sphere_light_eval_pdf(l, p):
    if distance_squared(p, l["center"]) - l["radius"] * l["radius"] < f(1.0e-4):
        return f(1) / l["surfaceArea"]
    s2 = l["radius"] * l["radius"] / distance_squared(p, l["center"])
    cmax = sqrt(amax(f(0), f(1) - s2))
    return f(1) / (f(2) * M_PI * (f(1) - cmax))


def sphere_light_sample(l, u1, u2, p):
    c, rad = l["center"], l["radius"]
    res = dict(pos=V(0, 0, 0), dir=V(0, 0, 0), intensity=l["intensity"], pdf=f(1))
    if distance_squared(p, c) - rad * rad < f(1.0e-4):
        z = f(1) - f(2) * u1
        r = sqrt(amax(f(0), f(1) - z * z))
        phi = f(2) * M_PI * u2
        res["pos"] = add(c, scale(rad, (r * cos(phi), r * sin(phi), z)))
        res["dir"] = normalize(sub(res["pos"], c))
        return res
    wc = normalize(sub(c, p))
    if abs(wc[0]) > abs(wc[1]):
        inv_len = f(1) / sqrt(wc[0] * wc[0] + wc[2] * wc[2])
        wx = (-wc[2] * inv_len, f(0), wc[0] * inv_len)
    else:
        inv_len = f(1) / sqrt(wc[1] * wc[1] + wc[2] * wc[2])
        wx = (f(0), wc[2] * inv_len, -wc[1] * inv_len)
    wy = cross(wc, wx)
    s2 = rad * rad / distance_squared(p, c)
    cmax = sqrt(amax(f(0), f(1) - s2))
    costheta = (f(1) - u1) * cmax + u1 * f(1)                       # lerp(u1, costhetamax, 1.0)
    sintheta = sqrt(f(1) - costheta * costheta)
    phi = u2 * f(2) * M_PI
    rdir = add(add(scale(cos(phi) * sintheta, wx), scale(sin(phi) * sintheta, wy)), scale(costheta, wc))
    rpos = add(p, scale(f(1.0e-3), rdir))
    k = sub(rpos, c)
    b = dot(k, rdir)
    cc = dot(k, k) - rad * rad
    d = b * b - cc
    if d >= 0:
        sq = sqrt(d)
        hx = amin(-b - sq, -b + sq)
    else:
        hx = -INFINITY
    thit = dot(sub(c, p), normalize(rdir)) if hx < 0 else hx
    res["pos"] = add(rpos, scale(thit, rdir))
    res["dir"] = normalize(sub(res["pos"], c))
    res["pdf"] = sphere_light_eval_pdf(l, p)
    return res


# ------------------------------------------------------------------------------------------------ materials.adb
def fresnel(cos1, eta_ext, eta_int):
    if cos1 < 0:
        eta_ext, eta_int = eta_int, eta_ext
    sin2 = (eta_ext / eta_int) * sqrt(amax(f(0), f(1) - cos1 * cos1))
    if sin2 > 1:
        return f(1)
    cos2 = sqrt(amax(f(0), f(1) - sin2 * sin2))
    c1 = abs(cos1)
    e, i = eta_int, eta_ext                                           # fresnelDielectric(|cos1|, cos2, etaExt => etaInt, etaInt => etaExt)
    rs = (e * c1 - i * cos2) / (e * c1 + i * cos2)
    rp = (i * c1 - e * cos2) / (i * c1 + e * cos2)
    return (rs * rs + rp * rp) / f(2)


def mirror_sample(refl, ray_dir, n):
    nd = reflect(ray_dir, n)
    cdiv = f(1) / amax(dot(nd, n), EPS_DIV)
    return dict(color=scale(cdiv, refl), dir=nd, pdf=f(1), specular=True)


def lambert_sample(kd, r1, r2, ray_dir, n):
    nd = map_sample_to_cosine_dist(r1, r2, n, n, f(1))
    ct = dot(nd, n)
    color = scale(INV_PI, kd)
    if ct < EPS_COS:
        color = V(0, 0, 0)
    return dict(color=color, dir=nd, pdf=abs(ct) * INV_PI, specular=False)


def glass_sample(refl0, trans0, ior, ksi, ray_dir, n):
    fr = fresnel(dot(ray_dir, n), ior, f(1))
    refl = scale(fr, refl0)
    trans = scale(f(1) - fr, trans0)
    ksitrans = length(trans) / (length(refl) + length(trans))
    ksirefl = length(refl) / (length(refl) + length(trans))
    if ksi > ksitrans:
        nd = reflect(ray_dir, n)
        bxdf = scale(f(1) / ksirefl, refl)
    else:
        bxdf = scale(f(1) / ksitrans, trans)
        ci = dot(scale(f(-1), ray_dir), n)
        eta = ior
        if ci < 0:
            eta = f(1) / eta
        tir = (f(1) - (f(1) - ci * ci) / (eta * eta)) < 0
        if not tir:
            nn = n
            wo = scale(f(-1), ray_dir)
            if ci < 0:
                ci = -ci
                nn = scale(f(-1), nn)
            c2 = sqrt(f(1) - (f(1) - ci * ci) / (eta * eta))
            nd = normalize(sub(scale(f(1) / eta, scale(f(-1), wo)), scale(c2 - ci / eta, nn)))
        else:
            nd = reflect(ray_dir, n)
    cdiv = f(1) / amax(abs(dot(nd, n)), EPS_DIV)
    return dict(color=scale(cdiv, bxdf), dir=nd, pdf=f(1), specular=True)


PHONG_CLAMP = f(np.float64(np.pi) * 0.499995)      # static expression M_PI*0.499995: folded exactly, then rounded


def ada_pow(x, y):                                  # Vector_Math.pow, vector_math.adb:24-47
    if y == 0:
        return f(1)
    if x == 0:
        return f(0)
    if x == 1:
        return f(1)
    if y == 1:
        return x
    return power(x, y)


def phong_sample(refl, pw, r1, r2, ray_dir, n):
    r = reflect(ray_dir, n)
    nd = map_sample_to_cosine_dist_fixed(r1, r2, r, n, pw)
    ct = clamp(dot(nd, r), f(0), PHONG_CLAMP)
    lobe = ada_pow(ct, pw)
    color = scale(lobe, scale(INV_PI, scale(f(0.5), scale(pw + f(2), refl))))
    pdf = lobe * (pw + f(1)) * (f(0.5) * INV_PI)
    cg = dot(nd, n)
    cdiv = f(1) / amax(abs(cg), EPS_DIV)
    if cg < EPS_COS:
        color = V(0, 0, 0)
    return dict(color=scale(cdiv, color), dir=nd, pdf=pdf, specular=False)


def phong_eval(refl, pw, l, v, n):
    r = reflect(scale(f(-1), v), n)
    ct = clamp(dot(l, r), f(0), PHONG_CLAMP)
    lobe = ada_pow(ct, pw)
    cdiv = f(1) / amax(dot(l, n), EPS_DIV)
    bxdf = scale(cdiv, scale(lobe, scale(INV_PI, scale(f(0.5), scale(pw + f(2), refl)))))
    return bxdf, lobe * (pw + f(1)) * (f(0.5) * INV_PI)


def lambert_eval(kd, l, v, n):
    return scale(INV_PI, kd), amax(dot(n, l), f(0)) * INV_PI
