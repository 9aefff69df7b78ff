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


# ================================================================================================ geometry.adb / scene.adb
# Independent transcription of the intersectors and of the three integrators, again straight from the Ada text:
#   geometry.adb:48-115 (spheres), :118-143 (flat light), :146-191 (box), :193-229 (Cornell box), :231-263 (triangle), :266-323 (mesh BF)
#   scene.adb:56-86 (Find_Closest_Hit), ray_tracer.adb:61-132 (camera rays, Compute_Shadow), ray_tracer-integrators.adb:82-301
# A scene is a dict: spheres [(pos, r, mat)], cornell {min, max, mat[6], nrm[6]}, light {..}, materials [{type, ...}], mesh {pos, nrm, idx, bbmin, bbmax}.
def intersect_all_spheres(o, d, spheres):
    min_t = INFINITY; min_i = 0
    for i, (c, r, _) in enumerate(spheres):
        k = sub(o, c)
        b = dot(k, d)
        cc = dot(k, k) - r * r
        disc = b * b - cc
        if disc >= 0:
            sq = sqrt(disc)
            t1 = -b - sq; t2 = -b + sq
            if t1 > 0 and t1 < min_t:
                min_t = t1; min_i = i
            elif t2 > 0 and t2 < min_t:
                min_t = t2; min_i = i
    is_hit = bool(min_t > 0 and min_t < INFINITY)
    normal = V(0, 1, 0)
    if not is_hit:
        min_t = f(1)
    else:
        normal = normalize(sub(add(o, scale(min_t, d)), spheres[min_i][0]))
    return dict(is_hit=is_hit, t=min_t, normal=normal, mat=spheres[min_i][2], prim=("sphere", min_i))


def intersect_box(o, d, bmin, bmax):
    with np.errstate(divide="ignore", invalid="ignore"):
        ix, iy, iz = f(1) / d[0], f(1) / d[1], f(1) / d[2]
        lo, hi = (bmax[0] - o[0]) * ix, (bmin[0] - o[0]) * ix
        lo1, hi1 = (bmax[1] - o[1]) * iy, (bmin[1] - o[1]) * iy
        lo2, hi2 = (bmax[2] - o[2]) * iz, (bmin[2] - o[2]) * iz
    tmin, tmax = amin(lo, hi), amax(lo, hi)
    tmin = amax(tmin, amin(lo1, hi1)); tmax = amin(tmax, amax(lo1, hi1))
    tmin = amax(tmin, amin(lo2, hi2)); tmax = amin(tmax, amax(lo2, hi2))
    return bool(tmax > 0 and tmin <= tmax), tmin, tmax


def intersect_cornell(o, d, cb):
    hit, _, tmax = intersect_box(o, d, cb["min"], cb["max"])
    if not hit:
        return dict(is_hit=False, t=f(0))
    p = add(o, scale(tmax, d))
    eps = f(1.0e-5); plane = 0
    if abs(p[0] - cb["min"][0]) < eps: plane = 0
    if abs(p[0] - cb["max"][0]) < eps: plane = 1
    if abs(p[1] - cb["min"][1]) < eps: plane = 2
    if abs(p[1] - cb["max"][1]) < eps: plane = 3
    if abs(p[2] - cb["min"][2]) < eps: plane = 4
    if abs(p[2] - cb["max"][2]) < eps: plane = 5
    return dict(is_hit=plane != 5, t=tmax, normal=cb["nrm"][plane], mat=cb["mat"][plane], prim=("plane", plane))


def intersect_triangle(o, d, A, B, C, t_min, t_max):
    e1, e2 = sub(B, A), sub(C, A)
    pv = cross(d, e2)
    tv = sub(o, A)
    qv = cross(tv, e1)
    inv = f(1) / amax(dot(e1, pv), f(1.0e-25))
    v = dot(tv, pv) * inv
    u = dot(qv, d) * inv
    t = dot(e2, qv) * inv
    if v > 0 and u > 0 and u + v < 1 and t > t_min and t < t_max:
        return dict(is_hit=True, u=u, v=v, tmin=t, tmax=t + f(1.0e-6))
    return dict(is_hit=False)


def intersect_mesh_bf(o, d, mesh):
    hit, _, _ = intersect_box(o, d, mesh["bbmin"], mesh["bbmax"])
    if not hit:
        return dict(is_hit=False, t=f(0))
    near = dict(is_hit=False, tmin=f(0), tmax=f(1000000.0), u=f(0), v=f(0)); tri_id = 0
    for i, (a, b, c) in enumerate(mesh["idx"]):
        h = intersect_triangle(o, d, mesh["pos"][a], mesh["pos"][b], mesh["pos"][c], near["tmin"], near["tmax"])
        if h["is_hit"]:
            near = h; tri_id = i
    a, b, c = mesh["idx"][tri_id]
    w = f(1) - near["u"] - near["v"]
    n = add(add(scale(w, mesh["nrm"][a]), scale(near["v"], mesh["nrm"][b])), scale(near["u"], mesh["nrm"][c]))
    return dict(is_hit=near["is_hit"], t=near["tmin"], normal=n, mat=2, prim=("triangle", tri_id))


def intersect_flat_light(o, d, light):
    with np.errstate(divide="ignore", invalid="ignore"):
        inv_y = f(1) / d[1]
        t = (light["boxMax"][1] - o[1]) * inv_y
    hp = add(o, scale(t, d))
    hit = bool(hp[0] > light["boxMin"][0] and hp[0] < light["boxMax"][0] and hp[2] > light["boxMin"][2] and hp[2] < light["boxMax"][2] and t >= 0)
    return dict(is_hit=hit, t=t, normal=V(0, -1, 0), mat=4, prim=("quad", 0))


def find_closest_hit(scn, o, d):
    hits = [intersect_all_spheres(o, d, scn["spheres"]), intersect_cornell(o, d, scn["cornell"])]
    hits.append(intersect_flat_light(o, d, scn["light"]) if scn["light"]["shape"] == 0 else dict(is_hit=False, t=f(0)))
    hits.append(intersect_mesh_bf(o, d, scn["mesh"]))
    nearest = 0; dist = INFINITY
    for i, h in enumerate(hits):
        if h["is_hit"] and h["t"] < dist:
            nearest = i; dist = h["t"]
    return hits[nearest]


def compute_shadow(scn, hit_pos, lpos):
    eps = amax(amax3(abs(hit_pos[0]), abs(hit_pos[1]), abs(hit_pos[2])) * f(0.000000001), f(1.0e-30))
    sd = normalize(sub(lpos, hit_pos))
    so = add(hit_pos, scale(eps, sd))
    h = find_closest_hit(scn, so, sd)
    max_dist = length(sub(hit_pos, lpos))
    eps2 = amax(max_dist * f(0.000001), f(1.0e-30))
    return bool(h["is_hit"] and (h["t"] < max_dist - eps2 and h["t"] > f(10) * eps))


def amax3(a, b, c):
    if a >= b and a >= c: return a
    if b >= c and b >= a: return b
    return c


def _material_sample(m, u3, u4, ray_dir, n):
    t = m["type"]
    if t == "lambert": return lambert_sample(m["kd"], u3, u4, ray_dir, n)
    if t == "mirror": return mirror_sample(m["refl"], ray_dir, n)
    if t == "glass": return glass_sample(m["refl"], m["trans"], m["ior"], u3, ray_dir, n)
    if t == "phong": return phong_sample(m["refl"], m["pw"], u3, u4, ray_dir, n)
    raise ValueError(t)


def _material_eval(m, l, v, n):
    t = m["type"]
    if t == "lambert": return lambert_eval(m["kd"], l, v, n)
    if t == "phong": return phong_eval(m["refl"], m["pw"], l, v, n)
    return V(0, 0, 0), f(1)


def _light_sample(light, u1, u2, p):
    return sphere_light_sample(light, u1, u2, p) if light["shape"] == 1 else area_light_sample(light, u1, u2, p)


def _light_eval_pdf(light, p, ray_dir, dist):
    return sphere_light_eval_pdf(light, p) if light["shape"] == 1 else area_light_eval_pdf(light, p, ray_dir, dist)


def mulv(a, b):
    return (a[0] * b[0], a[1] * b[1], a[2] * b[2])


G_EPS = f(1.0e-5)
G_EPS_DIV = f(1.0e-20)


def path_trace(scn, kind, o, d, prev, level, uniforms, max_depth):
    """kind: 'stupid' | 'shadow' | 'mis'.  uniforms(bounce) -> the four uniforms of that bounce (light x2, BSDF x2)."""
    if level == 0:
        return V(0, 0, 0)
    h = find_closest_hit(scn, o, d)
    if not h["is_hit"]:
        return V(0, 0, 0)
    m = scn["materials"][h["mat"]]
    n = h["normal"]
    u = uniforms(max_depth - level)
    if m["type"] == "light":
        if kind == "shadow":
            return V(0, 0, 0)
        if dot(scale(f(-1), d), n) < 0:
            return V(0, 0, 0)
        if kind == "stupid":
            return scn["light"]["intensity"]
        lp = _light_eval_pdf(scn["light"], o, d, h["t"])
        bp = prev["pdf"]
        mis = f(1) if prev["specular"] else bp * bp / (lp * lp + bp * bp)
        return scale(mis, scn["light"]["intensity"])
    explicit = V(0, 0, 0)
    hpos = add(o, scale(h["t"], d))
    if kind != "stupid":
        ls = _light_sample(scn["light"], u[0], u[1], hpos)
        sdir = normalize(sub(ls["pos"], hpos))
        if not compute_shadow(scn, hpos, ls["pos"]):
            bx, bp = _material_eval(m, sdir, scale(f(-1), d), n)
            c1 = amax(dot(sdir, n), f(0))
            with np.errstate(over="ignore", invalid="ignore"):
                if kind == "mis":
                    lp = ls["pdf"]
                    mis = lp * lp / (lp * lp + bp * bp)
                    explicit = scale(mis, mulv(scale(f(1) / amax(lp, G_EPS_DIV), ls["intensity"]), scale(c1, bx)))
                else:
                    explicit = scale(f(1) / amax(ls["pdf"], G_EPS_DIV), mulv(ls["intensity"], scale(c1, bx)))
    ms = _material_sample(m, u[2], u[3], d, n)
    bxv = scale(f(1) / amax(ms["pdf"], G_EPS_DIV), ms["color"])
    ct = dot(ms["dir"], n)
    sg = f(1) if ct >= 0 else f(-1)
    no = add(add(o, scale(h["t"], d)), scale(G_EPS, scale(sg, n)))
    nxt = path_trace(scn, kind, no, ms["dir"], dict(pdf=ms["pdf"], specular=ms["specular"]), level - 1, uniforms, max_depth)
    w = scale(abs(ct), bxv)
    with np.errstate(over="ignore", invalid="ignore"):
        if kind == "stupid":
            return mulv(w, nxt)
        return add(explicit, mulv(w, nxt))


def eye_ray_direction(x, y, ox, oy, width, height):
    """ray_tracer.adb:61-97: ox = oy = 0.5 (no AA) or the 1/3, 2/3 offsets; fov = pi/2"""
    fov = f(np.float64(np.pi) / 2.0)
    half = fov / f(2)
    tan_half = f(np.tan(np.float64(half)))                       # safe_tan: |x| /= pi/2 here
    r = (f(x) + ox - (f(width) / f(2)), f(y) + oy - (f(height) / f(2)), -f(width) / tan_half)
    return normalize(r)


# ================================================================================================ the pyramid's transform chain
# vector_math.adb:85-111 RotationMatrix, generic_vector_math.adb:233-256 "*"(Matrix4, Matrix4), vector_math.adb:137-144 "*"(float4x4, float3),
# scene.adb:194-206 (mtans * mrot * mscale), geometry.adb:593-607 (positions transformed, bounding box of the transformed positions)
def rotation_matrix(angle, axis):
    M = [[f(1) if r == c else f(0) for c in range(4)] for r in range(4)]
    v = normalize(axis)
    ct, st = cos(angle), sin(angle)
    one = f(1)
    M[0][0] = (one - ct) * v[0] * v[0] + ct;          M[0][1] = (one - ct) * v[0] * v[1] - st * v[2]; M[0][2] = (one - ct) * v[0] * v[2] + st * v[1]
    M[1][0] = (one - ct) * v[1] * v[0] + st * v[2];   M[1][1] = (one - ct) * v[1] * v[1] + ct;        M[1][2] = (one - ct) * v[1] * v[2] - st * v[0]
    M[2][0] = (one - ct) * v[0] * v[2] - st * v[1];   M[2][1] = (one - ct) * v[2] * v[1] + st * v[0]; M[2][2] = (one - ct) * v[2] * v[2] + ct
    return M


def mat_mul(a, b):
    return [[a[r][0] * b[0][c] + a[r][1] * b[1][c] + a[r][2] * b[2][c] + a[r][3] * b[3][c] for c in range(4)] for r in range(4)]


def mat_mul_point(m, v):
    return tuple(m[r][0] * v[0] + m[r][1] * v[1] + m[r][2] * v[2] + m[r][3] for r in range(3))


def cornell_mesh_transform():
    ident = lambda: [[f(1) if r == c else f(0) for c in range(4)] for r in range(4)]
    mrot = rotation_matrix(f(-np.float64(np.pi) / 6.0), V(0, 1, 0))          # static -PI/6.0, rounded once
    mscale = ident(); mtans = ident()
    for r, val in enumerate((-0.75, 0.1, 3.1, 1.0)):
        mtans[r][3] = f(val)
    mscale[0][0] = mscale[1][1] = mscale[2][2] = f(2)
    return mat_mul(mat_mul(mtans, mrot), mscale)
