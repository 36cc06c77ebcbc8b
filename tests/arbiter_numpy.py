"""An INDEPENDENT arbiter for the headline pixel: the whole C3 fragment — vertex varyings, setScene, raymarch, sdScene with one
Mandelbulb, getNormal, pnoise / bumpNormal, softshadow, getPhong with directional lights, the orbit-trap colouring of render and
main's composite — written again from the SHADER TEXT (resources/raymarch.vert, resources/raymarch.frag of the reference;
`frag:N` below) in vectorised NumPy float64.

Why it exists (missing #2): the binary64 arbiter of oracle/rm_oracle_f64.c is the oracle's own C source compiled with
`#define float double` — a formula transcribed wrongly into rm_oracle.c would be wrong in both and pass.  This file shares no
source with the oracle: its own control flow (whole frames of rays advance together, finished rays drop out of the index set),
NumPy's arccos / arctan2 / sin / cos / power / log / sqrt, no rm_math contract.  It is test infrastructure only.

One sub-expression is evaluated in binary32 ON PURPOSE, as in the C arbiter: the hash → lattice-gradient map
of the classic Perlin noise (frag:1626-1644).  Its selector gz = 0.5 − |gx| − |gy| is an EXACT tie in real arithmetic for 7 of
the 49 hash classes, so which of two unrelated gradients `step(gz, 0)` picks is DEFINED by binary32 rounding in the shader (every
binary32 implementation with IEEE division agrees; a binary64 evaluation of that line picks other gradients for no meaningful
reason).  Those lines run on np.float32 arrays with NumPy's IEEE operations; everything around them is float64.

Scope: exactly what a frame of scenefiles/simple/unit_mandelbulb.json (C3) or unit_mengersponge.json (C5: sdMengerSponge,
frag:1049-1071, its palette and main's reflection loop, frag:2491-2524) executes with the reference's default #defines
(WHITE_BACKGROUND, PERLIN_BUMP; soft shadows, AO, refraction, sky box off): one object, DIRECTIONAL lights — render_frame — and,
at the end of the file, tables of any of sdMatch's nine primitives (untextured) under directional, point and spot lights with soft
shadows and ambient occlusion and main's reflection loop (C2's class: the geometry of 38 of the reference's scenefiles) —
render_frame_table.  Anything else raises."""
import numpy as np

SURFACE_DIST = 1e-3          # frag:32
FRACTALS_BAILOUT = 2.0       # frag:30
BUMP_SCALE, BUMP_INTENSITY = 10.0, 2.0  # frag:128-129
RM_MANDELBULB, RM_MENGERSPONGE, RM_LIGHT_DIRECTIONAL = 10, 11, 1  # scenedata.h:18-33 / 10-15 ≡ frag:53-75
RM_FEAT_PERLIN_BUMP, RM_FEAT_WHITE_BACKGROUND = 128, 8  # include/raymarcher_amd.h: the shader's #defines as feature bits


def _dot(a, b):
    return (a * b).sum(-1)


def _normalize(v):
    return v / np.sqrt(_dot(v, v))[..., None]


def _mix(a, b, t):
    return a * (1.0 - t) + b * t


# ---------------------------------------------------------------------------------------------- frag:775-803
def sd_mandelbulb(pos, power, iters, julia=(0.0, 0.0)):
    """sdMandelBulb: returns (distance estimate, resColor = (m, trap.y, trap.z, trap.w)).  A point that has bailed out stops
    updating (the shader's `break`)."""
    w = np.array(pos, dtype=np.float64)
    c = w.copy()
    if np.hypot(*julia) != 0.0:  # frag:782-784
        c = np.broadcast_to(np.array([julia[0], julia[1], 0.0]), w.shape).copy()
    m = _dot(w, w)
    trap = np.concatenate([np.abs(w), m[:, None]], -1)
    dz = np.ones(len(w))
    live = np.ones(len(w), dtype=bool)
    with np.errstate(all="ignore"):
        for _ in range(iters):
            dz_n = power * np.power(m, (power - 1.0) / 2.0) * dz + 1.0
            r = np.sqrt(_dot(w, w))
            b = power * np.arccos(w[:, 1] / r)
            a = power * np.arctan2(w[:, 0], w[:, 2])
            w_n = c + np.power(r, power)[:, None] * np.stack([np.sin(b) * np.sin(a), np.cos(b), np.sin(b) * np.cos(a)], -1)
            trap_n = np.minimum(trap, np.concatenate([np.abs(w_n), m[:, None]], -1))
            m_n = _dot(w_n, w_n)
            dz = np.where(live, dz_n, dz)
            w = np.where(live[:, None], w_n, w)
            trap = np.where(live[:, None], trap_n, trap)
            m = np.where(live, m_n, m)
            live = live & ~(m > FRACTALS_BAILOUT)
            if not live.any():
                break
        d = 0.25 * np.log(m) * np.sqrt(m) / dz
    return d, np.concatenate([m[:, None], trap[:, 1:]], -1)


# ---------------------------------------------------------------------------------------------- frag:1586-1676
def _permute(x):  # frag:1602-1604; exact integers below 2^24 in either precision
    return np.mod((x * 34.0 + 1.0) * x, 289.0)


def _lattice_gradients32(ixy):
    """frag:1626-1634 (and 1636-1644) on np.float32 arrays: the hash → gradient map whose tie-break binary32 defines."""
    f = np.float32
    ixy = ixy.astype(f)
    gx = ixy / f(7.0)
    t = np.floor(gx) / f(7.0)
    gy = (t - np.floor(t)) - f(0.5)
    gx = gx - np.floor(gx)
    gz = f(0.5) - np.abs(gx) - np.abs(gy)
    sz = (gz <= f(0.0)).astype(f)                       # step(gz, 0)
    gx = gx - sz * ((gx >= f(0.0)).astype(f) - f(0.5))  # step(0, gx)
    gy = gy - sz * ((gy >= f(0.0)).astype(f) - f(0.5))
    return gx.astype(np.float64), gy.astype(np.float64), gz.astype(np.float64)


def pnoise(p):
    """Classic Perlin noise, 3-D (frag:1610-1676)."""
    p = np.asarray(p, dtype=np.float64)
    Pi0 = np.floor(p)
    Pi1 = np.mod(Pi0 + 1.0, 256.0)
    Pi0 = np.mod(Pi0, 256.0)
    Pf0 = p - np.floor(p)
    Pf1 = Pf0 - 1.0
    n = {}
    for cz in (0, 1):
        iz = (Pi1 if cz else Pi0)[:, 2]
        for cy in (0, 1):
            iy = (Pi1 if cy else Pi0)[:, 1]
            for cx in (0, 1):
                ix = (Pi1 if cx else Pi0)[:, 0]
                gx, gy, gz = _lattice_gradients32(_permute(_permute(_permute(ix) + iy) + iz))
                g = np.stack([gx, gy, gz], -1)
                g = g * (1.79284291400159 - 0.85373472095314 * _dot(g, g))[:, None]  # taylorInvSqrt, frag:1606-1608
                off = np.stack([(Pf1 if cx else Pf0)[:, 0], (Pf1 if cy else Pf0)[:, 1], (Pf1 if cz else Pf0)[:, 2]], -1)
                n[(cx, cy, cz)] = _dot(g, off)
    t = Pf0
    fade = t * t * t * (t * (t * 6.0 - 15.0) + 10.0)  # frag:1587-1589
    nz = {(cx, cy): _mix(n[(cx, cy, 0)], n[(cx, cy, 1)], fade[:, 2]) for cx in (0, 1) for cy in (0, 1)}
    ny = {cx: _mix(nz[(cx, 0)], nz[(cx, 1)], fade[:, 1]) for cx in (0, 1)}
    return 2.2 * _mix(ny[0], ny[1], fade[:, 0])


def bump_normal(normal, pos, scale=BUMP_SCALE, intensity=BUMP_INTENSITY):  # frag:1679-1691
    q = pos * scale
    n0 = pnoise(q)
    grad = np.stack([pnoise(q + np.array([0.1, 0.0, 0.0])) - n0, pnoise(q + np.array([0.0, 0.1, 0.0])) - n0,
                     pnoise(q + np.array([0.0, 0.0, 0.1])) - n0], -1)
    return _normalize(normal + grad * intensity)


# ---------------------------------------------------------------------------------------------- the frame
class Bulb:
    """What sdScene (frag:1406-1430) does with a table of one Mandelbulb: object-space point, estimate × scaleFactor."""

    def __init__(self, inv_model, scale_factor, power, iters, julia):
        self.M, self.sf, self.power, self.iters, self.julia = np.asarray(inv_model, np.float64), float(scale_factor), float(power), int(iters), julia

    def __call__(self, p):
        po = p @ self.M[:3, :3].T + self.M[:3, 3]  # vec3(invModelMatrix · vec4(p, 1)), frag:1417
        d, res = sd_mandelbulb(po, self.power, self.iters, self.julia)
        return d * self.sf, res


def raymarch(sd, ro, rd, end, max_steps, shadow=False):
    """frag:1453-1484 (shadow=False, side = +1) and the march of frag:1703-1725 (shadow=True: t += |d|, start at mint = 0).
    Returns (hit, depth as raymarch reports it: rayDepth − minD, trap of the last evaluation)."""
    n = len(ro)
    t = np.zeros(n)
    d = np.full(n, 1000000.0)
    trap = np.zeros((n, 4))
    idx = np.arange(n)
    for _ in range(max_steps):
        if len(idx) == 0:
            break
        dd, tr = sd(ro[idx] + rd[idx] * t[idx, None])
        d[idx], trap[idx] = dd, tr
        stop = (np.abs(dd) < SURFACE_DIST) | (t[idx] > end)
        go = ~stop
        t[idx[go]] += np.abs(dd[go]) if shadow else dd[go]
        idx = idx[go]
    hit = np.abs(d) < SURFACE_DIST
    return hit, t - d, trap


def get_normal(sd, p):  # frag:1436-1444
    e = np.array([1.0, -1.0, 0.0]) * 0.5773 * 0.0005
    x, y = e[0], e[1]
    taps = [np.array([x, y, y]), np.array([y, y, x]), np.array([y, x, y]), np.array([x, x, x])]  # e.xyy, e.yyx, e.yxy, e.xxx
    acc = np.zeros_like(p)
    for k in taps:
        acc = acc + k * sd(p + k)[0][:, None]
    return _normalize(acc)


def get_phong(sd, N, obj, lights, g, p, rd, far, max_steps):
    """frag:1842-1933 for an untextured object and DIRECTIONAL lights, soft shadows and ambient occlusion off."""
    ka, kd, ks = g
    total = np.broadcast_to(obj["cAmbient"] * ka, p.shape).copy()  # ao = 1
    V = _normalize(-rd)
    for li in lights:
        L = np.broadcast_to(_normalize(-li["dir"]), p.shape)
        occluded, _, _ = raymarch(sd, p + N * SURFACE_DIST * 5.0, L, far, max_steps, shadow=True)  # frag:1908
        ndl = _dot(N, L)
        lit = ~occluded & ~(ndl <= 0.005)  # frag:1909-1912
        col = (kd * obj["cDiffuse"]) * np.clip(ndl, 0.0, 1.0)[:, None] * li["color"]  # getDiffuse, texLoc == −1: frag:1749-1752
        R = (-L) - 2.0 * _dot(N, -L)[:, None] * N  # reflect(−L, N)
        rdv = np.clip(_dot(R, V), 0.0, 1.0)
        spec = ks * rdv if obj["shininess"] == 0 else ks * np.power(rdv, obj["shininess"])  # frag:1787-1792
        col = col + spec[:, None] * obj["cSpecular"] * li["color"]
        total = total + np.where(lit[:, None], col, 0.0)  # fAtt = aFall = 1 for a directional light
    return total


class Menger:
    """sdScene with a table of one Menger sponge (frag:1406-1430, 1049-1071).  `levels` is the loop bound of frag:1056 (4 in the
    shader, a build knob of the product); resColor = (d, min(0.2·da·db·dc), (1 + m)/4, 0) of the last level that raised d."""

    def __init__(self, inv_model, scale_factor, levels, itime):
        self.M, self.sf, self.levels, self.itime = np.asarray(inv_model, np.float64), float(scale_factor), int(levels), float(itime)

    def __call__(self, p):
        p = p @ self.M[:3, :3].T + self.M[:3, 3]
        q = np.abs(p) - 1.0                                        # sdBox(p, vec3(1)), frag:843-846
        d = np.sqrt(_dot(np.maximum(q, 0.0), np.maximum(q, 0.0))) + np.minimum(np.max(q, axis=-1), 0.0)
        res = np.stack([d, np.ones_like(d), np.zeros_like(d), np.zeros_like(d)], -1)
        x = -np.cos(0.5 * self.itime)                              # smoothstep(−0.2, 0.2, −cos(0.5·iTime)), frag:1052
        t = np.clip((x + 0.2) / 0.4, 0.0, 1.0)
        ani = t * t * (3.0 - 2.0 * t)
        off = 1.5 * np.sin(0.01 * self.itime)
        ma = np.array([[0.60, 0.00, -0.80], [0.00, 1.00, 0.00], [0.80, 0.00, 0.60]])  # columns (.6,0,.8), (0,1,0), (−.8,0,.6): frag:124-126
        s = 1.0
        for m in range(self.levels):
            p = _mix(p, (p + off) @ ma.T, ani)                     # frag:1057
            a = np.mod(p * s, 2.0) - 1.0
            s *= 3.0
            r = np.abs(1.0 - 3.0 * np.abs(a))
            da, db, dc = np.maximum(r[:, 0], r[:, 1]), np.maximum(r[:, 1], r[:, 2]), np.maximum(r[:, 2], r[:, 0])
            c = (np.minimum(da, np.minimum(db, dc)) - 1.0) / s
            up = c > d
            d = np.where(up, c, d)
            res = np.where(up[:, None], np.stack([d, np.minimum(res[:, 1], 0.2 * da * db * dc), np.full_like(d, (1.0 + m) / 4.0),
                                                   np.zeros_like(d)], -1), res)
        return d * self.sf, res


def render_rays(sd, kind, obj, lights, g, ro, rd, far, settings, bg):
    """render() (frag:2318-2375) for a batch of rays: (rgb, isEnv, hit point, shading normal).  Point and normal are only
    meaningful where isEnv is False (the `out IntersectionInfo` of a hit)."""
    n = len(ro)
    rgb = np.broadcast_to(bg, (n, 3)).copy()                # a miss: vec4(bgCol, 1), frag:2323-2329
    hit, depth, trap = raymarch(sd, ro, rd, far, settings.maxSteps)
    P, N = np.zeros((n, 3)), np.zeros((n, 3))
    if hit.any():
        p = ro[hit] + rd[hit] * depth[hit, None]             # frag:2333
        pn = get_normal(sd, p)
        if settings.features & RM_FEAT_PERLIN_BUMP:          # #define PERLIN_BUMP (frag:15, 2334-2336)
            pn = bump_normal(pn, p)
        ph = get_phong(sd, pn, obj, lights, (g.ka, g.kd, g.ks), p, rd[hit], far, settings.maxSteps)
        tr = trap[hit]
        if kind == "bulb":                                   # frag:2354-2361
            col = np.full((len(p), 3), 0.2)
            col = _mix(col, np.array([0.10, 0.20, 0.30]), np.clip(tr[:, 1], 0.0, 1.0)[:, None])
            col = _mix(col, np.array([0.02, 0.10, 0.30]), np.clip(tr[:, 2] * tr[:, 2], 0.0, 1.0)[:, None])
            col = _mix(col, np.array([0.30, 0.10, 0.02]), np.clip(np.power(tr[:, 3], 6.0), 0.0, 1.0)[:, None])
            col = col * 0.5
            col = col * (ph * 8.0)
        else:                                                # MENGERSPONGE, frag:2362-2365
            col = (0.5 + 0.5 * np.cos(np.array([0.0, 1.0, 2.0]) + 2.0 * tr[:, 2:3])) * ph
        rgb[hit], P[hit], N[hit] = col, p, pn
    return rgb, ~hit, P, N


def render_frame(tables, settings, W, H):
    """fragColor of every pixel, (H, W, 4) float64, row 0 = bottom (gl_FragCoord convention): a table of one Mandelbulb or one
    Menger sponge under directional lights; main's reflection loop (frag:2491-2524) when it is enabled."""
    assert tables.num_objects == 1 and tables.objects[0].type in (RM_MANDELBULB, RM_MENGERSPONGE) and tables.objects[0].texLoc == -1
    assert not (settings.enableSoftShadow or settings.enableAmbientOcclusion or settings.enableRefraction or settings.enableSkyBox)
    assert not tables.globals_.isTwoD and settings.features & RM_FEAT_WHITE_BACKGROUND
    o = tables.objects[0]
    obj = {"cAmbient": np.array(list(o.cAmbient), np.float64), "cDiffuse": np.array(list(o.cDiffuse), np.float64),
           "cSpecular": np.array(list(o.cSpecular), np.float64), "shininess": float(o.shininess)}
    c_refl = np.array(list(o.cReflective), np.float64)
    assert not o.isEmissive
    lights = []
    for i in range(tables.num_lights):
        li = tables.lights[i]
        assert li.type == RM_LIGHT_DIRECTIONAL
        lights.append({"dir": np.array(list(li.dir), np.float64), "color": np.array(list(li.color), np.float64)})
    gl = tables.globals_
    inv_model = np.array(list(o.invModel), np.float64).reshape(4, 4).T
    if o.type == RM_MANDELBULB:
        kind, sd = "bulb", Bulb(inv_model, o.scaleFactor, gl.power, settings.fractalIters, (gl.juliaSeed[0], gl.juliaSeed[1]))
    else:
        kind, sd = "menger", Menger(inv_model, o.scaleFactor, settings.mengerLevels, gl.iTime)
    inv_pv = np.array(list(tables.camera.invProjView), np.float64).reshape(4, 4).T  # column-major storage
    far = float(tables.camera.initialFar)
    # raymarch.vert:13-25: nearClip / farClip are affine in the quad position, so their interpolated value at a pixel centre
    # IS the matrix product at that position (in real arithmetic; DESIGN.md §2.3 is about binary32)
    ys, xs = np.mgrid[0:H, 0:W]
    ndc = np.stack([(xs.ravel() + 0.5) / W * 2.0 - 1.0, (ys.ravel() + 0.5) / H * 2.0 - 1.0], -1)
    near = np.concatenate([ndc, np.full((len(ndc), 1), -1.0), np.ones((len(ndc), 1))], -1) @ inv_pv.T
    farc = np.concatenate([ndc, np.ones((len(ndc), 1)), np.ones((len(ndc), 1))], -1) @ inv_pv.T
    ro = near[:, :3] / near[:, 3:]                      # setScene, frag:2388-2392
    rd = _normalize(farc[:, :3] / farc[:, 3:] - ro)
    bg = np.array([1.0, 1.0, 1.0])                      # WHITE_BACKGROUND, frag:2414-2416
    out = np.ones((len(ndc), 4))
    rgb, is_env, P, N = render_rays(sd, kind, obj, lights, gl, ro, rd, far, settings, bg)  # frag:2443
    out[:, :3] = rgb                                    # a miss returns here (frag:2459-2465); a hit: phong = ri.fragColor
    if settings.enableReflection and np.sqrt(_dot(c_refl, c_refl)) != 0.0:  # frag:2491-2524
        idx = np.nonzero(~is_env)[0]
        p, n, d = P[idx], N[idx], rd[idx]
        fil = np.ones(3)
        for _ in range(settings.numReflection):
            if len(idx) == 0:
                break
            r = d - 2.0 * _dot(n, d)[:, None] * n       # reflect(info.rd, info.n)
            sro = p + r * SURFACE_DIST * 3.0
            fil = fil * c_refl                          # the FIRST hit's material throughout (frag:2500-2501), one object here
            rgb, env, P2, N2 = render_rays(sd, kind, obj, lights, gl, sro, r, far, settings, bg)
            out[idx, :3] += gl.ks * fil * rgb           # refl += vec4(ks·fil·res.rgb, 1)
            out[idx, 3] += 1.0
            go = ~env                                   # `if (res.isEnv) break;`
            idx, p, n, d = idx[go], P2[go], N2[go], r[go]
    return out.reshape(H, W, 4), (~is_env).reshape(H, W)


# ---------------------------------------------------------------------------------------------- tables of primitives (C2)
# Round 4, late: the lighting configuration (scenefiles/lighting/directional_light_2.json with soft shadows and ambient occlusion)
# transcribed independently as well — sdScene over a table (frag:1406-1430), sdMatch's nine primitives (frag:832-896, 991-1019,
# 1262-1280), softshadow with its penumbra factor (frag:1703-1725; UB1 of DESIGN.md §4: r.d is the factor on a miss too), calcAO
# (frag:1729-1740), getPhong with directional, point and spot lights (frag:1842-1933, 439-461).
RM_CUBE, RM_CONE, RM_CYLINDER, RM_SPHERE, RM_LIGHT_POINT, RM_LIGHT_SPOT = 0, 1, 2, 3, 0, 2


def _len2(a, b):
    return np.sqrt(a * a + b * b)


def sd_cube(p):      # sdBox(p, vec3(0.5)), frag:843-846
    q = np.abs(p) - 0.5
    return np.sqrt(_dot(np.maximum(q, 0.0), np.maximum(q, 0.0))) + np.minimum(np.max(q, axis=-1), 0.0)


def sd_cone(p, r=0.5, h=0.5):  # frag:853-862
    po = np.stack([_len2(p[:, 0], p[:, 2]) - r, p[:, 1] + h], -1)
    e = np.array([-r, 2.0 * h])
    q = po - e * np.clip((po @ e) / (e @ e), 0.0, 1.0)[:, None]
    d = _len2(q[:, 0], q[:, 1])
    return np.where(np.maximum(q[:, 0], q[:, 1]) > 0.0, d, -np.minimum(d, po[:, 1]))


def sd_cylinder(p, h=0.5, r=0.5):  # frag:869-872
    d = np.abs(np.stack([_len2(p[:, 0], p[:, 2]), p[:, 1]], -1)) - np.array([r, h])
    return np.minimum(np.maximum(d[:, 0], d[:, 1]), 0.0) + _len2(np.maximum(d[:, 0], 0.0), np.maximum(d[:, 1], 0.0))


def sd_sphere(p, r=0.5):  # frag:832-834
    return np.sqrt(_dot(p, p)) - r


def sd_octahedron(p, s=0.5):  # frag:877-888
    p = np.abs(p)
    m = p[:, 0] + p[:, 1] + p[:, 2] - s
    r = 3.0 * p - m[:, None]
    q = np.where((r[:, 0] < 0.0)[:, None], p, np.where((r[:, 1] < 0.0)[:, None], p[:, [1, 2, 0]], p[:, [2, 0, 1]]))
    k = np.clip(0.5 * (q[:, 2] - q[:, 1] + s), 0.0, s)
    inside = np.sqrt(q[:, 0] ** 2 + (q[:, 1] - s + k) ** 2 + (q[:, 2] - k) ** 2)
    return np.where((r[:, 0] < 0.0) | (r[:, 1] < 0.0) | (r[:, 2] < 0.0), inside, m * 0.57735027)


def sd_torus(p, t=(0.5, 0.5 / 4)):  # frag:893-896
    return _len2(_len2(p[:, 0], p[:, 2]) - t[0], p[:, 1]) - t[1]


def sd_capsule(p, h=0.5, r=0.1):  # frag:991-994
    y = p[:, 1] - np.clip(p[:, 1], 0.0, h)
    return np.sqrt(p[:, 0] ** 2 + y ** 2 + p[:, 2] ** 2) - r


def sd_deathstar(p2, ra=0.5, rb=0.35, d=0.5):  # frag:1005-1019
    px, py = p2[:, 0], _len2(p2[:, 1], p2[:, 2])
    a = (ra * ra - rb * rb + d * d) / (2.0 * d)
    b = np.sqrt(max(ra * ra - a * a, 0.0))
    rim = _len2(px - a, py - b)
    body = np.maximum(_len2(px, py) - ra, -(_len2(px - d, py) - rb))
    return np.where(px * b - py * a > d * np.maximum(b - py, 0.0), rim, body)


def sd_rectangle(p):  # sdBox(p, vec3(0.5, 0.5, 0)), frag:1279
    q = np.abs(p) - np.array([0.5, 0.5, 0.0])
    return np.sqrt(_dot(np.maximum(q, 0.0), np.maximum(q, 0.0))) + np.minimum(np.max(q, axis=-1), 0.0)


def sd_sierpinski(p):  # frag:807-826: 14 fold-and-scale iterations (Scale 1.85, Offset 2), length(p)·Scale^−14
    # Scale is the shader's binary32 constant 1.85 (1.850000023841858): fourteen scalings amplify the 2.4e-8 between it and the
    # binary64 1.85 to 1e-3 of a pixel — the constant is part of the function, not of the precision it is evaluated in
    scale = float(np.float32(1.85))
    p = p.copy()
    for _ in range(14):
        f = p[:, 0] + p[:, 1] < 0.0
        p[f, 0], p[f, 1] = -p[f, 1], -p[f, 0]       # p.xy = −p.yx
        f = p[:, 0] + p[:, 2] < 0.0
        p[f, 0], p[f, 2] = -p[f, 2], -p[f, 0]       # p.xz = −p.zx
        f = p[:, 1] + p[:, 2] < 0.0
        p[f, 2], p[f, 1] = -p[f, 1], -p[f, 2]       # p.zy = −p.yz
        p = p * scale - 2.0 * (scale - 1.0)
    return np.sqrt(_dot(p, p)) * np.power(scale, -14.0)


class Table:
    """sdScene (frag:1406-1430) over a table of primitives: the minimum of sdMatch(po)·scaleFactor with a strict `<` (the first of
    equal objects wins); the index of the minimum travels in the first component of what the fractal classes call the trap."""
    SDF = {RM_CUBE: sd_cube, RM_CONE: sd_cone, RM_CYLINDER: sd_cylinder, RM_SPHERE: sd_sphere, 4: sd_octahedron, 5: sd_torus,
           6: sd_capsule, 7: sd_deathstar, 8: sd_rectangle, 12: sd_sierpinski}  # scenedata.h's PrimitiveType order ≡ frag:53-66

    def __init__(self, objects):
        self.objs = [(self.SDF[t], np.asarray(M, np.float64), float(sf)) for t, M, sf in objects]

    def __call__(self, p):
        best = np.full(len(p), 1000000.0)
        idx = np.full(len(p), -1.0)
        for k, (f, M, sf) in enumerate(self.objs):
            d = f(p @ M[:3, :3].T + M[:3, 3]) * sf
            closer = d < best
            best, idx = np.where(closer, d, best), np.where(closer, float(k), idx)
        return best, np.stack([idx, np.zeros_like(idx), np.zeros_like(idx), np.zeros_like(idx)], -1)


def softshadow(sd, ro, rd, maxt, max_steps, k=8.0):
    """frag:1703-1725 with mint = 0: (hit, res).  res = min over the steps of k·d/t — the first step divides by t = 0: +inf for d > 0,
    which min drops, as in the shader."""
    n = len(ro)
    t, res, d = np.zeros(n), np.ones(n), np.full(n, 1000000.0)
    idx = np.arange(n)
    with np.errstate(divide="ignore", invalid="ignore"):
        for _ in range(max_steps):
            if len(idx) == 0:
                break
            dd = sd(ro[idx] + rd[idx] * t[idx, None])[0]
            d[idx] = dd
            go = ~((np.abs(dd) < SURFACE_DIST) | (t[idx] > maxt[idx]))
            g = idx[go]
            res[g] = np.minimum(res[g], k * dd[go] / t[g])
            t[g] += np.abs(dd[go])
            idx = g
    return np.abs(d) < SURFACE_DIST, res


def calc_ao(sd, pos, nor):  # frag:1729-1740
    occ, sca = np.zeros(len(pos)), 1.0
    live = np.ones(len(pos), bool)
    for i in range(5):
        hgt = 0.01 + 0.12 * float(i) / 4.0
        d = sd(pos + hgt * nor)[0]
        occ = np.where(live, occ + (hgt - d) * sca, occ)
        sca *= 0.95
        live = live & ~(occ > 0.35)
    return np.clip(1.0 - 3.0 * occ, 0.0, 1.0) * (0.5 + 0.5 * nor[:, 1])


def get_phong_table(sd, N, mats, lights, g, p, rd, far, settings):
    """frag:1842-1933 for untextured objects (one material ROW per point in `mats`), directional and point lights, with the soft
    shadow factor and ambient occlusion when the settings ask for them."""
    ka, kd, ks = g
    ao = calc_ao(sd, p, N) if settings.enableAmbientOcclusion else np.ones(len(p))
    total = mats["cAmbient"] * ka * ao[:, None]
    V = _normalize(-rd)
    for li in lights:
        if li["type"] in (RM_LIGHT_POINT, RM_LIGHT_SPOT):
            to = li["pos"] - p
            dist = np.sqrt(_dot(to, to))
            L, maxt = to / dist[:, None], dist
            f = li["func"]
            with np.errstate(divide="ignore"):
                f_att = np.minimum(1.0 / (f[0] + dist * f[1] + dist * dist * f[2]), 1.0)  # attenuationFactor, frag:445-447
            if li["type"] == RM_LIGHT_SPOT:  # angularFalloff, frag:439-461
                cosalpha = _dot(np.broadcast_to(-_normalize(li["dir"]), p.shape), L)
                inner = li["angle"] - li["penumbra"]
                tt = (np.arccos(np.clip(cosalpha, -1.0, 1.0)) - inner) / (li["angle"] - inner)
                fall = 1.0 - (-2.0 * tt ** 3 + 3.0 * tt ** 2)
                f_att = f_att * np.where(cosalpha <= np.cos(li["angle"]), 0.0, np.where(cosalpha > np.cos(inner), 1.0, fall))
        else:
            L = np.broadcast_to(_normalize(-li["dir"]), p.shape)
            maxt, f_att = np.full(len(p), far), np.ones(len(p))
        occluded, pen = softshadow(sd, p + N * SURFACE_DIST * 5.0, L, maxt, settings.maxSteps)  # frag:1908
        ndl = _dot(N, L)
        lit = ~occluded & ~(ndl <= 0.005)
        col = (kd * mats["cDiffuse"]) * np.clip(ndl, 0.0, 1.0)[:, None] * li["color"]
        R = (-L) - 2.0 * _dot(N, -L)[:, None] * N
        rdv = np.clip(_dot(R, V), 0.0, 1.0)
        sh = mats["shininess"]
        with np.errstate(invalid="ignore"):
            spec = np.where(sh == 0.0, ks * rdv, ks * np.power(rdv, np.where(sh == 0.0, 1.0, sh)))  # getSpecular, frag:1787-1792
        col = (col + spec[:, None] * mats["cSpecular"] * li["color"]) * f_att[:, None]
        if settings.enableSoftShadow:
            col = col * pen[:, None]  # frag:1928 (UB1: the penumbra factor, hit or miss)
        total = total + np.where(lit[:, None], col, 0.0)
    return total


def _refract(I, Nn, eta):  # GLSL refract (spec 8.5): k = 1 − η²(1 − (N·I)²); k < 0 → 0, else η·I − (η·(N·I) + √k)·N
    eta = np.asarray(eta, np.float64) * np.ones(len(I))
    ni = _dot(Nn, I)
    k = 1.0 - eta * eta * (1.0 - ni * ni)
    with np.errstate(invalid="ignore"):
        r = eta[:, None] * I - (eta * ni + np.sqrt(k))[:, None] * Nn
    return np.where((k < 0.0)[:, None], 0.0, r)


def _raymarch_depth(sd, ro, rd, end, max_steps, side):
    """raymarch(…).d (frag:1453-1484) with `side`: rayDepth − minD on a hit, rayDepth on a miss (UB2)."""
    n = len(ro)
    t, d = np.zeros(n), np.full(n, 1000000.0)
    idx = np.arange(n)
    for _ in range(max_steps):
        if len(idx) == 0:
            break
        dd = sd(ro[idx] + rd[idx] * t[idx, None])[0]
        d[idx] = dd
        go = ~((np.abs(dd) < SURFACE_DIST) | (t[idx] > end))
        t[idx[go]] += dd[go] * side
        idx = idx[go]
    return np.where(np.abs(d) < SURFACE_DIST, t - d, t)


def _table_rays(sd, mats, lights, g, ro, rd, far, settings):
    """render() (frag:2318-2375) of a table of primitives for a batch of rays: (rgb, isEnv, hit point, shading normal, object)."""
    n = len(ro)
    rgb = np.ones((n, 3))                                   # WHITE_BACKGROUND
    P, N, K = np.zeros((n, 3)), np.zeros((n, 3)), np.full(n, -1)
    hit, depth, trap = raymarch(sd, ro, rd, far, settings.maxSteps)
    if hit.any():
        p = ro[hit] + rd[hit] * depth[hit, None]
        pn = get_normal(sd, p)
        if settings.features & RM_FEAT_PERLIN_BUMP:
            pn = bump_normal(pn, p)
        k = trap[hit, 0].astype(int)
        m = {key: v[k] for key, v in mats.items() if key in ("cAmbient", "cDiffuse", "cSpecular", "shininess")}
        rgb[hit] = get_phong_table(sd, pn, m, lights, (g.ka, g.kd, g.ks), p, rd[hit], far, settings)
        P[hit], N[hit], K[hit] = p, pn, k
    return rgb, ~hit, P, N, K


def render_frame_table(tables, settings, W, H):
    """fragColor of every pixel of a table of untextured primitives under directional, point and spot lights: (H, W, 4) float64 and
    the hit mask; main's reflection loop (frag:2491-2524: the FIRST hit's cReflective filters every bounce) and its two-interface
    refraction (frag:2526-2570) when they are enabled."""
    assert not settings.enableSkyBox and not tables.globals_.isTwoD
    assert settings.features & RM_FEAT_WHITE_BACKGROUND
    objs, mats = [], {"cAmbient": [], "cDiffuse": [], "cSpecular": [], "cReflective": [], "cTransparent": [], "shininess": [], "ior": []}
    for i in range(tables.num_objects):
        o = tables.objects[i]
        assert o.type in Table.SDF and o.texLoc == -1 and not o.isEmissive
        objs.append((o.type, np.array(list(o.invModel), np.float64).reshape(4, 4).T, o.scaleFactor))
        for k in ("cAmbient", "cDiffuse", "cSpecular", "cReflective", "cTransparent"):
            mats[k].append(list(getattr(o, k)))
        mats["shininess"].append(float(o.shininess))
        mats["ior"].append(float(o.ior))
    mats = {k: np.array(v, np.float64) for k, v in mats.items()}
    lights = []
    for i in range(tables.num_lights):
        li = tables.lights[i]
        assert li.type in (RM_LIGHT_DIRECTIONAL, RM_LIGHT_POINT, RM_LIGHT_SPOT)
        lights.append({"type": li.type, "dir": np.array(list(li.dir), np.float64), "pos": np.array(list(li.pos), np.float64),
                       "func": np.array(list(li.func), np.float64), "color": np.array(list(li.color), np.float64),
                       "angle": float(li.angle), "penumbra": float(li.penumbra)})
    gl = tables.globals_
    sd = Table(objs)
    inv_pv = np.array(list(tables.camera.invProjView), np.float64).reshape(4, 4).T
    far = float(tables.camera.initialFar)
    ys, xs = np.mgrid[0:H, 0:W]
    ndc = np.stack([(xs.ravel() + 0.5) / W * 2.0 - 1.0, (ys.ravel() + 0.5) / H * 2.0 - 1.0], -1)
    near = np.concatenate([ndc, np.full((len(ndc), 1), -1.0), np.ones((len(ndc), 1))], -1) @ inv_pv.T
    farc = np.concatenate([ndc, np.ones((len(ndc), 1)), np.ones((len(ndc), 1))], -1) @ inv_pv.T
    ro = near[:, :3] / near[:, 3:]
    rd = _normalize(farc[:, :3] / farc[:, 3:] - ro)
    out = np.ones((len(ndc), 4))
    rgb, is_env, P, N, K = _table_rays(sd, mats, lights, gl, ro, rd, far, settings)
    out[:, :3] = rgb
    if settings.enableReflection:  # frag:2491-2524
        c_refl = mats["cReflective"][np.maximum(K, 0)]
        idx = np.nonzero(~is_env & (np.sqrt(_dot(c_refl, c_refl)) != 0.0))[0]
        p, n, d, cr = P[idx], N[idx], rd[idx], c_refl[idx]
        fil = np.ones((len(idx), 3))
        for _ in range(settings.numReflection):
            if len(idx) == 0:
                break
            r = d - 2.0 * _dot(n, d)[:, None] * n
            fil = fil * cr
            rgb2, env, P2, N2, _k2 = _table_rays(sd, mats, lights, gl, p + r * SURFACE_DIST * 3.0, r, far, settings)
            out[idx, :3] += gl.ks * fil * rgb2
            out[idx, 3] += 1.0
            go = ~env
            idx, p, n, d, cr, fil = idx[go], P2[go], N2[go], r[go], cr[go], fil[go]
    if settings.enableRefraction:  # frag:2526-2570: two interfaces, air → medium → air, from the PRIMARY hit (oi)
        k0 = np.maximum(K, 0)
        c_refr = mats["cTransparent"][k0]
        idx = np.nonzero(~is_env & (np.sqrt(_dot(c_refr, c_refr)) != 0.0))[0]
        if len(idx):
            ior, ct = mats["ior"][k0][idx], c_refr[idx]
            rd_in = _refract(rd[idx], N[idx], 1.0 / ior)
            p_enter = P[idx] - N[idx] * SURFACE_DIST * 3.0
            d_in = _raymarch_depth(sd, p_enter, rd_in, far, settings.maxSteps, side=-1.0)  # INSIDE; UB2: the depth travelled on a miss
            p_exit = p_enter + rd_in * d_in[:, None]
            n_exit = -get_normal(sd, p_exit)
            rd_out = _refract(rd_in, n_exit, ior)
            ok = np.sqrt(_dot(rd_out, rd_out)) != 0.0  # otherwise total internal reflection: refr = vec4(0)
            if ok.any():
                rgb3, _e, _p, _n, _k = _table_rays(sd, mats, lights, gl, (p_exit - n_exit * SURFACE_DIST * 5.0)[ok], rd_out[ok], far, settings)
                out[idx[ok], :3] += gl.kt * ct[ok] * rgb3
                out[idx[ok], 3] += 1.0
    return out.reshape(H, W, 4), (~is_env).reshape(H, W)
