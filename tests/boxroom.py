"""Synthetic deep-BVH stress scene (the stand-in for BASELINE config 4, whose
Sponza asset does not ship with the reference): a closed box room of tessellated quads, noisy
icospheres in matte and plastic, one spherical area light — only features of the hot path.
Deterministic: geometry from a seeded numpy generator, floats written with %.9g so that strtof
reads back exactly the float32 values generated here."""
import numpy as np


def _icosphere(levels):
    t = (1.0 + 5 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
         (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10),
         (8, 6, 7), (9, 8, 1)]
    v = [np.array(p, np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(levels):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.array(v, np.float64), np.array(f, np.int32)


def _fmt(a):
    return " ".join("%.9g" % x for x in np.asarray(a, np.float32).ravel())


def _mesh(P, F, uv=None):
    return 'Shape "trianglemesh" "point P" [ %s ]\n  "integer indices" [ %s ]\n%s' % (
        _fmt(P), " ".join(str(int(i)) for i in F.ravel()), "" if uv is None else '  "float uv" [ %s ]\n' % _fmt(uv))


def write_test_images(directory, seed=5):
    """Image files for the textured room, one per reader: a non-power-of-two PFM (Lanczos resampling), an
    8-bit PNG (inverse gamma) and a run-length TGA. Returns {name: path}."""
    import os
    import struct
    import zlib
    rng = np.random.default_rng(seed)
    os.makedirs(directory, exist_ok=True)
    paths = {}
    # PFM 24 x 20: soft checker with a colour ramp
    h, w = 20, 24
    y, x = np.mgrid[0:h, 0:w]
    chk = ((x // 4 + y // 5) % 2).astype(np.float32)
    img = np.stack([0.15 + 0.7 * chk, 0.2 + 0.6 * (x / (w - 1)), 0.25 + 0.5 * (y / (h - 1))], -1).astype(np.float32)
    paths["checker"] = os.path.join(directory, "checker.pfm")
    open(paths["checker"], "wb").write(b"PF\n%d %d\n-1.0\n" % (w, h) + img[::-1].tobytes())
    # PNG 32 x 32 RGB noise blobs, filter type "up" on every row
    n = 32
    base = rng.integers(40, 230, (n // 4, n // 4, 3))
    px = np.repeat(np.repeat(base, 4, 0), 4, 1).astype(np.uint8)
    rows = px.reshape(n, n * 3).astype(np.int32)
    raw = b""
    prev = np.zeros(n * 3, np.int32)
    for r in rows:
        raw += b"\x02" + ((r - prev) & 255).astype(np.uint8).tobytes()
        prev = r

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data))

    paths["noise"] = os.path.join(directory, "noise.png")
    open(paths["noise"], "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", n, n, 8, 2, 0, 0, 0)) +
                                     chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))
    # TGA 16 x 8, 24 bit, run-length encoded, bottom-up: vertical stripes
    w, h = 16, 8
    stripes = np.zeros((h, w, 3), np.uint8)
    for i in range(w):
        stripes[:, i] = (250, 240, 230) if (i // 2) % 2 == 0 else (60, 90, 140)
    stripes[h // 2:, :, 1] //= 2
    body = b""
    for row in stripes[::-1, :, ::-1].reshape(h, w, 3):
        i = 0
        while i < w:
            j = i
            while j + 1 < w and (row[j + 1] == row[i]).all():
                j += 1
            body += bytes([0x80 | (j - i)]) + row[i].tobytes()
            i = j + 1
    paths["stripes"] = os.path.join(directory, "stripes.tga")
    open(paths["stripes"], "wb").write(struct.pack("<BBBHHBHHHHBB", 0, 0, 10, 0, 0, 0, 0, 0, w, h, 24, 0) + body)
    # single-channel PFM 8 x 8 alpha mask: zeros (holes), ones, and a few fractional texels
    m = np.ones((8, 8), np.float32)
    m[2:6, 1:4] = 0
    m[0:2, 5:8] = 0
    m[6, 6] = 0.25
    # single-channel PFM 32 x 32 height field for bump maps: smooth bumps
    yy, xx = np.mgrid[0:32, 0:32]
    bumps = (0.04 * (np.sin(xx * np.pi / 4) * np.cos(yy * np.pi / 8) + 1)).astype(np.float32)
    paths["bumps"] = os.path.join(directory, "bumps.pfm")
    open(paths["bumps"], "wb").write(b"Pf\n32 32\n-1.0\n" + bumps.tobytes())
    paths["mask"] = os.path.join(directory, "mask.pfm")
    open(paths["mask"], "wb").write(b"Pf\n8 8\n-1.0\n" + m.tobytes())
    return paths


def _grid_uv(n, repeat):
    return np.array([(repeat * i / n, repeat * j / n) for j in range(n + 1) for i in range(n + 1)], np.float64)


def _grid_quad(origin, du, dv, n):
    """n x n tessellated parallelogram (2 n^2 triangles)."""
    o, du, dv = (np.array(x, np.float64) for x in (origin, du, dv))
    P = np.array([o + du * (i / n) + dv * (j / n) for j in range(n + 1) for i in range(n + 1)])
    F = []
    for j in range(n):
        for i in range(n):
            a = j * (n + 1) + i
            F += [(a, a + 1, a + n + 2), (a, a + n + 2, a + n + 1)]
    return P, np.array(F, np.int32)


def boxroom_pbrt(xres=64, yres=64, spp=4, ico_levels=4, n_blobs=6, wall_n=24, seed=12111, maxdepth=5, light="area", materials="plain",
                 textures=None, env_dir=None):
    """Returns the scene text. Triangles: 5 * 2 * wall_n^2 + n_blobs * 20 * 4^ico_levels
    (defaults: 5 760 + 30 720; ico_levels=5, n_blobs=12, wall_n=64 gives ~287 k).
    `textures`: a directory — image files are written there (write_test_images) and the walls (uv-mapped,
    repeated) and the matte / plastic blobs (per-triangle default uv) take their Kd / Ks from image textures."""
    rng = np.random.default_rng(seed)
    out = ['LookAt 0 -9.5 1   0 0 0.5   0 0 1', 'Camera "perspective" "float fov" [55]',
           'Film "image" "integer xresolution" [%d] "integer yresolution" [%d]' % (xres, yres),
           'Sampler "halton" "integer pixelsamples" [%d]' % spp, 'Integrator "path" "integer maxdepth" [%d]' % maxdepth,
           'WorldBegin']
    if light == "envmap":  # an environment-mapped sky (lat-long PFM, 12 x 6: resampled to 16 x 8) through the open top
        import os
        # the map goes into `env_dir` (an environment map over untextured materials), else into the `textures` directory
        env_dir = env_dir if env_dir is not None else textures
        assert env_dir is not None, "light='envmap' writes its map into the `env_dir` / `textures` directory"
        os.makedirs(env_dir, exist_ok=True)
        h, w = 6, 12
        y, x = np.mgrid[0:h, 0:w]
        sky = np.stack([0.3 + 0.05 * y, 0.4 + 0.04 * y, 0.9 - 0.1 * y], -1).astype(np.float32)
        sky[1, 3] = (60, 50, 30)   # the sun
        sky[4:, :] *= 0.2          # dim ground
        open(os.path.join(env_dir, "sky.pfm"), "wb").write(b"PF\n%d %d\n-1.0\n" % (w, h) + sky[::-1].tobytes())
        out.append('AttributeBegin\n  Rotate 25 0 1 0\n  Rotate 40 0 0 1\n  LightSource "infinite" "color L" [1 .9 .8] "color scale" [1.5 1.5 1.5] '
                   '"string mapname" ["%s"]\nAttributeEnd' % os.path.join(env_dir, "sky.pfm"))
    elif light == "sky":  # uniform sky through the open top (no ceiling below) and a warm point light inside
        out.append('AttributeBegin\n  Rotate 25 0 1 0\n  LightSource "infinite" "color L" [.6 .7 1] "color scale" [1.5 1.5 1.5]\nAttributeEnd')
        out.append('LightSource "point" "color I" [25 15 8] "point from" [-6 5 2]')
    elif light == "quad":  # a rectangular ceiling panel: two triangle emitters (one light each) + a point light
        out.append('AttributeBegin\n  Material "matte" "color Kd" [.1 .1 .1]\n  AreaLightSource "diffuse" "color L" [12 12 11]\n'
                   '  Shape "trianglemesh" "point P" [ -2 -3 8.5   3 -3 8.5   3 1 8.5   -2 1 8.5 ]\n'
                   '    "integer indices" [ 0 2 1   0 3 2 ]\nAttributeEnd')
        out.append('LightSource "point" "color I" [15 10 5] "point from" [-6 5 2]')
    elif light == "multi":  # three lights of three kinds: the path integrator's spatial light distribution
        out.append('AttributeBegin\n  Material "matte" "color Kd" [0 0 0]\n  Translate 1.5 -2 7.5\n'
                   '  AreaLightSource "area" "color L" [40 40 40]\n  Shape "sphere" "float radius" [0.6]\nAttributeEnd')
        out.append('LightSource "point" "color I" [30 20 10] "point from" [-6 5 2]')
        out.append('LightSource "spot" "color I" [200 200 260] "point from" [7 -6 6] "point to" [0 2 -3] '
                   '"float coneangle" [35] "float conedeltaangle" [10]')
    elif light == "many":  # eight point lights (kMaxLights): UniformSampleAllLights makes eight NEE records per hit
        for i in range(8):
            out.append('LightSource "point" "color I" [%g %g %g] "point from" [%g %g %g]' % (
                8 + i, 9, 10 - i, -6 + 1.7 * i, 5 - 1.2 * i, 1.5 + 0.7 * (i % 4)))
    elif light == "spot":  # a spot light from the emitter's position down into the room
        out.append('AttributeBegin\n  Translate 1.5 -2 0\n  LightSource "spot" "color I" [300 300 300] "point from" [0 0 7.5] '
                   '"point to" [-1.5 3 -3] "float coneangle" [40] "float conedeltaangle" [12]\nAttributeEnd')
    elif light == "distant":  # sun through the open top (the ceiling is left out below)
        out.append('AttributeBegin\n  Rotate 20 1 0 0\n  LightSource "distant" "color L" [3 3 2.7] "point from" [0.2 -0.3 1] '
                   '"point to" [0 0 0]\nAttributeEnd')
    elif light == "point":  # a delta light instead of the emitting sphere (same position, similar power)
        out.append('AttributeBegin\n  Translate 1.5 -2 0\n  LightSource "point" "color I" [68 68 68] "point from" [0 0 7.5]\n'
                   'AttributeEnd')
    else:
        out.append('AttributeBegin\n  Material "matte" "color Kd" [0 0 0]\n  Translate 1.5 -2 7.5\n'
                   '  AreaLightSource "area" "color L" [60 60 60]\n  Shape "sphere" "float radius" [0.6]\nAttributeEnd')
    tex = None
    if textures is not None:
        tex = write_test_images(textures)
        out.append('Texture "checker" "spectrum" "imagemap" "string filename" ["%s"]' % tex["checker"])
        out.append('Texture "checker-tri" "spectrum" "imagemap" "string filename" ["%s"] "bool trilinear" ["true"] '
                   '"float uscale" [2] "float vscale" [3] "float udelta" [.25]' % tex["checker"])
        out.append('Texture "noise" "spectrum" "imagemap" "string filename" ["%s"] "string wrap" ["clamp"] "float scale" [.9]' % tex["noise"])
        out.append('Texture "noise-black" "spectrum" "imagemap" "string filename" ["%s"] "string wrap" ["black"] '
                   '"float uscale" [1.5] "float vscale" [1.5] "float maxanisotropy" [2]' % tex["noise"])
        out.append('Texture "stripes" "spectrum" "imagemap" "string filename" ["%s"] "bool gamma" ["false"]' % tex["stripes"])
        out.append('Texture "noise-tint" "spectrum" "scale" "texture tex1" ["noise"] "color tex2" [.9 .6 .3]')
        out.append('Texture "sigma" "float" "imagemap" "string filename" ["%s"] "float scale" [700] "bool gamma" ["false"]' % tex["bumps"])
        out.append('Texture "rough" "float" "imagemap" "string filename" ["%s"] "float scale" [4] "bool gamma" ["false"]' % tex["bumps"])
        out.append('Texture "bumps" "float" "imagemap" "string filename" ["%s"] "float uscale" [4] "float vscale" [4]' % tex["bumps"])
        # alpha masks (triangle.cpp:325-331, 509-541): a free-standing screen full of holes whose shadow has
        # further holes (shadowalpha), and an invisible box ("float alpha" [0]) around a blob
        out.append('Texture "mask" "float" "imagemap" "string filename" ["%s"] "float uscale" [3] "float vscale" [2]' % tex["mask"])
        out.append('Texture "mask-coarse" "float" "imagemap" "string filename" ["%s"] "bool trilinear" ["true"]' % tex["mask"])
        P, F = _grid_quad((-5, -4, -3), (6, 1.5, 0), (0, 0, 7), 6)
        out.append('AttributeBegin\n  Material "matte" "texture Kd" ["noise"]\n%s  "texture alpha" ["mask"] "texture shadowalpha" ["mask-coarse"]\nAttributeEnd'
                   % _mesh(P, F, _grid_uv(6, 1.0)))
        P, F = _grid_quad((1, -6, -2.5), (4, 0, 0), (0, 0, 4), 2)
        out.append('AttributeBegin\n  Material "mirror"\n%s  "float alpha" [0]\nAttributeEnd' % _mesh(P, F))
        P, F = _grid_quad((-8, 2, 4), (5, 0, 0), (0, 3, 0), 3)
        out.append('AttributeBegin\n  Material "matte" "color Kd" [.9 .2 .2]\n%s  "float shadowalpha" [0]\nAttributeEnd' % _mesh(P, F))
    wall_tex = ["checker", "noise-tint", "checker-tri", "stripes", "noise-black"]
    s = 10.0
    walls = [((-s, -s, -3), (2 * s, 0, 0), (0, 2 * s, 0), (.7, .7, .7)),      # floor
             ((-s, -s, 9), (0, 2 * s, 0), (2 * s, 0, 0), (.8, .8, .8)),       # ceiling
             ((-s, s, -3), (2 * s, 0, 0), (0, 0, 12), (.6, .6, .8)),          # back
             ((-s, -s, -3), (0, 2 * s, 0), (0, 0, 12), (.8, .3, .3)),         # left
             ((s, -s, -3), (0, 0, 12), (0, 2 * s, 0), (.3, .8, .3))]          # right
    for o, du, dv, kd in walls:
        if light in ("distant", "sky", "envmap") and o[2] == 9:
            continue  # no ceiling
        P, F = _grid_quad(o, du, dv, wall_n)
        rough = ' "float sigma" [%g]' % (20 + 10 * len(out) % 50) if materials == "all" else ""  # Oren-Nayar walls
        if tex is not None:
            name = wall_tex[walls.index((o, du, dv, kd))]
            bump = ' "texture bumpmap" ["bumps"]' if name in ("checker", "stripes") else ""
            out.append('AttributeBegin\n  Material "matte" "texture Kd" ["%s"]%s%s\n%sAttributeEnd' % (
                name, rough, bump, _mesh(P, F, _grid_uv(wall_n, 3.0))))
            continue
        out.append('AttributeBegin\n  Material "matte" "color Kd" [%g %g %g]%s\n%sAttributeEnd' % (*kd, rough, _mesh(P, F)))
    V, F = _icosphere(ico_levels)
    for b in range(n_blobs):
        r = rng.uniform(0.7, 1.6)
        c = np.array([rng.uniform(-6, 6), rng.uniform(-4, 7), rng.uniform(-2, 5)])
        noise = 1.0 + 0.18 * rng.standard_normal(len(V)).clip(-2.5, 2.5)
        P = c + V * (r * noise)[:, None]
        if materials == "roughglass" and b % 3 != 2:  # GlassMaterial with uroughness = vroughness != 0 (glass.cpp:66-90): closed refractive blobs
            rough = (.05, .2, .5)[b % 3] if b % 2 else rng.uniform(.02, .6)
            mat = 'Material "glass" "color Kr" [%g %g %g] "color Kt" [%g %g %g] "float uroughness" [%g] "float vroughness" [%g] "float index" [%g]%s' % (
                *((1, 1, 1) if b % 4 else (0, 0, 0)), *rng.uniform(.7, 1, 3), rough, rough, rng.uniform(1.3, 1.7), ' "bool remaproughness" ["false"]' if b % 5 == 0 else "")
        elif materials == "aniso" and b % 3 == 0:  # anisotropic Trowbridge-Reitz (uroughness != vroughness) on rough glass, glass.cpp:52-73 ...
            ur, vr = ((0., .3), (.4, .05), (.08, .5))[(b // 3) % 3] if b % 2 else tuple(rng.uniform(.02, .6, 2))
            mat = 'Material "glass" "color Kr" [%g %g %g] "color Kt" [%g %g %g] "float uroughness" [%g] "float vroughness" [%g] "float index" [%g]%s' % (
                *((1, 1, 1) if b % 4 else (0, 0, 0)), *rng.uniform(.7, 1, 3), ur, vr, rng.uniform(1.3, 1.7), ' "bool remaproughness" ["false"]' if b % 5 == 0 and ur > 0 else "")
        elif materials == "aniso" and b % 3 == 1:  # ... and on uber's glossy lobe (uber.cpp:73-86), beside its other lobes
            mat = 'Material "uber" "color Kd" [%g %g %g] "color Ks" [%g %g %g] "color Kr" [.1 .1 .1] "float uroughness" [%g] "float vroughness" [%g] "float index" [%g]%s' % (
                *rng.uniform(.05, .4, 3), *rng.uniform(.3, .8, 3), rng.uniform(.02, .5), rng.uniform(.02, .5), rng.uniform(1.2, 1.8),
                ' "color opacity" [.7 .7 .7]' if b % 2 else "")
        elif materials == "aniso" and tex is not None and b % 3 == 2:  # ... with float images for uroughness / vroughness / roughness (uber.cpp:73-86)
            which = ('"texture uroughness" ["rough"] "float vroughness" [.1]',                                  # u from an image, v a number
                     '"texture roughness" ["rough"] "texture vroughness" ["sigma"] "bool remaproughness" ["false"]',  # u from "roughness", v from another image
                     '"texture roughness" ["rough"] "float uroughness" [.3] "texture vroughness" ["rough"]',      # "roughness" is not looked at
                     '"texture uroughness" ["rough"]')[(b // 3) % 4]                                             # v follows u
            mat = 'Material "uber" "color Kd" [%g %g %g] "color Ks" [.5 .5 .5] %s "float index" [%g]' % (*rng.uniform(.1, .4, 3), which, rng.uniform(1.2, 1.8))
        elif materials == "ubertrans" and b % 4 == 0:  # UberMaterial's pass-through (uber.cpp:53-61): a grey and a coloured opacity
            op = (.35, .35, .35) if b % 8 == 0 else tuple(rng.uniform(.1, .9, 3))
            mat = 'Material "uber" "color Kd" [%g %g %g] "color Ks" [.2 .2 .2] "float roughness" [%g] "color opacity" [%g %g %g] "float index" [%g]' % (
                *rng.uniform(.2, .7, 3), rng.uniform(.05, .3), *op, rng.uniform(1.2, 1.8))
        elif materials == "ubertrans" and b % 4 == 1:  # ... its Kt lobe (uber.cpp:94-99) beside the other four: closed refractive blobs
            mat = 'Material "uber" "color Kd" [%g %g %g] "color Ks" [.1 .1 .1] "color Kr" [.2 .2 .2] "color Kt" [%g %g %g] "float roughness" [.1] "float index" [%g]' % (
                *rng.uniform(.05, .3, 3), *rng.uniform(.5, .9, 3), rng.uniform(1.2, 1.7))
        elif materials == "ubertrans" and b % 4 == 2:  # ... both, and nothing else: every lobe specular, no light sampling at these vertices
            mat = 'Material "uber" "color Kd" [0 0 0] "color Ks" [0 0 0] "color Kt" [%g %g %g] "color opacity" [.6 .6 .6] "float index" [%g]' % (
                *rng.uniform(.6, 1, 3), rng.uniform(1.3, 1.6))
        elif materials == "ubertrans" and tex is not None and b % 4 == 3:  # "opacity" (and Kt) from image textures: the pass-through varies over the surface
            mat = 'Material "uber" "color Kd" [%g %g %g] "color Ks" [.2 .2 .2] "texture opacity" ["%s"] "texture Kt" ["stripes"] "float index" [%g]' % (
                *rng.uniform(.2, .7, 3), ("noise", "noise-tint", "checker")[(b // 4) % 3], rng.uniform(1.2, 1.6))
        elif materials == "mixed" and b % 3 == 1:  # specular lobes: uber (diffuse + glossy + mirror-like) and mirror
            mat = 'Material "uber" "color Kd" [%g %g %g] "color Ks" [.2 .2 .2] "color Kr" [.3 .3 .3] "float roughness" [%g] "float index" [%g]' % (
                *rng.uniform(.1, .4, 3), rng.uniform(.05, .3), rng.uniform(1.2, 1.8))
        elif materials == "mixed" and b % 3 == 2:
            mat = 'Material "mirror" "color Kr" [%g %g %g]' % tuple(rng.uniform(.6, .95, 3))
        elif materials == "all" and b % 5 == 3:
            mat = 'Material "glass" "color Kr" [1 1 1] "color Kt" [%g %g %g] "float index" [%g]' % (*rng.uniform(.7, 1, 3), rng.uniform(1.3, 1.7))
        elif materials == "all" and b % 5 == 1:
            mat = 'Material "uber" "color Kd" [%g %g %g] "color Ks" [.2 .2 .2] "color Kr" [.3 .3 .3] "float roughness" [%g] "float index" [%g]' % (
                *rng.uniform(.1, .4, 3), rng.uniform(.05, .3), rng.uniform(1.2, 1.8))
        elif materials == "all" and b % 5 == 2:
            mat = 'Material "mirror" "color Kr" [%g %g %g]' % tuple(rng.uniform(.6, .95, 3))
        elif materials == "glass" and b % 2 == 1:  # closed refractive blobs, one of them tinted
            mat = 'Material "glass" "color Kr" [1 1 1] "color Kt" [%g %g %g] "float index" [%g]' % (
                *((1, 1, 1) if b % 4 == 1 else rng.uniform(.7, 1, 3)), rng.uniform(1.3, 1.7))
        elif tex is not None and b % 4 == 0:  # blobs have no uv: every triangle maps the unit half-square
            mat = 'Material "plastic" "texture Kd" ["noise"] "texture Ks" ["stripes"] "texture roughness" ["rough"]'
            rng.uniform(.02, .3)
        elif tex is not None and b % 4 == 2:  # smooth-shaded (vertex normals: dn/du, dn/dv enter Material::Bump) and bumpy
            mat = 'Material "plastic" "color Kd" [.5 .4 .2] "color Ks" [.3 .3 .3] "float roughness" [.1] "texture bumpmap" ["bumps"]'
            nrm = P - c
            nrm /= np.linalg.norm(nrm, axis=1)[:, None]
            out.append('AttributeBegin\n  %s\n%s  "normal N" [ %s ]\nAttributeEnd' % (mat, _mesh(P, F), _fmt(nrm)))
            continue
        elif tex is not None and b % 4 == 1:
            mat = 'Material "matte" "texture Kd" ["checker"] "texture sigma" ["sigma"]'
        elif b % 2 == 0:
            mat = 'Material "plastic" "color Kd" [%g %g %g] "color Ks" [.4 .4 .4] "float roughness" [%g]' % (
                *rng.uniform(.2, .7, 3), rng.uniform(.02, .3))
        else:
            mat = 'Material "matte" "color Kd" [%g %g %g]' % tuple(rng.uniform(.2, .8, 3))
        out.append('AttributeBegin\n  %s\n%sAttributeEnd' % (mat, _mesh(P, F)))
    out.append('WorldEnd')
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    import sys
    sys.stdout.write(boxroom_pbrt())
