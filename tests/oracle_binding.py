"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg — never by the product package.
"""
import ctypes
import os

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "oracle", "_build", "liboracle.so")

TRIG_LIBM, TRIG_PORTABLE = 0, 1

c_vp = ctypes.c_void_p


class OracleStats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in ("camera_rays regular_rays shadow_rays tri_tests tri_hits sphere_tests "
                                               "nodes_closest nodes_any nee_evals zero_radiance").split()] + [
        ("path_length", ctypes.c_uint64 * 8), ("max_stack_depth", ctypes.c_int32), ("threads", ctypes.c_int32),
        ("seconds", ctypes.c_double)]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_ if n != "path_length"}
        d["path_length"] = list(self.path_length)
        return d


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class Oracle:
    def __init__(self):
        if not os.path.exists(LIB):
            raise RuntimeError(f"{LIB} missing: run __graft_entry__.build()")
        lib = ctypes.CDLL(LIB)
        lib.oracle_render.argtypes = [c_vp] + [ctypes.c_int] * 6 + [c_vp, ctypes.POINTER(OracleStats)]
        lib.oracle_halton_index.restype = ctypes.c_int64
        lib.oracle_bsdf_sample_batch.argtypes = [c_vp, ctypes.c_int, ctypes.c_int, c_vp, ctypes.c_int, c_vp, c_vp, c_vp]
        lib.oracle_bsdf_pdf_batch.argtypes = [c_vp, ctypes.c_int, ctypes.c_int, c_vp, ctypes.c_int, c_vp, c_vp]
        lib.oracle_render_probe.argtypes = [c_vp, ctypes.c_int, c_vp, c_vp, c_vp, c_vp, c_vp]
        lib.oracle_texture_eval.argtypes = [c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp, c_vp, c_vp]
        lib.oracle_camera_hit_differentials.argtypes = [c_vp, ctypes.c_int, ctypes.c_float, ctypes.c_float, c_vp]
        lib.oracle_distribution1d.argtypes = [c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.POINTER(ctypes.c_float),
                                              ctypes.POINTER(ctypes.c_float)]
        lib.oracle_log.argtypes = [ctypes.c_int, ctypes.c_float]
        lib.oracle_log.restype = ctypes.c_float
        lib.oracle_light_solid_angle.argtypes = [c_vp, ctypes.c_int, c_vp, ctypes.c_int, c_vp, c_vp]
        lib.oracle_sphere_solid_angle.argtypes = [c_vp, ctypes.c_int, c_vp, ctypes.c_int, c_vp, c_vp]
        lib.oracle_check_next_float.restype = ctypes.c_int64
        lib.oracle_check_next_float.argtypes = [ctypes.c_int, ctypes.c_uint64]
        lib.oracle_check_efloat.restype = ctypes.c_int64
        lib.oracle_check_efloat.argtypes = [ctypes.c_int, ctypes.c_uint64]
        lib.oracle_check_reintersect.restype = ctypes.c_int64
        lib.oracle_check_reintersect.argtypes = [c_vp, ctypes.c_int, c_vp, c_vp, ctypes.c_int, ctypes.c_uint64, c_vp]
        lib.oracle_halton_index.argtypes = [c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int64]
        lib.oracle_halton_sample.restype = ctypes.c_float
        lib.oracle_halton_sample.argtypes = [c_vp, ctypes.c_int64, ctypes.c_int]
        lib.oracle_sample_index.restype = ctypes.c_int64
        lib.oracle_sample_index.argtypes = [c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int64]
        lib.oracle_sample_dimension.restype = ctypes.c_float
        lib.oracle_sample_dimension.argtypes = [c_vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        lib.oracle_reverse_bits32.restype = ctypes.c_uint32
        lib.oracle_reverse_bits32.argtypes = [ctypes.c_uint32]
        lib.oracle_multiply_generator.restype = ctypes.c_uint32
        lib.oracle_multiply_generator.argtypes = [c_vp, ctypes.c_uint32]
        lib.oracle_sample_generator_matrix.restype = ctypes.c_float
        lib.oracle_sample_generator_matrix.argtypes = [c_vp, ctypes.c_uint32, ctypes.c_uint32]
        lib.oracle_gray_code_sample.argtypes = [c_vp, ctypes.c_uint32, ctypes.c_uint32, c_vp]
        lib.oracle_sobol_sample_float.restype = ctypes.c_float
        lib.oracle_sobol_sample_float.argtypes = [c_vp, ctypes.c_int64, ctypes.c_int, ctypes.c_uint32]
        lib.oracle_sobol_sample_double.restype = ctypes.c_double
        lib.oracle_sobol_sample_double.argtypes = [c_vp, ctypes.c_int64, ctypes.c_int, ctypes.c_uint64]
        lib.oracle_sobol_interval_to_index.restype = ctypes.c_uint64
        lib.oracle_sobol_interval_to_index.argtypes = [c_vp, c_vp, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int, ctypes.c_int]
        lib.oracle_radical_inverse.restype = ctypes.c_float
        lib.oracle_radical_inverse.argtypes = [ctypes.c_int, ctypes.c_uint64]
        lib.oracle_scrambled_radical_inverse.restype = ctypes.c_float
        lib.oracle_scrambled_radical_inverse.argtypes = [c_vp, ctypes.c_int, ctypes.c_uint64]
        lib.oracle_scrambled_radical_inverse_perm.restype = ctypes.c_float
        lib.oracle_scrambled_radical_inverse_perm.argtypes = [ctypes.c_int, c_vp, ctypes.c_uint64]
        lib.oracle_camera_ray.argtypes = [c_vp] + [ctypes.c_float] * 4 + [c_vp, c_vp]
        lib.oracle_intersect.argtypes = [c_vp, ctypes.c_int, c_vp, c_vp, c_vp, c_vp, c_vp]
        lib.oracle_intersect_p.argtypes = [c_vp, ctypes.c_int, c_vp, c_vp, c_vp, c_vp]
        lib.oracle_li.argtypes = [c_vp, ctypes.c_int, ctypes.c_int, c_vp, c_vp, c_vp, c_vp, c_vp]
        lib.oracle_bsdf_eval.argtypes = [c_vp, ctypes.c_int, ctypes.c_int, c_vp, c_vp, c_vp, c_vp]
        lib.oracle_bsdf_sample.argtypes = [c_vp, ctypes.c_int, ctypes.c_int, c_vp, c_vp, c_vp, c_vp, c_vp]
        lib.oracle_sincos.argtypes = [ctypes.c_int, ctypes.c_float, c_vp, c_vp]
        lib.oracle_sincos_d.argtypes = [ctypes.c_int, ctypes.c_double, c_vp, c_vp]
        lib.oracle_acos.restype = ctypes.c_float
        lib.oracle_acos.argtypes = [ctypes.c_int, ctypes.c_float]
        self.lib = lib

    def render(self, scene, trig_mode=TRIG_PORTABLE, threads=0, k_begin=0, k_end=-1, tile_rank=0, tile_nranks=1):
        h, w = scene.film_shape
        film = np.zeros((h, w, 4), np.float32)
        st = OracleStats()
        rc = self.lib.oracle_render(scene.desc, trig_mode, threads, k_begin, k_end, tile_rank, tile_nranks,
                                    film.ctypes.data, ctypes.byref(st))
        assert rc == 0
        return film, st.as_dict()

    def iispt_hemi_points(self, scene, task, trig_mode=TRIG_PORTABLE):
        nx, ny = task.grid()
        valid = np.zeros((ny, nx), np.uint8)
        pos = np.zeros((ny, nx, 3), np.float32)
        dr = np.zeros((ny, nx, 3), np.float32)
        self.lib.oracle_iispt_hemi_points.argtypes = [c_vp, ctypes.c_int, c_vp, c_vp, c_vp, c_vp]
        n = self.lib.oracle_iispt_hemi_points(scene.desc, trig_mode, ctypes.byref(task), valid.ctypes.data, pos.ctypes.data, dr.ctypes.data)
        assert n == nx * ny
        return valid, pos, dr

    def iispt_gather(self, scene, task, valid, pos, direction, nn_films, trig_mode=TRIG_PORTABLE):
        valid, pos, direction, nn = np.ascontiguousarray(valid, np.uint8), _f32(pos), _f32(direction), _f32(nn_films)
        out = np.zeros((task.y1 - task.y0, task.x1 - task.x0, 4), np.float32)
        self.lib.oracle_iispt_gather.argtypes = [c_vp, ctypes.c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]
        rc = self.lib.oracle_iispt_gather(scene.desc, trig_mode, ctypes.byref(task), valid.ctypes.data, pos.ctypes.data, direction.ctypes.data,
                                          nn.ctypes.data, out.ctypes.data)
        assert rc == 0
        return out

    def bvh_hlbvh(self, bounds6, max_prims_in_node=4):
        """oracle_bvh_hlbvh (oracle/oracle_bvh.cpp): (nodes as the binding's BVH_NODE array, order (n,) int32, sorted Morton codes)."""
        import importlib.util
        import sys
        b = sys.modules.get("iile_binding")
        dt = b.BVH_NODE if b is not None else np.dtype([("bmin", "<f4", 3), ("bmax", "<f4", 3), ("offset", "<i4"), ("nprims", "<u2"), ("axis", "u1"), ("pad", "u1")])
        b6 = _f32(bounds6).reshape(-1, 6)
        n = len(b6)
        nodes = np.zeros(max(2 * n, 1), dtype=dt)
        order = np.zeros(max(n, 1), dtype=np.int32)
        codes = np.zeros(max(n, 1), dtype=np.uint32)
        n_nodes = ctypes.c_int32(0)
        f = self.lib.oracle_bvh_hlbvh
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32), ctypes.c_void_p, ctypes.c_void_p]
        rc = f(n, b6.ctypes.data, int(max_prims_in_node), nodes.ctypes.data, ctypes.byref(n_nodes), order.ctypes.data, codes.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"oracle_bvh_hlbvh: {rc} (1: a CHECK of the reference would abort)")
        return nodes[:n_nodes.value].copy(), order[:n].copy(), codes[:n].copy()

    def iispt_direct(self, scene, n_passes, first_pass=0, threads=0, trig_mode=TRIG_PORTABLE):
        """oracle_iispt_direct: the direct film monitor {sum r, g, b, weight} as float64 (h, w, 4)."""
        h, w = scene.film_shape
        film = np.zeros((h, w, 4), np.float64)
        f = self.lib.oracle_iispt_direct
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        rc = f(scene.desc, trig_mode, int(n_passes), int(first_pass), int(threads), film.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"oracle_iispt_direct: {rc}")
        return film

    def iispt_merge(self, direct, indirect):
        """oracle_iispt_merge: IisptFilmMonitor::merge_into + to_intensity_film of two (h, w, 4) float64 monitors -> (h, w, 3) float32."""
        d = np.ascontiguousarray(direct, np.float64)
        i = np.ascontiguousarray(indirect, np.float64)
        out = np.zeros(d.shape[:2] + (3,), np.float32)
        f = self.lib.oracle_iispt_merge
        f.restype = None
        f.argtypes = [ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        f(d.shape[0] * d.shape[1], d.ctypes.data, i.ctypes.data, out.ctypes.data)
        return out

    def tile_owner(self, tx, ty, nranks):
        return int(self.lib.oracle_tile_owner(int(tx), int(ty), int(nranks)))

    def sample_index(self, scene, px, py, k):
        return int(self.lib.oracle_sample_index(scene.desc, int(px), int(py), int(k)))

    def sample_dimension(self, scene, index, dim, px, py):
        return np.float32(self.lib.oracle_sample_dimension(scene.desc, int(index), int(dim), int(px), int(py)))

    def halton_index(self, scene, px, py, k):
        return int(self.lib.oracle_halton_index(scene.desc, int(px), int(py), int(k)))

    def halton_sample(self, scene, index, dim):
        return np.float32(self.lib.oracle_halton_sample(scene.desc, int(index), int(dim)))

    def radical_inverse(self, base_index, a):
        return np.float32(self.lib.oracle_radical_inverse(int(base_index), int(a)))

    def scrambled_radical_inverse(self, scene, base_index, a):
        return np.float32(self.lib.oracle_scrambled_radical_inverse(scene.desc, int(base_index), int(a)))

    def scrambled_radical_inverse_perm(self, base, perm, a):
        perm = np.ascontiguousarray(perm, np.uint16)
        return np.float32(self.lib.oracle_scrambled_radical_inverse_perm(int(base), perm.ctypes.data, int(a)))

    def camera_rays(self, scene, pfilm, plens=None):
        pfilm = _f32(pfilm)
        n = len(pfilm)
        o = np.empty((n, 3), np.float32)
        d = np.empty((n, 3), np.float32)
        for i in range(n):
            lx, ly = (plens[i] if plens is not None else (0.0, 0.0))
            self.lib.oracle_camera_ray(scene.desc, float(pfilm[i, 0]), float(pfilm[i, 1]), float(lx), float(ly),
                                       o[i].ctypes.data, d[i].ctypes.data)
        return o, d

    def intersect(self, scene, o, d, tmax):
        o, d, tmax = _f32(o), _f32(d), _f32(tmax)
        n = len(tmax)
        prim = np.empty(n, np.int32)
        tb = np.empty((n, 4), np.float32)
        self.lib.oracle_intersect(scene.desc, n, o.ctypes.data, d.ctypes.data, tmax.ctypes.data, prim.ctypes.data,
                                  tb.ctypes.data)
        return prim, tb

    def light_solid_angle(self, scene, light, p, n_samples):
        p = _f32(p)
        a, b = ctypes.c_double(), ctypes.c_double()
        self.lib.oracle_light_solid_angle(scene.desc, light, p.ctypes.data, n_samples, ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    def sphere_solid_angle(self, scene, sphere, p, n_samples):
        p = _f32(p)
        a, b = ctypes.c_double(), ctypes.c_double()
        self.lib.oracle_sphere_solid_angle(scene.desc, sphere, p.ctypes.data, n_samples, ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    def check_next_float(self, iters=100000, seed=1):
        return int(self.lib.oracle_check_next_float(iters, seed))

    def check_efloat(self, iters=200000, seed=1):
        return int(self.lib.oracle_check_efloat(iters, seed))

    def check_reintersect(self, scene, o, d, n_out=200, seed=1):
        o, d = _f32(o), _f32(d)
        stats = np.zeros(2, np.int64)
        bad = int(self.lib.oracle_check_reintersect(scene.desc, len(o), o.ctypes.data, d.ctypes.data, n_out, seed,
                                                    stats.ctypes.data))
        return bad, int(stats[0]), int(stats[1])

    def intersect_p(self, scene, o, d, tmax):
        o, d, tmax = _f32(o), _f32(d), _f32(tmax)
        n = len(tmax)
        hit = np.empty(n, np.int32)
        self.lib.oracle_intersect_p(scene.desc, n, o.ctypes.data, d.ctypes.data, tmax.ctypes.data, hit.ctypes.data)
        return hit

    def li(self, scene, px, py, k, trig_mode=TRIG_PORTABLE):
        px, py, k = _i32(px), _i32(py), _i32(k)
        n = len(px)
        L = np.empty((n, 3), np.float32)
        nr = np.empty((n, 2), np.int32)
        self.lib.oracle_li(scene.desc, trig_mode, n, px.ctypes.data, py.ctypes.data, k.ctypes.data, L.ctypes.data,
                           nr.ctypes.data)
        return L, nr

    def bsdf_eval(self, scene, mat, wo, wi, trig_mode=TRIG_PORTABLE):
        wo, wi = _f32(wo), _f32(wi)
        out = np.empty((len(wo), 4), np.float32)
        for i in range(len(wo)):
            self.lib.oracle_bsdf_eval(scene.desc, trig_mode, mat, wo[i].ctypes.data, wi[i].ctypes.data,
                                      out[i].ctypes.data, out[i, 3:].ctypes.data)
        return out

    def bsdf_sample(self, scene, mat, wo, u, trig_mode=TRIG_PORTABLE):
        wo, u = _f32(wo), _f32(u)
        out = np.zeros((len(wo), 7), np.float32)
        for i in range(len(wo)):
            self.lib.oracle_bsdf_sample(scene.desc, trig_mode, mat, wo[i].ctypes.data, u[i].ctypes.data,
                                        out[i].ctypes.data, out[i, 3:].ctypes.data, out[i, 6:].ctypes.data)
        return out

    def bsdf_sample_batch(self, scene, mat, wo, u, trig_mode=TRIG_LIBM):
        wo, u = _f32(wo), _f32(u)
        n = len(u)
        wi, pdf = np.empty((n, 3), np.float32), np.empty(n, np.float32)
        self.lib.oracle_bsdf_sample_batch(scene.desc, trig_mode, mat, wo.ctypes.data, n, u.ctypes.data, wi.ctypes.data,
                                          pdf.ctypes.data)
        return wi, pdf

    def bsdf_pdf_batch(self, scene, mat, wo, wi, trig_mode=TRIG_LIBM):
        wo, wi = _f32(wo), _f32(wi)
        pdf = np.empty(len(wi), np.float32)
        self.lib.oracle_bsdf_pdf_batch(scene.desc, trig_mode, mat, wo.ctypes.data, len(wi), wi.ctypes.data,
                                       pdf.ctypes.data)
        return pdf

    def render_probe(self, scene, pos, direction, hemi=32, trig_mode=TRIG_PORTABLE):
        """(intensity (h, h, 3), normals (h, h, 3), distance (h, h)) of one IISPT probe, [y][x] raster order."""
        pos, direction = _f32(pos), _f32(direction)
        inten, nrm, dist = np.zeros((hemi, hemi, 3), np.float32), np.zeros((hemi, hemi, 3), np.float32), np.zeros((hemi, hemi), np.float32)
        rc = self.lib.oracle_render_probe(scene.desc, trig_mode, pos.ctypes.data, direction.ctypes.data, inten.ctypes.data,
                                          nrm.ctypes.data, dist.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"oracle_render_probe failed ({rc})")
        return inten, nrm, dist

    def texture_eval(self, scene, tex, uv, duv, trig_mode=TRIG_PORTABLE):
        uv, duv = _f32(uv), _f32(duv)
        out = np.empty((len(uv), 3), np.float32)
        self.lib.oracle_texture_eval(scene.desc, trig_mode, tex, len(uv), uv.ctypes.data, duv.ctypes.data, out.ctypes.data)
        return out

    def hit_geometry(self, scene, o, d, trig_mode=TRIG_LIBM):
        """{p, n, ns, dpdu, dpdv, dndu, dndv} (7, 3) of the closest hit along (o, d), or None."""
        o, d = _f32(o), _f32(d)
        out = np.zeros(24, np.float32)
        self.lib.oracle_hit_geometry.argtypes = [c_vp, ctypes.c_int, c_vp, c_vp, c_vp]
        if not self.lib.oracle_hit_geometry(scene.desc, trig_mode, o.ctypes.data, d.ctypes.data, out.ctypes.data):
            return None
        return out[:21].reshape(7, 3)

    def camera_hit_differentials(self, scene, pfx, pfy, trig_mode=TRIG_PORTABLE):
        out = np.zeros(6, np.float32)
        if not self.lib.oracle_camera_hit_differentials(scene.desc, trig_mode, pfx, pfy, out.ctypes.data):
            return None
        return out

    def distribution1d(self, func, u, continuous=False):
        """Distribution1D(func).SampleDiscrete(u) -> (offset, pdf) or SampleContinuous(u) -> (value, pdf, offset)."""
        func = _f32(func)
        value, pdf = ctypes.c_float(), ctypes.c_float()
        off = self.lib.oracle_distribution1d(func.ctypes.data, len(func), 1 if continuous else 0, float(u), ctypes.byref(value),
                                             ctypes.byref(pdf))
        assert off >= 0
        return (value.value, pdf.value, off) if continuous else (off, pdf.value)

    def log(self, x, trig_mode=TRIG_PORTABLE):
        return np.array([self.lib.oracle_log(trig_mode, float(v)) for v in _f32(x)], np.float32)

    def sincos(self, x, trig_mode=TRIG_PORTABLE):
        x = _f32(x)
        out = np.empty((len(x), 2), np.float32)
        for i in range(len(x)):
            self.lib.oracle_sincos(trig_mode, float(x[i]), out[i].ctypes.data, out[i, 1:].ctypes.data)
        return out

    def sincos_d(self, x, trig_mode=TRIG_PORTABLE):
        x = np.ascontiguousarray(x, np.float64)
        out = np.empty((len(x), 2), np.float64)
        for i in range(len(x)):
            self.lib.oracle_sincos_d(trig_mode, float(x[i]), out[i].ctypes.data, out[i, 1:].ctypes.data)
        return out

    def acos(self, x, trig_mode=TRIG_PORTABLE):
        return np.array([self.lib.oracle_acos(trig_mode, float(v)) for v in _f32(x)], np.float32)
