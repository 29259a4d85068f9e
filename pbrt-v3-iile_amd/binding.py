"""ctypes bindings over the two C-ABI libraries of the path.

    libiile_host.so  scene preparation + film finalisation   (include/iile_host.h)
    libiile_gpu.so   gfx950 wavefront path tracer            (include/iile_gpu.h)
    libiile_dist.so  the multi-GPU film merge over RCCL      (include/iile_dist.h)

Python is plumbing only (tests, bench.py, __graft_entry__): every number comes
out of the HIP kernels. There is no CPU fallback here — if libiile_gpu.so is
missing or no GPU is visible, the calls raise.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
REPO_ROOT = os.path.dirname(_HERE)
DEFAULT_SCENE = os.path.join(REPO_ROOT, "scenes", "killeroo-simple.pbrt")

c_i32, c_u32, c_u64, c_f32, c_f64 = (ctypes.c_int32, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_float,
                                     ctypes.c_double)
c_vp = ctypes.c_void_p


class HostOverrides(ctypes.Structure):
    _fields_ = [("xres", c_i32), ("yres", c_i32), ("spp", c_i32), ("max_depth", c_i32), ("sampler", c_i32),
                ("accel_split", c_i32), ("bvh_build", c_vp), ("quick_render", c_i32)]


SAMPLERS = {None: 0, "": 0, "halton": 1, "sobol": 2}  # IILE_SAMPLER_* of include/iile_host.h
SPLITS = {None: 0, "": 0, "sah": 1, "hlbvh": 2, "middle": 3, "equal": 4}  # IILE_SPLIT_*


class HostSceneInfo(ctypes.Structure):
    _fields_ = [(n, c_i32) for n in ("n_prims n_triangles n_spheres n_meshes n_nodes n_interior_nodes "
                                      "n_leaf_nodes n_materials n_lights xres yres spp max_depth probe_hemi_size integrator").split()]


class FilmDesc(ctypes.Structure):
    _fields_ = [(n, c_i32) for n in "xres yres crop_x0 crop_y0 crop_x1 crop_y1 samp_x0 samp_y0 samp_x1 samp_y1".split()
                ] + [(n, c_f32) for n in "filter_rx filter_ry scale max_sample_luminance".split()]


class Texture(ctypes.Structure):
    """iile_texture (include/iile_scene.h)."""
    _fields_ = [("n_levels", c_i32), ("wrap", c_i32), ("trilinear", c_i32), ("max_aniso", c_f32), ("su", c_f32),
                ("sv", c_f32), ("du", c_f32), ("dv", c_f32), ("level_w", c_i32 * 16), ("level_h", c_i32 * 16),
                ("level_offset", ctypes.c_int64 * 16)]


class Light(ctypes.Structure):
    """iile_light (include/iile_scene.h)."""
    _fields_ = [("lemit", c_f32 * 3), ("two_sided", c_i32), ("sphere", c_i32), ("type", c_i32), ("pos", c_f32 * 3), ("w2l", c_f32 * 9),
                ("cos_total_width", c_f32), ("cos_falloff_start", c_f32), ("world_radius", c_f32), ("prim", c_i32), ("n_samples", c_i32),
                ("l2w", c_f32 * 9), ("env_tex", c_i32), ("dist_w", c_i32), ("dist_h", c_i32), ("dist_offset", ctypes.c_int64)]


class RenderParams(ctypes.Structure):
    _fields_ = [("k_begin", c_i32), ("k_end", c_i32), ("tile_rank", c_i32), ("tile_nranks", c_i32),
                ("spp_per_pass", c_i32), ("collect_stats", c_i32), ("time_kernels", c_i32),
                ("film_on_device", c_i32), ("stream", c_vp)]


class IisptTask(ctypes.Structure):
    """iile_iispt_task (include/iile_scene.h): one task of the IISPT render runner."""
    _fields_ = [("x0", c_i32), ("y0", c_i32), ("x1", c_i32), ("y1", c_i32), ("tilesize", c_i32), ("counter_base", c_u32),
                ("rng_seed", c_u64)]

    def grid(self):
        def count(a0, a1, ts):
            n, t = 1, a0
            while t != a1 - 1:
                t = min(t + ts, a1 - 1)
                n += 1
            return n
        return count(self.x0, self.x1, self.tilesize), count(self.y0, self.y1, self.tilesize)


# iile_bvh_node (include/iile_scene.h): LinearBVHNode
BVH_NODE = np.dtype([("bmin", "<f4", 3), ("bmax", "<f4", 3), ("offset", "<i4"), ("nprims", "<u2"), ("axis", "u1"), ("pad", "u1")])


class BvhBuildStats(ctypes.Structure):
    _fields_ = [(n, c_f32) for n in "ms_total ms_morton ms_sort ms_treelets ms_upper ms_flatten ms_download".split()] + [
        (n, c_i32) for n in "n_treelets n_nodes n_interior n_leaf".split()]


class SceneDescHead(ctypes.Structure):
    """The first members of iile_scene_desc (include/iile_scene.h): the flattened BVH and the primitives in BVH order."""
    _fields_ = [("n_nodes", c_i32), ("nodes", c_vp), ("n_prims", c_i32), ("prim_flags", c_vp), ("prim_material", c_vp),
                ("prim_light", c_vp), ("prim_shape", c_vp), ("tri_p", c_vp)]


class GpuStats(ctypes.Structure):
    _fields_ = [(n, c_u64) for n in ("camera_rays closest_rays shadow_rays nodes_closest nodes_any tri_tests "
                                     "tri_hits sphere_tests nee_evals zero_radiance").split()] + [
        ("path_length", c_u64 * 8),
        ("ms_total", c_f64), ("ms_generate", c_f64), ("ms_extend", c_f64), ("ms_shade", c_f64),
        ("ms_connect", c_f64), ("ms_film", c_f64),
        ("n_extend_launches", c_i32), ("n_connect_launches", c_i32), ("n_shade_launches", c_i32),
        ("n_passes", c_i32), ("n_paths", c_u64), ("workspace_bytes", c_u64),
        ("ext_rays", c_u64), ("ext_nodes", c_u64), ("ext_tri_tests", c_u64), ("ext_sphere_tests", c_u64),
        ("any_tri_tests", c_u64), ("ms_shadow", c_f64), ("ms_mis", c_f64), ("ms_resolve", c_f64),
        ("mis_rays_traced", c_u64), ("ext_rays_traced", c_u64)]

    def as_dict(self):
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = list(v) if name == "path_length" else v
        return d


# every symbol the headers declare; checked by tests/test_abi.py
HOST_SYMBOLS = ["iile_host_load_pbrt", "iile_host_scene_desc", "iile_host_scene_film", "iile_host_scene_get_info",
                "iile_host_scene_free", "iile_host_film_to_rgb", "iile_host_write_pfm", "iile_host_last_error", "iile_host_read_image",
                "iile_host_scene_texture", "iile_host_scene_texture_level", "iile_host_scene_filter_table",
                "iile_host_sobol_matrices", "iile_host_sobol_vdc", "iile_host_write_exr", "iile_host_write_image",
                "iile_host_scene_film_filename", "iile_host_scene_light"]
class NetWeights(ctypes.Structure):
    """iile_iispt_net_weights (include/iile_gpu.h)."""
    _fields_ = [("conv_weight", c_vp * 15), ("conv_bias", c_vp * 15), ("bn_weight", c_vp * 5), ("bn_bias", c_vp * 5),
                ("bn_mean", c_vp * 5), ("bn_var", c_vp * 5), ("bn_eps", c_f32)]


class DirectParams(ctypes.Structure):  # iile_direct_params
    _fields_ = [("n_passes", ctypes.c_int32), ("first_pass", ctypes.c_int32), ("accumulate", ctypes.c_int32), ("film_on_device", ctypes.c_int32),
                ("stream", ctypes.c_void_p)]


GPU_SYMBOLS = ["iile_device_count", "iile_last_error", "iile_scene_create", "iile_scene_destroy", "iile_render",
               "iile_trace_closest", "iile_trace_any", "iile_halton_samples", "iile_camera_rays", "iile_li_samples",
               "iile_bsdf_eval", "iile_bsdf_sample", "iile_trig_probe", "iile_texture_eval", "iile_render_probes",
               "iile_device_select", "iile_device_alloc", "iile_device_free", "iile_device_download", "iile_device_upload", "iile_device_zero",
               "iile_stream_create", "iile_stream_wait", "iile_stream_destroy",
               "iile_iispt_film_add", "iile_iispt_film_merge",
               "iile_iispt_hemi_points", "iile_iispt_gather", "iile_iispt_hemi_points_batch", "iile_iispt_gather_batch", "iile_bvh_build_hlbvh", "iile_bvh_pack_probe", "iile_render_direct",
               "iile_wide_ref_shift", "iile_render_status", "iile_test_patch_capacity", "iile_iispt_net_create", "iile_iispt_net_load", "iile_iispt_net_forward", "iile_iispt_net_predict", "iile_iispt_net_destroy"]
DIST_SYMBOLS = ["iile_dist_unique_id", "iile_dist_create", "iile_dist_create_deadline", "iile_dist_abort", "iile_dist_wait", "iile_dist_destroy", "iile_dist_rank", "iile_dist_size", "iile_dist_ranks_seen",
                "iile_dist_film_reduce", "iile_dist_monitor_reduce", "iile_dist_barrier", "iile_dist_sum_u64", "iile_dist_max_f64",
                "iile_dist_rendezvous_file", "iile_dist_rendezvous_file_token", "iile_dist_rendezvous_done", "iile_dist_all_ok",
                "iile_dist_last_error"]
DIST_ID_BYTES = 128

_host = None
_gpu = None
_dist = None


def host_lib():
    global _host
    if _host is None:
        path = os.path.join(LIB_DIR, "libiile_host.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run __graft_entry__.build() (make -C pbrt-v3-iile_amd/csrc host)")
        lib = ctypes.CDLL(path)
        lib.iile_host_last_error.restype = ctypes.c_char_p
        lib.iile_host_load_pbrt.argtypes = [ctypes.c_char_p, ctypes.POINTER(HostOverrides), ctypes.POINTER(c_vp)]
        lib.iile_host_scene_desc.restype = c_vp
        lib.iile_host_scene_desc.argtypes = [c_vp]
        lib.iile_host_scene_film.restype = ctypes.POINTER(FilmDesc)
        lib.iile_host_scene_film.argtypes = [c_vp]
        lib.iile_host_scene_get_info.argtypes = [c_vp, ctypes.POINTER(HostSceneInfo)]
        lib.iile_host_scene_free.argtypes = [c_vp]
        lib.iile_host_scene_free.restype = None
        lib.iile_host_film_to_rgb.argtypes = [ctypes.POINTER(FilmDesc), c_vp, c_vp]
        lib.iile_host_write_pfm.argtypes = [ctypes.c_char_p, c_vp, c_i32, c_i32]
        lib.iile_host_read_image.argtypes = [ctypes.c_char_p, ctypes.POINTER(c_i32), ctypes.POINTER(c_i32), c_vp]
        lib.iile_host_write_exr.argtypes = [ctypes.c_char_p, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32]
        lib.iile_host_write_image.argtypes = [ctypes.c_char_p, ctypes.POINTER(FilmDesc), c_vp]
        lib.iile_host_scene_film_filename.argtypes = [c_vp]
        lib.iile_host_scene_film_filename.restype = ctypes.c_char_p
        lib.iile_host_scene_texture.argtypes = [c_vp, c_i32, ctypes.POINTER(Texture)]
        lib.iile_host_scene_texture_level.argtypes = [c_vp, c_i32, c_i32, c_vp]
        lib.iile_host_scene_filter_table.argtypes = [c_vp, c_vp]
        lib.iile_host_sobol_matrices.argtypes = [c_i32, c_vp, c_vp]
        lib.iile_host_sobol_vdc.argtypes = [c_i32, c_vp, c_vp]
        _host = lib
    return _host


def sobol_matrices(n_dims):
    """(SobolMatrices32, SobolMatrices64) as the host builds them: (n_dims, 52) uint32 / uint64."""
    m32 = np.zeros((n_dims, 52), np.uint32)
    m64 = np.zeros((n_dims, 52), np.uint64)
    if host_lib().iile_host_sobol_matrices(int(n_dims), m32.ctypes.data, m64.ctypes.data) != 0:
        raise RuntimeError(host_lib().iile_host_last_error().decode())
    return m32, m64


def sobol_vdc(log2_resolution):
    """(VdCSobolMatrices[m - 1], VdCSobolMatricesInv[m - 1]) as the host builds them: 52 uint64 each."""
    vdc = np.zeros(52, np.uint64)
    inv = np.zeros(52, np.uint64)
    if host_lib().iile_host_sobol_vdc(int(log2_resolution), vdc.ctypes.data, inv.ctypes.data) != 0:
        raise RuntimeError(host_lib().iile_host_last_error().decode())
    return vdc, inv


def read_image(path):
    """ReadImage for .pfm / .png / .tga: (H, W, 3) float32, row 0 = top scanline."""
    lib = host_lib()
    w, h = c_i32(), c_i32()
    if lib.iile_host_read_image(os.fsencode(path), ctypes.byref(w), ctypes.byref(h), None) != 0:
        raise RuntimeError(lib.iile_host_last_error().decode())
    rgb = np.empty((h.value, w.value, 3), np.float32)
    if lib.iile_host_read_image(os.fsencode(path), ctypes.byref(w), ctypes.byref(h), rgb.ctypes.data) != 0:
        raise RuntimeError(lib.iile_host_last_error().decode())
    return rgb


def write_exr(path, rgb, origin=(0, 0), display=None):
    """WriteImageEXR: (H, W, 3) float32 (row 0 = top scanline) as a half-float RGB OpenEXR file; `origin` = (x0, y0) of
    the data window, `display` = (total_w, total_h) of the display window (default: the image itself)."""
    rgb = _f32(rgb)
    h, w = rgb.shape[:2]
    x0, y0 = origin
    tw, th = display if display else (x0 + w, y0 + h)
    if host_lib().iile_host_write_exr(os.fsencode(path), rgb.ctypes.data, x0, y0, x0 + w, y0 + h, tw, th) != 0:
        raise RuntimeError(host_lib().iile_host_last_error().decode())


def gpu_lib():
    """The HIP library. Fails loudly when it has not been built."""
    global _gpu
    if _gpu is None:
        # IILE_GPU_LIB selects an alternative build of the same library (kernel A/B experiments)
        path = os.environ.get("IILE_GPU_LIB") or os.path.join(LIB_DIR, "libiile_gpu.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: the HIP extension was not built; there is no CPU fallback "
                               "(run __graft_entry__.build())")
        lib = ctypes.CDLL(path)
        lib.iile_last_error.restype = ctypes.c_char_p
        lib.iile_scene_create.argtypes = [c_vp, ctypes.POINTER(c_vp)]
        lib.iile_scene_destroy.argtypes = [c_vp]
        lib.iile_scene_destroy.restype = None
        lib.iile_render.argtypes = [c_vp, ctypes.POINTER(RenderParams), c_vp, ctypes.POINTER(GpuStats)]
        lib.iile_trace_closest.argtypes = [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(GpuStats)]
        lib.iile_trace_any.argtypes = [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(GpuStats)]
        lib.iile_halton_samples.argtypes = [c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp]
        lib.iile_camera_rays.argtypes = [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp]
        lib.iile_li_samples.argtypes = [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]
        lib.iile_bsdf_eval.argtypes = [c_vp, c_i32, c_i32, c_vp, c_vp, c_vp]
        lib.iile_texture_eval.argtypes = [c_vp, c_i32, c_i32, c_vp, c_vp, c_vp]
        lib.iile_render_probes.argtypes = [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, ctypes.POINTER(GpuStats), c_vp]
        lib.iile_bsdf_sample.argtypes = [c_vp, c_i32, c_i32, c_vp, c_vp, c_vp]
        lib.iile_trig_probe.argtypes = [c_i32, c_vp, c_vp]
        lib.iile_iispt_hemi_points.argtypes = [c_vp, ctypes.POINTER(IisptTask), c_vp, c_vp, c_vp]
        lib.iile_iispt_gather.argtypes = [c_vp, ctypes.POINTER(IisptTask), c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32]
        lib.iile_iispt_hemi_points_batch.argtypes = [c_vp, ctypes.POINTER(IisptTask), c_i32, c_vp, c_vp, c_vp, c_vp]
        lib.iile_iispt_gather_batch.argtypes = [c_vp, ctypes.POINTER(IisptTask), c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_vp]
        lib.iile_bvh_build_hlbvh.argtypes = [c_i32, c_vp, c_i32, c_vp, ctypes.POINTER(c_i32), c_vp, ctypes.POINTER(BvhBuildStats)]
        lib.iile_bvh_pack_probe.argtypes = [c_i32, c_vp, c_i32, c_vp, c_vp, ctypes.POINTER(c_i32)]
        lib.iile_render_status.argtypes = [c_vp, c_vp]
        lib.iile_test_patch_capacity.argtypes = [c_vp, c_u32]
        lib.iile_iispt_net_create.argtypes = [ctypes.POINTER(NetWeights), ctypes.POINTER(c_vp)]
        lib.iile_iispt_net_load.argtypes = [ctypes.c_char_p, ctypes.POINTER(c_vp)]
        lib.iile_iispt_net_forward.argtypes = [c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_i32]
        lib.iile_iispt_net_predict.argtypes = [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]
        lib.iile_iispt_film_add.argtypes = [c_vp, ctypes.POINTER(IisptTask), c_i32, c_vp, c_vp, c_i32, c_i32, c_vp]
        lib.iile_iispt_film_merge.argtypes = [c_vp, c_vp, ctypes.c_int64, c_vp, c_vp]
        lib.iile_device_upload.argtypes = [c_vp, c_vp, ctypes.c_uint64, c_vp]
        lib.iile_device_zero.argtypes = [c_vp, ctypes.c_uint64, c_vp]
        lib.iile_iispt_net_destroy.argtypes = [c_vp]
        lib.iile_iispt_net_destroy.restype = None
        _gpu = lib
    return _gpu


def dist_lib():
    """The RCCL film-merge library (needs librccl; loads without a GPU)."""
    global _dist
    if _dist is None:
        path = os.path.join(LIB_DIR, "libiile_dist.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run __graft_entry__.build() (make -C pbrt-v3-iile_amd/csrc dist)")
        lib = ctypes.CDLL(path)
        lib.iile_dist_last_error.restype = ctypes.c_char_p
        lib.iile_dist_unique_id.argtypes = [c_vp]
        lib.iile_dist_create.argtypes = [c_vp, c_i32, c_i32, ctypes.POINTER(c_vp)]
        lib.iile_dist_create_deadline.argtypes = [c_vp, c_i32, c_i32, ctypes.c_double, ctypes.POINTER(c_vp)]
        lib.iile_dist_wait.argtypes = [c_vp, c_vp]
        lib.iile_dist_abort.argtypes = [c_vp]
        lib.iile_dist_abort.restype = None
        lib.iile_dist_destroy.argtypes = [c_vp]
        lib.iile_dist_destroy.restype = None
        lib.iile_dist_rank.argtypes = [c_vp]
        lib.iile_dist_size.argtypes = [c_vp]
        lib.iile_dist_film_reduce.argtypes = [c_vp, c_vp, ctypes.c_int64, c_i32, c_vp]
        lib.iile_dist_barrier.argtypes = [c_vp, c_vp]
        lib.iile_dist_sum_u64.argtypes = [c_vp, c_vp, c_i32]
        lib.iile_dist_max_f64.argtypes = [c_vp, c_vp, c_i32]
        lib.iile_dist_rendezvous_file.argtypes = [ctypes.c_char_p, c_i32, c_vp, c_i32]
        lib.iile_dist_rendezvous_file_token.argtypes = [ctypes.c_char_p, c_i32, ctypes.c_uint64, c_vp, c_i32]
        lib.iile_dist_rendezvous_done.argtypes = [c_vp, ctypes.c_char_p]
        lib.iile_dist_all_ok.argtypes = [c_vp, c_i32, ctypes.POINTER(c_i32)]
        _dist = lib
    return _dist


class Dist:
    """One rank of the film-merge communicator (include/iile_dist.h)."""

    @staticmethod
    def unique_id():
        buf = (ctypes.c_uint8 * DIST_ID_BYTES)()
        lib = dist_lib()
        if lib.iile_dist_unique_id(buf) != 0:
            raise RuntimeError(f"iile_dist_unique_id failed: {lib.iile_dist_last_error().decode()}")
        return bytes(buf)

    def __init__(self, unique_id, rank, nranks, timeout_s=None):
        """timeout_s: None -> iile_dist_create (RCCL's blocking set-up: a launcher owns the job's tear-down); a number ->
        iile_dist_create_deadline (non-blocking set-up polled against that deadline, which then bounds every wait on the communicator)."""
        lib = dist_lib()
        self._c = c_vp()
        buf = (ctypes.c_uint8 * DIST_ID_BYTES).from_buffer_copy(unique_id)
        if timeout_s is None:
            rc, what = lib.iile_dist_create(buf, int(rank), int(nranks), ctypes.byref(self._c)), "iile_dist_create"
        else:
            rc, what = lib.iile_dist_create_deadline(buf, int(rank), int(nranks), float(timeout_s), ctypes.byref(self._c)), "iile_dist_create_deadline"
        if rc != 0:
            err = RuntimeError(f"{what} failed ({rc}): {lib.iile_dist_last_error().decode()}")
            err.code = rc
            raise err
        self.rank, self.size = int(rank), int(nranks)
        lib.iile_dist_ranks_seen.restype = ctypes.c_int
        lib.iile_dist_ranks_seen.argtypes = [c_vp]
        self.ranks_seen = int(lib.iile_dist_ranks_seen(self._c))  # ncclCommCount

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {dist_lib().iile_dist_last_error().decode()}")

    def film_reduce(self, film_device_ptr, n_pixels, root=0, stream=None):
        self._check(dist_lib().iile_dist_film_reduce(self._c, c_vp(int(film_device_ptr)), int(n_pixels), int(root),
                                                     c_vp(stream) if stream else None), "iile_dist_film_reduce")

    def barrier(self, stream=None):
        self._check(dist_lib().iile_dist_barrier(self._c, c_vp(stream) if stream else None), "iile_dist_barrier")

    def wait(self, stream=None):
        """iile_dist_wait: what was queued on `stream` (the film merge) has finished, or the communicator's deadline has passed."""
        self._check(dist_lib().iile_dist_wait(self._c, c_vp(stream) if stream else None), "iile_dist_wait")

    def all_ok(self, ok):
        out = c_i32(0)
        self._check(dist_lib().iile_dist_all_ok(self._c, 1 if ok else 0, ctypes.byref(out)), "iile_dist_all_ok")
        return bool(out.value)

    def abort(self):
        if self._c:
            dist_lib().iile_dist_abort(self._c)
            self._c = c_vp()

    def sum_u64(self, values):
        a = np.ascontiguousarray(values, dtype=np.uint64).copy()
        self._check(dist_lib().iile_dist_sum_u64(self._c, a.ctypes.data, len(a)), "iile_dist_sum_u64")
        return a

    def max_f64(self, values):
        a = np.ascontiguousarray(values, dtype=np.float64).copy()
        self._check(dist_lib().iile_dist_max_f64(self._c, a.ctypes.data, len(a)), "iile_dist_max_f64")
        return a

    def close(self):
        if self._c:
            dist_lib().iile_dist_destroy(self._c)
            self._c = c_vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class HostScene:
    """Scene loaded and flattened by libiile_host (ParseFile + MakeScene in the reference)."""

    def __init__(self, path=DEFAULT_SCENE, xres=0, yres=0, spp=0, max_depth=0, sampler=None, accel_split=None, bvh_on_device=False,
                 quick=False):
        """sampler: None keeps the scene file's; "sobol" is what the fork's path integrator renders with under
        IILE_PATH_SAMPLES_OVERRIDE (src/integrators/path.cpp:202-212). accel_split: BVHAccel's "splitmethod" in place of
        the file's. bvh_on_device: split method "hlbvh" is built by libiile_gpu's iile_bvh_build_hlbvh,
        plugged into the host loader as its bvh_build hook."""
        lib = host_lib()
        self._h = c_vp()
        hook = None
        if bvh_on_device is True:
            hook = ctypes.cast(gpu_lib().iile_bvh_build_hlbvh, c_vp)
        elif bvh_on_device:  # any other builder with iile_host_overrides::bvh_build's signature (a ctypes function object)
            hook = ctypes.cast(bvh_on_device, c_vp)
        ov = HostOverrides(int(xres), int(yres), int(spp), int(max_depth), SAMPLERS[sampler], SPLITS[accel_split], hook, int(bool(quick)))
        rc = lib.iile_host_load_pbrt(os.fsencode(path), ctypes.byref(ov), ctypes.byref(self._h))
        if rc != 0:
            raise RuntimeError(f"iile_host_load_pbrt({path}) failed: {lib.iile_host_last_error().decode()}")
        self.desc = lib.iile_host_scene_desc(self._h)
        self._film = lib.iile_host_scene_film(self._h)
        self.film = self._film.contents
        info = HostSceneInfo()
        lib.iile_host_scene_get_info(self._h, ctypes.byref(info))
        self.info = {n: getattr(info, n) for n, _ in HostSceneInfo._fields_}

    @property
    def film_shape(self):
        f = self.film
        return (f.crop_y1 - f.crop_y0, f.crop_x1 - f.crop_x0)

    def bvh(self):
        """(flattened nodes as a BVH_NODE array, triangle vertices (n_prims, 9) in BVH order, prim_shape (n_prims,)): copies."""
        head = ctypes.cast(self.desc, ctypes.POINTER(SceneDescHead)).contents
        nodes = np.frombuffer(ctypes.string_at(head.nodes, head.n_nodes * BVH_NODE.itemsize), dtype=BVH_NODE).copy()
        tri_p = np.frombuffer(ctypes.string_at(head.tri_p, head.n_prims * 36), dtype=np.float32).reshape(-1, 9).copy()
        shape = np.frombuffer(ctypes.string_at(head.prim_shape, head.n_prims * 4), dtype=np.int32).copy()
        return nodes, tri_p, shape

    def film_to_rgb(self, film_xyzw):
        """Film::to_rgb_array on a (H, W, 4) {X,Y,Z,weight} film."""
        h, w = self.film_shape
        film = _f32(film_xyzw).reshape(h, w, 4)
        rgb = np.empty((h, w, 3), np.float32)
        rc = host_lib().iile_host_film_to_rgb(self._film, film.ctypes.data, rgb.ctypes.data)
        if rc != 0:
            raise RuntimeError(host_lib().iile_host_last_error().decode())
        return rgb

    def filter_table(self):
        """(Film::filterTable as a (16, 16) array [y][x], wide?)"""
        t = np.empty((16, 16), np.float32)
        wide = host_lib().iile_host_scene_filter_table(self._h, t.ctypes.data)
        return t, bool(wide)

    def light(self, index):
        """iile_light number `index` of the scene (a copy)."""
        lt = Light()
        if host_lib().iile_host_scene_light(self._h, int(index), ctypes.byref(lt)) != 0:
            raise RuntimeError(host_lib().iile_host_last_error().decode())
        return lt

    def texture(self, index):
        """(iile_texture, [level arrays (h, w, 3), row 0 = bottom scanline]) of image texture `index`."""
        lib = host_lib()
        t = Texture()
        if lib.iile_host_scene_texture(self._h, index, ctypes.byref(t)) != 0:
            raise RuntimeError(lib.iile_host_last_error().decode())
        levels = []
        for l in range(t.n_levels):
            a = np.empty((t.level_h[l], t.level_w[l], 3), np.float32)
            lib.iile_host_scene_texture_level(self._h, index, l, a.ctypes.data)
            levels.append(a)
        return t, levels

    @property
    def film_filename(self):
        """The scene file's Film "filename"."""
        return host_lib().iile_host_scene_film_filename(self._h).decode()

    def write_image(self, path, rgb):
        """Film::WriteImage: .exr (data window = the cropped pixel bounds) or .pfm by the extension."""
        rgb = _f32(rgb)
        if host_lib().iile_host_write_image(os.fsencode(path), self._film, rgb.ctypes.data) != 0:
            raise RuntimeError(host_lib().iile_host_last_error().decode())

    def write_pfm(self, path, rgb):
        rgb = _f32(rgb)
        rc = host_lib().iile_host_write_pfm(os.fsencode(path), rgb.ctypes.data, rgb.shape[1], rgb.shape[0])
        if rc != 0:
            raise RuntimeError(host_lib().iile_host_last_error().decode())

    def close(self):
        if self._h:
            host_lib().iile_host_scene_free(self._h)
            self._h = c_vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GpuScene:
    """Device-resident scene + wavefront integrator (libiile_gpu)."""

    def __init__(self, host_scene):
        lib = gpu_lib()
        self.host = host_scene
        self._s = c_vp()
        rc = lib.iile_scene_create(host_scene.desc, ctypes.byref(self._s))
        if rc != 0:
            raise RuntimeError(f"iile_scene_create failed ({rc}): {lib.iile_last_error().decode()}")

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {gpu_lib().iile_last_error().decode()}")

    def render(self, k_begin=0, k_end=0, tile_rank=0, tile_nranks=1, spp_per_pass=0, collect_stats=False,
               time_kernels=False, film_device_ptr=None, stream=None, want_stats=True):
        """SamplerIntegrator::Render. Returns (film, stats): film is a (H, W, 4) float32
        array, or None when `film_device_ptr` (a raw device pointer) receives it."""
        prm = RenderParams(int(k_begin), int(k_end), int(tile_rank), int(tile_nranks), int(spp_per_pass),
                           int(bool(collect_stats)), int(time_kernels), int(film_device_ptr is not None),
                           c_vp(stream) if stream else None)
        st = GpuStats()
        if film_device_ptr is not None:
            film, ptr = None, c_vp(int(film_device_ptr))
        else:
            h, w = self.host.film_shape
            film = np.zeros((h, w, 4), np.float32)
            ptr = film.ctypes.data
        rc = gpu_lib().iile_render(self._s, ctypes.byref(prm), ptr, ctypes.byref(st) if want_stats else None)
        self._check(rc, "iile_render")
        return film, (st.as_dict() if want_stats else None)

    def render_status(self, stream=None):
        """iile_render_status: waits for `stream`; raises if the last asynchronous render's exact film finish overflowed."""
        self._check(gpu_lib().iile_render_status(self._s, c_vp(stream) if stream else None), "iile_render_status")

    def test_patch_capacity(self, capacity):
        self._check(gpu_lib().iile_test_patch_capacity(self._s, int(capacity)), "iile_test_patch_capacity")

    def trace_closest(self, o, d, tmax, instrumented=True):
        """instrumented=False runs the traversal of the uninstrumented render kernels."""
        o, d, tmax = _f32(o), _f32(d), _f32(tmax)
        n = len(tmax)
        prim = np.empty(n, np.int32)
        tb = np.empty((n, 4), np.float32)
        st = GpuStats()
        self._check(gpu_lib().iile_trace_closest(self._s, n, o.ctypes.data, d.ctypes.data, tmax.ctypes.data,
                                                 prim.ctypes.data, tb.ctypes.data,
                                                 ctypes.byref(st) if instrumented else None), "iile_trace_closest")
        return prim, tb, st.as_dict()

    def trace_any(self, o, d, tmax, instrumented=True):
        o, d, tmax = _f32(o), _f32(d), _f32(tmax)
        n = len(tmax)
        hit = np.empty(n, np.int32)
        st = GpuStats()
        self._check(gpu_lib().iile_trace_any(self._s, n, o.ctypes.data, d.ctypes.data, tmax.ctypes.data,
                                             hit.ctypes.data, ctypes.byref(st) if instrumented else None),
                    "iile_trace_any")
        return hit, st.as_dict()

    def halton_samples(self, px, py, k, dim0, ndims):
        px, py, k = _i32(px), _i32(py), _i32(k)
        n = len(px)
        out = np.empty((n, ndims), np.float32)
        idx = np.empty(n, np.uint32)
        self._check(gpu_lib().iile_halton_samples(self._s, n, px.ctypes.data, py.ctypes.data, k.ctypes.data, dim0,
                                                  ndims, out.ctypes.data, idx.ctypes.data), "iile_halton_samples")
        return out, idx

    def camera_rays(self, pfilm, plens=None):
        pfilm = _f32(pfilm)
        n = len(pfilm)
        o = np.empty((n, 3), np.float32)
        d = np.empty((n, 3), np.float32)
        pl = _f32(plens) if plens is not None else None
        self._check(gpu_lib().iile_camera_rays(self._s, n, pfilm.ctypes.data, pl.ctypes.data if pl is not None else None,
                                               o.ctypes.data, d.ctypes.data), "iile_camera_rays")
        return o, d

    def li_samples(self, px, py, k):
        px, py, k = _i32(px), _i32(py), _i32(k)
        n = len(px)
        L = np.empty((n, 3), np.float32)
        nr = np.empty((n, 2), np.int32)
        self._check(gpu_lib().iile_li_samples(self._s, n, px.ctypes.data, py.ctypes.data, k.ctypes.data, L.ctypes.data,
                                              nr.ctypes.data), "iile_li_samples")
        return L, nr

    def bsdf_eval(self, mat, wo, wi):
        wo, wi = _f32(wo), _f32(wi)
        out = np.empty((len(wo), 4), np.float32)
        self._check(gpu_lib().iile_bsdf_eval(self._s, len(wo), mat, wo.ctypes.data, wi.ctypes.data, out.ctypes.data),
                    "iile_bsdf_eval")
        return out

    def render_probes(self, pos, direction, hemi=None, device_out=None, stream=None):
        """IISPT probe pass: (n, 3) origins and directions -> intensity (n, hemi, hemi, 3), camera-space normals
        (n, hemi, hemi, 3), distances (n, hemi, hemi), [y][x] in raster order; plus the stats dict. hemi is the
        scene's probe film size (iile_scene_desc::probe.hemi_size through iile_host_scene_get_info); passing another
        value is an error. device_out: three device pointers (ints), each with room for n * hemi * hemi pixels, to
        write the images to instead (they then stay in HBM). stream: the HIP stream the pass is queued on (None: the null stream)."""
        pos, direction = _f32(pos).reshape(-1, 3), _f32(direction).reshape(-1, 3)
        n = len(pos)
        scene_hemi = int(self.host.info["probe_hemi_size"])
        if hemi is not None and int(hemi) != scene_hemi:
            raise ValueError(f"render_probes: the scene's probe films are {scene_hemi} x {scene_hemi}, not {hemi}")
        hemi = scene_hemi
        st = GpuStats()
        if device_out is not None:
            self._check(gpu_lib().iile_render_probes(self._s, n, pos.ctypes.data, direction.ctypes.data, device_out[0], device_out[1],
                                                     device_out[2], 1, ctypes.byref(st), stream), "iile_render_probes")
            return None, None, None, st.as_dict()
        inten = np.zeros((n, hemi, hemi, 3), np.float32)
        nrm = np.zeros((n, hemi, hemi, 3), np.float32)
        dist = np.zeros((n, hemi, hemi), np.float32)
        self._check(gpu_lib().iile_render_probes(self._s, n, pos.ctypes.data, direction.ctypes.data, inten.ctypes.data,
                                                 nrm.ctypes.data, dist.ctypes.data, 0, ctypes.byref(st), stream), "iile_render_probes")
        return inten, nrm, dist, st.as_dict()

    def iispt_hemi_points(self, task):
        """The hemi points of an IISPT task: (valid (ny, nx) uint8, aux ray origins (ny, nx, 3), directions (ny, nx, 3))."""
        nx, ny = task.grid()
        valid = np.zeros((ny, nx), np.uint8)
        pos = np.zeros((ny, nx, 3), np.float32)
        dr = np.zeros((ny, nx, 3), np.float32)
        self._check(gpu_lib().iile_iispt_hemi_points(self._s, ctypes.byref(task), valid.ctypes.data, pos.ctypes.data, dr.ctypes.data),
                    "iile_iispt_hemi_points")
        return valid, pos, dr

    def iispt_gather(self, task, valid, pos, direction, nn_films=None, nn_device_ptr=None, out_device_ptr=None):
        """The runner's per-pixel loop over the predicted hemispheres nn_films (ny, nx, 32, 32, 3) -> (h, w, 4)
        {f_beta * L, weight}; nn_device_ptr / out_device_ptr: raw device pointers instead of host arrays."""
        valid, pos, direction = np.ascontiguousarray(valid, np.uint8), _f32(pos), _f32(direction)
        h, w = task.y1 - task.y0, task.x1 - task.x0
        out = None if out_device_ptr is not None else np.zeros((h, w, 4), np.float32)
        nn = _f32(nn_films) if nn_device_ptr is None else None
        self._check(gpu_lib().iile_iispt_gather(self._s, ctypes.byref(task), valid.ctypes.data, pos.ctypes.data, direction.ctypes.data,
                                                nn.ctypes.data if nn is not None else c_vp(int(nn_device_ptr)), int(nn is None),
                                                out.ctypes.data if out is not None else c_vp(int(out_device_ptr)), int(out is None)),
                    "iile_iispt_gather")
        return out

    def iispt_hemi_points_batch(self, tasks, stream=None):
        """iile_iispt_hemi_points_batch: the hemi points of several tasks from one set of launches — (valid (n,), origins (n, 3),
        directions (n, 3)) with the tasks' hemi points one task after the other, each in its own row-by-row order. stream: the HIP
        stream the call's copies and kernels are queued on (None: the null stream)."""
        arr = (IisptTask * len(tasks))(*tasks)
        n = sum(t.grid()[0] * t.grid()[1] for t in tasks)
        valid, pos, dr = np.zeros(n, np.uint8), np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
        self._check(gpu_lib().iile_iispt_hemi_points_batch(self._s, arr, len(tasks), valid.ctypes.data, pos.ctypes.data, dr.ctypes.data, stream),
                    "iile_iispt_hemi_points_batch")
        return valid, pos, dr

    def iispt_gather_batch(self, tasks, valid, pos, direction, nn_films=None, nn_device_ptr=None, out_device_ptr=None, stream=None):
        """iile_iispt_gather_batch: the per-pixel loop of several tasks from one set of launches. valid / pos / direction / nn_films
        (n_hemi, 32, 32, 3) as iispt_hemi_points_batch orders them; the result is (n_pixels, 4), the tasks' pixels one task after the
        other, row-major inside a task (None when written to out_device_ptr).
        With out_device_ptr the call returns once its kernels are queued on `stream` (None: the null stream, which is PyTorch's
        default stream); they read the scene's shared scratch block: keep a scene's IISPT calls on one stream."""
        arr = (IisptTask * len(tasks))(*tasks)
        valid, pos, direction = np.ascontiguousarray(valid, np.uint8), _f32(pos), _f32(direction)
        n_pix = sum((t.x1 - t.x0) * (t.y1 - t.y0) for t in tasks)
        out = None if out_device_ptr is not None else np.zeros((n_pix, 4), np.float32)
        nn = _f32(nn_films) if nn_device_ptr is None else None
        self._check(gpu_lib().iile_iispt_gather_batch(self._s, arr, len(tasks), valid.ctypes.data, pos.ctypes.data, direction.ctypes.data,
                                                      nn.ctypes.data if nn is not None else c_vp(int(nn_device_ptr)), int(nn is None),
                                                      out.ctypes.data if out is not None else c_vp(int(out_device_ptr)), int(out is None), stream),
                    "iile_iispt_gather_batch")
        return out

    def iispt_film_add(self, tasks, out_device_ptr, film_device_ptr, stream=None):
        """iile_iispt_film_add: IisptFilmMonitor::add_n_samples for every pixel of the (non-overlapping) tasks, whose gathered
        {f_beta * L, weight} lie at out_device_ptr, into the (h, w, 4) float64 monitor at film_device_ptr."""
        h, w = self.host.film_shape
        arr = (IisptTask * len(tasks))(*tasks)
        self._check(gpu_lib().iile_iispt_film_add(self._s, arr, len(tasks), c_vp(int(out_device_ptr)), c_vp(int(film_device_ptr)), w, h,
                                                  c_vp(stream) if stream else None), "iile_iispt_film_add")

    def render_direct(self, n_passes, first_pass=0, film_device_ptr=None, accumulate=False, stream=None):
        """The IISPT direct pass (iile_render_direct): the direct film monitor {sum r, g, b, weight} as (h, w, 4) float64, or
        accumulated into device memory at film_device_ptr (returns None)."""
        h, w = self.host.film_shape
        out = None if film_device_ptr is not None else np.zeros((h, w, 4), np.float64)
        prm = DirectParams(int(n_passes), int(first_pass), int(bool(accumulate)), int(out is None), c_vp(stream) if stream else None)
        f = gpu_lib().iile_render_direct
        f.argtypes = [c_vp, ctypes.POINTER(DirectParams), c_vp]
        self._check(f(self._s, ctypes.byref(prm), out.ctypes.data if out is not None else c_vp(int(film_device_ptr))), "iile_render_direct")
        return out

    def texture_eval(self, tex, uv, duv):
        """ImageTexture::Evaluate at (n, 2) uv with (n, 4) differentials {dudx, dvdx, dudy, dvdy} -> (n, 3) RGB."""
        uv, duv = _f32(uv), _f32(duv)
        out = np.empty((len(uv), 3), np.float32)
        self._check(gpu_lib().iile_texture_eval(self._s, tex, len(uv), uv.ctypes.data, duv.ctypes.data, out.ctypes.data),
                    "iile_texture_eval")
        return out

    def bsdf_sample(self, mat, wo, u):
        wo, u = _f32(wo), _f32(u)
        out = np.empty((len(wo), 7), np.float32)
        self._check(gpu_lib().iile_bsdf_sample(self._s, len(wo), mat, wo.ctypes.data, u.ctypes.data, out.ctypes.data),
                    "iile_bsdf_sample")
        return out

    def close(self):
        if self._s:
            gpu_lib().iile_scene_destroy(self._s)
            self._s = c_vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# the reference's state_dict names in forward order (ml/iispt_net.py:27-88): the 15 convolutions, the 5 BatchNorm2d
NET_CONVS = ("encoder0.0", "encoder0.2", "encoder1.1", "encoder1.4", "encoder2.1", "encoder2.4", "encoder3.1", "encoder3.4",
             "decoder0.0", "decoder0.3", "decoder1.0", "decoder1.3", "decoder2.0", "decoder2.2", "decoder2.4")
NET_BNS = ("encoder1.3", "encoder2.3", "encoder3.3", "decoder0.2", "decoder1.2")


def save_net_weights(state_dict, path, bn_eps=1e-5):
    """The flat file iile_iispt_net_load reads (and `iile_pbrt --iisptNet=`): "IILENET1", eps, the tensors of a state_dict of
    IISPTNet in iile_iispt_net_weights' order, float32."""
    def arr(name):
        t = state_dict[name]
        return np.ascontiguousarray(t.detach().cpu().numpy() if hasattr(t, "detach") else t, dtype=np.float32)
    with open(path, "wb") as f:
        f.write(b"IILENET1")
        f.write(np.float32(bn_eps).tobytes())
        for k in NET_CONVS:
            f.write(arr(k + ".weight").tobytes())
            f.write(arr(k + ".bias").tobytes())
        for k in NET_BNS:
            for q in ("weight", "bias", "running_mean", "running_var"):
                f.write(arr(k + "." + q).tobytes())


def iispt_film_merge(direct_ptr, indirect_ptr, n_pixels, rgb_ptr, stream=None):
    """iile_iispt_film_merge: the two monitors normalised and added -> float RGB (device pointers)."""
    lib = gpu_lib()
    rc = lib.iile_iispt_film_merge(c_vp(int(direct_ptr)), c_vp(int(indirect_ptr)), int(n_pixels), c_vp(int(rgb_ptr)), c_vp(stream) if stream else None)
    if rc != 0:
        raise RuntimeError(f"iile_iispt_film_merge failed ({rc}): {lib.iile_last_error().decode()}")


class GpuNet:
    """The IISPT network on the device (iile_iispt_net_*): built from a state_dict with the reference's entry names
    (numpy arrays or torch tensors), run on device pointers."""

    def __init__(self, state_dict=None, bn_eps=1e-5, path=None):
        self._lib = gpu_lib()
        self._h = c_vp()
        if path is not None:   # an IILENET1 file (save_net_weights)
            rc = self._lib.iile_iispt_net_load(os.fsencode(path), ctypes.byref(self._h))
            if rc != 0:
                raise RuntimeError(f"iile_iispt_net_load failed ({rc}): {self._lib.iile_last_error().decode()}")
            return
        keep = []

        def arr(name):
            t = state_dict[name]
            a = np.ascontiguousarray(t.detach().cpu().numpy() if hasattr(t, "detach") else t, dtype=np.float32)
            keep.append(a)
            return a.ctypes.data

        w = NetWeights()
        for i, k in enumerate(NET_CONVS):
            w.conv_weight[i], w.conv_bias[i] = arr(k + ".weight"), arr(k + ".bias")
        for i, k in enumerate(NET_BNS):
            w.bn_weight[i], w.bn_bias[i] = arr(k + ".weight"), arr(k + ".bias")
            w.bn_mean[i], w.bn_var[i] = arr(k + ".running_mean"), arr(k + ".running_var")
        w.bn_eps = bn_eps
        rc = self._lib.iile_iispt_net_create(ctypes.byref(w), ctypes.byref(self._h))
        if rc != 0:
            raise RuntimeError(f"iile_iispt_net_create failed ({rc}): {self._lib.iile_last_error().decode()}")

    def forward(self, in_ptr, out_ptr, n, max_batch=0, stream=None, layer_out_ptr=None, layer=0):
        """(n, 7, 32, 32) -> (n, 3, 32, 32), device pointers; queued on `stream`."""
        rc = self._lib.iile_iispt_net_forward(self._h, in_ptr, out_ptr, int(n), int(max_batch), stream, layer_out_ptr, int(layer))
        if rc != 0:
            raise RuntimeError(f"iile_iispt_net_forward failed ({rc}): {self._lib.iile_last_error().decode()}")

    def predict(self, intensity_ptr, normals_ptr, distance_ptr, pred_ptr, n, film_rows=False, max_batch=0, stream=None, slot_ptr=None):
        """normalizeMapsDownstream -> network -> transformMapsUpstream over n rendered probes (device pointers; raster order in,
        (n, 32, 32, 3) out: ImageFilm row order when film_rows, else raster). slot_ptr: n int32 on the device — probe i's image goes to
        image slot[i] of pred_ptr."""
        rc = self._lib.iile_iispt_net_predict(self._h, intensity_ptr, normals_ptr, distance_ptr, pred_ptr, slot_ptr, int(n), int(bool(film_rows)),
                                              int(max_batch), stream)
        if rc != 0:
            raise RuntimeError(f"iile_iispt_net_predict failed ({rc}): {self._lib.iile_last_error().decode()}")

    def close(self):
        if self._h:
            self._lib.iile_iispt_net_destroy(self._h)
            self._h = c_vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def bvh_build_hlbvh(bounds6, max_prims_in_node=4):
    """BVHAccel's HLBVH build + flattening on the device: (nodes as a BVH_NODE array, order (n,) int32, stats dict)."""
    b = _f32(bounds6).reshape(-1, 6)
    n = len(b)
    nodes = np.zeros(max(2 * n, 1), dtype=BVH_NODE)
    order = np.zeros(max(n, 1), dtype=np.int32)
    n_nodes = c_i32(0)
    st = BvhBuildStats()
    rc = gpu_lib().iile_bvh_build_hlbvh(n, b.ctypes.data, int(max_prims_in_node), nodes.ctypes.data, ctypes.byref(n_nodes), order.ctypes.data,
                                        ctypes.byref(st))
    if rc != 0:
        raise RuntimeError(f"iile_bvh_build_hlbvh failed ({rc}): {gpu_lib().iile_last_error().decode()}")
    return nodes[:n_nodes.value].copy(), order[:n].copy(), {k: getattr(st, k) for k, _ in BvhBuildStats._fields_}


def bvh_pack_probe(nodes):
    """The traversal records iile_scene_create packs on the device for a flattened tree: (wide (ni, 16) float32,
    wide4 (ni, 32) float32, nested bool)."""
    nodes = np.ascontiguousarray(nodes, dtype=BVH_NODE)
    ni = int((nodes["nprims"] == 0).sum())
    wide = np.zeros((max(ni, 1), 16), np.float32)
    wide4 = np.zeros((max(ni, 1), 32), np.float32)
    nested = c_i32(0)
    rc = gpu_lib().iile_bvh_pack_probe(len(nodes), nodes.ctypes.data, ni, wide.ctypes.data, wide4.ctypes.data, ctypes.byref(nested))
    if rc != 0:
        raise RuntimeError(f"iile_bvh_pack_probe failed ({rc}): {gpu_lib().iile_last_error().decode()}")
    return wide[:ni], wide4[:ni], bool(nested.value)


def trig_probe(x):
    x = _f32(x)
    out = np.empty((len(x), 3), np.float32)
    rc = gpu_lib().iile_trig_probe(len(x), x.ctypes.data, out.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"iile_trig_probe failed ({rc}): {gpu_lib().iile_last_error().decode()}")
    return out


def device_count():
    return int(gpu_lib().iile_device_count())
