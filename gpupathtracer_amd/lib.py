"""ctypes binding of the C ABI in include/firefly/ff_api.h (libfirefly_hip.so, built in-tree by
``python -c 'import __graft_entry__ as g; g.build()'`` or ``make -C gpupathtracer_amd/csrc``).

There is no CPU fallback: if the HIP library is missing, importing this module's ``load()`` raises.
"""
import ctypes as C
import os

import numpy as np

from . import types as T

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FF_LIB_PATH") or os.path.join(_HERE, "libfirefly_hip.so")  # FF_LIB_PATH: another build of the same library (A/B runs)

# every exported symbol declared in include/firefly/ff_api.h
EXPORTS = [
    "ff_create", "ff_destroy", "ff_last_error", "ff_version", "ff_set_stream",
    "ff_geometry_init", "ff_bxdf_init", "ff_camera_init_default", "ff_camera_update_basis", "ff_camera_ray_matrix",
    "ff_render_tile", "ff_upload_scene", "ff_set_builder", "ff_update_transforms", "ff_update_mesh", "ff_build_stats", "ff_debug_download_bvh", "ff_debug_download_bvh4",
    "ff_scene_info", "ff_debug_wall_table", "ff_debug_wall_entries", "ff_render", "ff_render_strips", "ff_strips_local_rows", "ff_deinterleave_strips",
    "ff_intersect_rays", "ff_register_gl_pbo", "ff_unregister_gl_pbo", "ff_render_to_pbo",
    "ff_render_progressive", "ff_render_to_pbo_progressive", "ff_save_ppm",
    "ff_set_collect_stats", "ff_stats", "ff_debug_kernel_name", "ff_debug_counters", "ff_debug_timeline", "ff_debug_check_ieee", "ff_debug_reload_switches", "ff_load_obj", "ff_free_triangles",
    "ff_scene_file_load", "ff_scene_file_geometries", "ff_scene_file_camera", "ff_scene_file_free",
    "ff_dist_unique_id", "ff_dist_init", "ff_dist_available", "ff_dist_shutdown", "ff_dist_strip_rows", "ff_dist_strip_rows_for", "ff_dist_part_bytes", "ff_render_distributed", "ff_debug_dist_fail_rank",
    "ff_multi_create", "ff_multi_destroy", "ff_multi_count", "ff_multi_state", "ff_multi_uses_rccl", "ff_multi_upload_scene",
    "ff_multi_render", "ff_multi_render_to_pbo", "ff_multi_stats",
]
DIST_ID_BYTES = 128

_lib = None


class FireflyError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"firefly status {status}: {message}")
        self.status = status
        self.message = message


class _Tolerant:
    """Prototype declarations against a library that may lack some symbols."""

    class _Missing:
        argtypes = restype = None

    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        try:
            return getattr(self._lib, name)
        except AttributeError:
            return _Tolerant._Missing()


def load():
    """Load libfirefly_hip.so (once) and declare prototypes. Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension has not been built (run __graft_entry__.build()); "
            "there is no CPU fallback for the trace path")
    real = C.CDLL(LIB_PATH)
    # (an older build loaded through FF_LIB_PATH for an A/B run may lack the newest entry points: skip their prototypes)
    lib = _Tolerant(real) if os.environ.get("FF_LIB_PATH") else real
    vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
    P = C.POINTER
    lib.ff_create.argtypes = [P(vp), i32]
    lib.ff_destroy.argtypes = [vp]
    lib.ff_last_error.restype = C.c_char_p
    lib.ff_version.restype = i32
    lib.ff_set_stream.argtypes = [vp, vp]
    lib.ff_geometry_init.argtypes = [P(T.FfGeometry), i32, T.FfVec3, T.FfVec3, T.FfVec3, P(T.FfTriangle), i32, f32]
    lib.ff_geometry_init.restype = None
    lib.ff_bxdf_init.argtypes = [P(T.FfBXDF)]
    lib.ff_bxdf_init.restype = None
    lib.ff_camera_init_default.argtypes = [P(T.FfCamera), i32, i32]
    lib.ff_camera_init_default.restype = None
    lib.ff_camera_update_basis.argtypes = [P(T.FfCamera)]
    lib.ff_camera_update_basis.restype = None
    lib.ff_camera_ray_matrix.argtypes = [P(T.FfCamera), P(T.FfMat4)]
    lib.ff_camera_ray_matrix.restype = None
    lib.ff_upload_scene.argtypes = [vp, P(T.FfGeometry), i32]
    lib.ff_set_builder.argtypes = [vp, i32]
    lib.ff_update_transforms.argtypes = [vp, P(T.FfGeometry), i32]
    lib.ff_update_mesh.argtypes = [vp, i32, P(T.FfTriangle), i32, i32]
    lib.ff_build_stats.argtypes = [vp, P(T.FfBuildStats)]
    lib.ff_debug_download_bvh.argtypes = [vp, vp, i32, P(i32), vp, i32, P(i32), P(i32), i32]
    lib.ff_debug_download_bvh4.argtypes = [vp, vp, i32, P(i32), P(i32), i32]
    lib.ff_scene_info.argtypes = [P(T.FfGeometry), i32, P(T.FfSceneInfo)]
    lib.ff_debug_wall_table.argtypes = [P(T.FfGeometry), i32, P(C.c_float), i32]
    lib.ff_debug_wall_entries.argtypes = [P(T.FfGeometry), i32]
    lib.ff_render.argtypes = [vp, P(T.FfCamera), P(T.FfRenderParams), vp, i32, vp, i32]
    lib.ff_render_tile.argtypes = [vp, P(T.FfCamera), P(T.FfRenderParams), i32, i32, i32, i32, vp, i32, vp, i32]
    lib.ff_render_strips.argtypes = [vp, P(T.FfCamera), P(T.FfRenderParams), i32, i32, i32, vp, i32, vp, i32, P(i32)]
    lib.ff_strips_local_rows.argtypes = [i32, i32, i32, i32]
    lib.ff_deinterleave_strips.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32]
    lib.ff_intersect_rays.argtypes = [vp, P(T.FfRay), i32, P(T.FfIntersect), i32]
    lib.ff_register_gl_pbo.argtypes = [vp, C.c_uint, i32, i32]
    lib.ff_unregister_gl_pbo.argtypes = [vp]
    lib.ff_render_to_pbo.argtypes = [vp, P(T.FfCamera), P(T.FfRenderParams)]
    lib.ff_render_progressive.argtypes = [vp, P(T.FfCamera), P(T.FfRenderParams), i32, vp, i32, vp, i32]
    lib.ff_render_to_pbo_progressive.argtypes = [vp, P(T.FfCamera), P(T.FfRenderParams), i32]
    lib.ff_save_ppm.argtypes = [C.c_char_p, vp, i32, i32]
    lib.ff_set_collect_stats.argtypes = [vp, i32]
    lib.ff_stats.argtypes = [vp, P(T.FfStats)]
    lib.ff_debug_counters.argtypes = [vp, P(C.c_ulonglong)]
    lib.ff_debug_timeline.argtypes = [vp, P(C.c_uint), P(C.c_int)]
    lib.ff_debug_kernel_name.argtypes = [vp]
    lib.ff_debug_kernel_name.restype = C.c_char_p
    lib.ff_debug_check_ieee.argtypes = [vp, P(C.c_ulonglong)]
    lib.ff_debug_reload_switches.argtypes = [vp]
    lib.ff_load_obj.argtypes = [C.c_char_p, P(P(T.FfTriangle)), P(i32)]
    lib.ff_free_triangles.argtypes = [P(T.FfTriangle)]
    lib.ff_free_triangles.restype = None
    lib.ff_scene_file_load.argtypes = [C.c_char_p, P(vp)]
    lib.ff_scene_file_geometries.argtypes = [vp, P(i32)]
    lib.ff_scene_file_geometries.restype = P(T.FfGeometry)
    lib.ff_scene_file_camera.argtypes = [vp, i32, i32, P(T.FfCamera)]
    lib.ff_scene_file_free.argtypes = [vp]
    lib.ff_scene_file_free.restype = None
    lib.ff_dist_unique_id.argtypes = [vp, i32]
    lib.ff_dist_init.argtypes = [vp, i32, i32, vp, i32]
    lib.ff_dist_shutdown.argtypes = [vp]
    lib.ff_debug_dist_fail_rank.argtypes = [vp, C.c_int]
    lib.ff_dist_available.argtypes = []
    lib.ff_dist_strip_rows.argtypes = [i32]
    lib.ff_dist_strip_rows_for.argtypes = [i32, i32]
    lib.ff_dist_part_bytes.argtypes = [i32, i32, i32, i32, i32, P(C.c_longlong)]
    lib.ff_dist_part_bytes.restype = C.c_longlong
    lib.ff_render_distributed.argtypes = [vp, P(T.FfCamera), P(T.FfRenderParams), i32, vp, i32, vp, i32]
    lib.ff_multi_create.argtypes = [P(vp), P(i32), i32]
    lib.ff_multi_destroy.argtypes = [vp]
    lib.ff_multi_count.argtypes = [vp]
    lib.ff_multi_state.argtypes = [vp, i32]
    lib.ff_multi_state.restype = vp
    lib.ff_multi_uses_rccl.argtypes = [vp]
    lib.ff_multi_upload_scene.argtypes = [vp, P(T.FfGeometry), i32]
    lib.ff_multi_render.argtypes = [vp, P(T.FfCamera), P(T.FfRenderParams), i32, vp, i32, vp, i32]
    lib.ff_multi_render_to_pbo.argtypes = [vp, P(T.FfCamera), P(T.FfRenderParams), i32]
    lib.ff_multi_stats.argtypes = [vp, P(T.FfStats)]
    _lib = real
    return real


def check(status):
    if status != T.FF_OK:
        raise FireflyError(status, load().ff_last_error().decode("utf-8", "replace"))


def load_obj(path):
    """LoadMesh (utilities.h:781-840) -> float32 [n, 24] triangles."""
    lib = load()
    ptr = C.POINTER(T.FfTriangle)()
    n = C.c_int(0)
    check(lib.ff_load_obj(os.fsencode(path), C.byref(ptr), C.byref(n)))
    try:
        return T.triangles_to_array(ptr, n.value)
    finally:
        lib.ff_free_triangles(ptr)


def save_ppm(path, rgb8):
    """saveToPPM (utilities.h:842-856) for an [H, W, 3] uint8 frame."""
    a = np.ascontiguousarray(rgb8, dtype=np.uint8)
    check(load().ff_save_ppm(os.fsencode(path), a.ctypes.data, a.shape[1], a.shape[0]))


class SceneFile:
    """A scene description file (ff_scene_file_load): exposes `.geometries` / `len()` like scenes.Scene."""

    def __init__(self, path):
        self._lib = load()
        self._handle = C.c_void_p()
        check(self._lib.ff_scene_file_load(os.fsencode(path), C.byref(self._handle)))
        n = C.c_int(0)
        self.geometries = self._lib.ff_scene_file_geometries(self._handle, C.byref(n))
        self._count = n.value

    def __len__(self):
        return self._count

    @property
    def triangle_count(self):
        return int(sum(self.geometries[i].m_numberOfTriangles for i in range(self._count)
                       if self.geometries[i].m_geometryType == T.GEOM_TRIANGLEMESH))

    def camera(self, width, height):
        cam = T.FfCamera()
        check(self._lib.ff_scene_file_camera(self._handle, width, height, C.byref(cam)))
        return cam

    def close(self):
        if self._handle:
            self._lib.ff_scene_file_free(self._handle)
            self._handle = C.c_void_p()
            self.geometries = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def scene_info(scene):
    """Host-only dry run of the scene compiler (sizes, BVH shape, structural self-check)."""
    info = T.FfSceneInfo()
    check(load().ff_scene_info(scene.geometries, len(scene), C.byref(info)))
    return info


def render_params(width, height, bounces=1, spp=1, seed=1234, trace_mode=T.TRACE_BVH, shade_mode=T.SHADE_DIFFUSE_PATH,
                  grid_mode=T.GRID_FULL, spp_per_launch=0):
    return T.FfRenderParams(width, height, bounces, spp, seed, trace_mode, shade_mode, grid_mode, spp_per_launch)


class Tracer:
    """Owner of one FfState (one HIP device). Mirrors the call sequence of the reference's main():
    upload once (kernel.cu:268-298), then render per frame (kernel.cu:335-344)."""

    def __init__(self, device_id=0):
        self._lib = load()
        self._state = C.c_void_p()
        check(self._lib.ff_create(C.byref(self._state), device_id))
        self._scene_keepalive = None

    def close(self):
        if self._state:
            self._lib.ff_destroy(self._state)
            self._state = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream_ptr):
        check(self._lib.ff_set_stream(self._state, C.c_void_p(hip_stream_ptr)))

    def upload_scene(self, scene):
        """scene: gpupathtracer_amd.scenes.Scene"""
        check(self._lib.ff_upload_scene(self._state, scene.geometries, len(scene)))

    def set_builder(self, builder):
        """T.BUILD_HOST_SAH (default) or T.BUILD_GPU_LBVH for the following upload_scene calls."""
        check(self._lib.ff_set_builder(self._state, builder))

    def update_transforms(self, scene):
        """Same geometries as uploaded, new transforms / materials: rewrites the per-geometry records only."""
        check(self._lib.ff_update_transforms(self._state, scene.geometries, len(scene)))

    def update_mesh(self, geometry_index, triangles, mode=T.UPDATE_REFIT):
        """New vertices (float32 [n, 24], as load_obj returns) for one uploaded mesh: refit or rebuild its tree on the device."""
        buf = T.triangles_from_array(triangles)
        check(self._lib.ff_update_mesh(self._state, geometry_index, buf, len(buf), mode))

    def build_stats(self):
        st = T.FfBuildStats()
        check(self._lib.ff_build_stats(self._state, C.byref(st)))
        return st

    def download_bvh(self, num_geometries):
        """(nodes, triangle records, mesh table) of the compiled scene (tests).  mesh table: int32 [num_geometries, 5] =
        bvh_root, node_count, tri_first, tri_count, depth per uploaded geometry."""
        nn, nt = C.c_int(0), C.c_int(0)
        check(self._lib.ff_debug_download_bvh(self._state, None, 0, C.byref(nn), None, 0, C.byref(nt), None, 0))
        nodes = np.zeros(max(nn.value, 1), dtype=T.BVH_NODE_DTYPE)
        tris = np.zeros(max(nt.value, 1), dtype=T.TRI_RECORD_DTYPE)
        table = np.full((num_geometries, 5), -1, dtype=np.int32)
        check(self._lib.ff_debug_download_bvh(self._state, nodes.ctypes.data, nn.value, C.byref(nn), tris.ctypes.data, nt.value, C.byref(nt),
                                              table.ctypes.data_as(C.POINTER(C.c_int)), num_geometries))
        return nodes[:nn.value], tris[:nt.value], table

    def download_bvh4(self, num_geometries):
        """(4-wide nodes [capacity] with fields mn[3][4], mx[3][4], link[4]; table int32 [num_geometries, 6] = first node, node
        count, depth, first LDS slot, nodes in LDS, LDS node slots)."""
        cap = C.c_int(0)
        check(self._lib.ff_debug_download_bvh4(self._state, None, 0, C.byref(cap), None, 0))
        nodes = np.zeros(max(cap.value, 1), dtype=T.BVH4_NODE_DTYPE)
        table = np.full((num_geometries, 6), -1, dtype=np.int32)
        check(self._lib.ff_debug_download_bvh4(self._state, nodes.ctypes.data, cap.value, C.byref(cap),
                                               table.ctypes.data_as(C.POINTER(C.c_int)), num_geometries))
        return nodes[:cap.value], table

    def set_collect_stats(self, on):
        check(self._lib.ff_set_collect_stats(self._state, 1 if on else 0))

    def stats(self):
        st = T.FfStats()
        check(self._lib.ff_stats(self._state, C.byref(st)))
        return st

    def kernel_name(self):
        """Trace-kernel instantiation of the last frame, as rocprofv3 names it."""
        return self._lib.ff_debug_kernel_name(self._state).decode()

    def check_ieee(self):
        """(reciprocal mismatches, square-root mismatches) of the kernels' lean IEEE sequences over all 2^32 floats."""
        buf = (C.c_ulonglong * 2)()
        check(self._lib.ff_debug_check_ieee(self._state, buf))
        return int(buf[0]), int(buf[1])

    def reload_switches(self):
        """Re-read the FF_* experiment switches from the environment (they are read once, at ff_create)."""
        check(self._lib.ff_debug_reload_switches(self._state))

    def debug_counters(self):
        buf = (C.c_ulonglong * 32)()
        check(self._lib.ff_debug_counters(self._state, buf))
        return list(buf)

    def debug_timeline(self):
        """(bucket_us, counts[1024]): rays completed per wall-clock bucket of the last instrumented launch (FF_DEBUG_TIMELINE_US)."""
        buf = (C.c_uint * 1024)()
        us = C.c_int(0)
        check(self._lib.ff_debug_timeline(self._state, buf, C.byref(us)))
        return int(us.value), np.frombuffer(buf, dtype=np.uint32).copy()

    def render(self, camera, params, want_rgb8=True, want_radiance=True):
        """Headless frame to host numpy arrays: (rgb8 [H,W,3] uint8, radiance [H,W,3] float32)."""
        h, w = params.height, params.width
        rgb8 = np.zeros((h, w, 3), dtype=np.uint8) if want_rgb8 else None
        rad = np.zeros((h, w, 3), dtype=np.float32) if want_radiance else None
        check(self._lib.ff_render(self._state, C.byref(camera), C.byref(params),
                                  rgb8.ctypes.data if want_rgb8 else None, 0,
                                  rad.ctypes.data if want_radiance else None, 0))
        return rgb8, rad

    def render_progressive(self, camera, params, frame_index):
        """Frame `frame_index` of a progressive sequence -> (rgb8, radiance) of the mean over frames 0..frame_index."""
        h, w = params.height, params.width
        rgb8 = np.zeros((h, w, 3), dtype=np.uint8)
        rad = np.zeros((h, w, 3), dtype=np.float32)
        check(self._lib.ff_render_progressive(self._state, C.byref(camera), C.byref(params), frame_index, rgb8.ctypes.data, 0, rad.ctypes.data, 0))
        return rgb8, rad

    def render_tile(self, camera, params, x0, y0, w, h):
        """Tile [x0, x0+w) x [y0, y0+h) of the frame -> (rgb8 [h,w,3], radiance [h,w,3])."""
        rgb8 = np.zeros((h, w, 3), dtype=np.uint8)
        rad = np.zeros((h, w, 3), dtype=np.float32)
        check(self._lib.ff_render_tile(self._state, C.byref(camera), C.byref(params), x0, y0, w, h, rgb8.ctypes.data, 0, rad.ctypes.data, 0))
        return rgb8, rad

    def render_device(self, camera, params, rgb8_ptr=None, radiance_ptr=None):
        """Headless frame into caller-owned DEVICE buffers (raw pointers, e.g. torch tensor.data_ptr())."""
        check(self._lib.ff_render(self._state, C.byref(camera), C.byref(params),
                                  C.c_void_p(rgb8_ptr) if rgb8_ptr else None, 1,
                                  C.c_void_p(radiance_ptr) if radiance_ptr else None, 1))

    def strips_local_rows(self, height, strip_rows, part, num_parts):
        return self._lib.ff_strips_local_rows(height, strip_rows, part, num_parts)

    def render_strips(self, camera, params, strip_rows, part, num_parts, want_rgb8=True, want_radiance=True):
        rows = self._lib.ff_strips_local_rows(params.height, strip_rows, part, num_parts)
        w = params.width
        rgb8 = np.zeros((rows, w, 3), dtype=np.uint8) if want_rgb8 else None
        rad = np.zeros((rows, w, 3), dtype=np.float32) if want_radiance else None
        n = C.c_int(0)
        check(self._lib.ff_render_strips(self._state, C.byref(camera), C.byref(params), strip_rows, part, num_parts,
                                         rgb8.ctypes.data if want_rgb8 and rows else None, 0,
                                         rad.ctypes.data if want_radiance and rows else None, 0, C.byref(n)))
        assert n.value == rows
        return rgb8, rad

    def render_strips_device(self, camera, params, strip_rows, part, num_parts, rgb8_ptr=None, radiance_ptr=None):
        n = C.c_int(0)
        check(self._lib.ff_render_strips(self._state, C.byref(camera), C.byref(params), strip_rows, part, num_parts,
                                         C.c_void_p(rgb8_ptr) if rgb8_ptr else None, 1,
                                         C.c_void_p(radiance_ptr) if radiance_ptr else None, 1, C.byref(n)))
        return n.value

    def deinterleave_strips(self, src_ptr, dst_ptr, width, height, strip_rows, num_parts, elem_bytes):
        check(self._lib.ff_deinterleave_strips(self._state, C.c_void_p(src_ptr), C.c_void_p(dst_ptr), width, height,
                                               strip_rows, num_parts, elem_bytes))

    def intersect_rays(self, origins, directions, trace_mode=T.TRACE_BVH):
        """intersectRays (kernel.cu:127-176) for arrays of world-space rays -> structured result arrays."""
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        n = o.shape[0]
        rays = np.concatenate([o, d], axis=1).astype(np.float32)
        rbuf = (T.FfRay * n)()
        C.memmove(rbuf, rays.ctypes.data, rays.nbytes)
        out = (T.FfIntersect * n)()
        check(self._lib.ff_intersect_rays(self._state, rbuf, n, out, trace_mode))
        return intersect_array(out, n)


    # ---- multi-GPU, one process per GPU (ff_dist_*) ----

    def dist_init(self, rank, world_size, unique_id):
        """Join the job's RCCL communicator.  `unique_id`: the 128 bytes dist_unique_id() returned on rank 0."""
        buf = (C.c_char * DIST_ID_BYTES).from_buffer_copy(bytes(unique_id))
        check(self._lib.ff_dist_init(self._state, rank, world_size, buf, DIST_ID_BYTES))

    def dist_shutdown(self):
        check(self._lib.ff_dist_shutdown(self._state))

    def debug_dist_fail_rank(self, rank):
        """Tests: rank `rank` reports an injected local failure from the next distributed frame on (-1: off)."""
        check(self._lib.ff_debug_dist_fail_rank(self._state, rank))

    def render_distributed(self, camera, params, strip_rows=0, rank=0, want_rgb8=True, want_radiance=True):
        """One frame over all ranks; returns (rgb8, radiance) host arrays on rank 0, (None, None) elsewhere."""
        h, w = params.height, params.width
        root = rank == 0
        rgb8 = np.zeros((h, w, 3), dtype=np.uint8) if want_rgb8 and root else None
        rad = np.zeros((h, w, 3), dtype=np.float32) if want_radiance and root else None
        check(self._lib.ff_render_distributed(self._state, C.byref(camera), C.byref(params), strip_rows,
                                              rgb8.ctypes.data if rgb8 is not None else None, 0,
                                              rad.ctypes.data if rad is not None else None, 0))
        return rgb8, rad

    def render_distributed_device(self, camera, params, strip_rows=0, rgb8_ptr=None, radiance_ptr=None):
        """The same into caller-owned device buffers on rank 0 (raw pointers; ignored on other ranks)."""
        check(self._lib.ff_render_distributed(self._state, C.byref(camera), C.byref(params), strip_rows,
                                              C.c_void_p(rgb8_ptr) if rgb8_ptr else None, 1,
                                              C.c_void_p(radiance_ptr) if radiance_ptr else None, 1))


def dist_unique_id():
    """128-byte RCCL id for Tracer.dist_init (call on rank 0, hand to the other ranks)."""
    buf = (C.c_char * DIST_ID_BYTES)()
    check(load().ff_dist_unique_id(buf, DIST_ID_BYTES))
    return bytes(buf)


def dist_available():
    """True if this process can load RCCL (what dist_unique_id / Tracer.dist_init need)."""
    return load().ff_dist_available() == T.FF_OK


def dist_strip_rows(world_size, height=1080):
    return load().ff_dist_strip_rows_for(height, world_size)


class MultiTracer:
    """Several GPUs driven by ONE process (ff_multi_*): what a single-process viewer calls.  device_ids[0] gathers."""

    def __init__(self, device_ids):
        self._lib = load()
        self._handle = C.c_void_p()
        ids = (C.c_int * len(device_ids))(*device_ids)
        check(self._lib.ff_multi_create(C.byref(self._handle), ids, len(device_ids)))

    def close(self):
        if self._handle:
            self._lib.ff_multi_destroy(self._handle)
            self._handle = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __len__(self):
        return self._lib.ff_multi_count(self._handle)

    @property
    def uses_rccl(self):
        return bool(self._lib.ff_multi_uses_rccl(self._handle))

    def upload_scene(self, scene):
        check(self._lib.ff_multi_upload_scene(self._handle, scene.geometries, len(scene)))

    def render(self, camera, params, strip_rows=0, want_rgb8=True, want_radiance=True):
        h, w = params.height, params.width
        rgb8 = np.zeros((h, w, 3), dtype=np.uint8) if want_rgb8 else None
        rad = np.zeros((h, w, 3), dtype=np.float32) if want_radiance else None
        check(self._lib.ff_multi_render(self._handle, C.byref(camera), C.byref(params), strip_rows,
                                        rgb8.ctypes.data if want_rgb8 else None, 0, rad.ctypes.data if want_radiance else None, 0))
        return rgb8, rad

    def render_device(self, camera, params, strip_rows=0, rgb8_ptr=None, radiance_ptr=None):
        check(self._lib.ff_multi_render(self._handle, C.byref(camera), C.byref(params), strip_rows,
                                        C.c_void_p(rgb8_ptr) if rgb8_ptr else None, 1, C.c_void_p(radiance_ptr) if radiance_ptr else None, 1))

    def stats(self):
        st = T.FfStats()
        check(self._lib.ff_multi_stats(self._handle, C.byref(st)))
        return st


INTERSECT_DTYPE = np.dtype([("point", np.float32, 3), ("normal", np.float32, 3), ("t", np.float32), ("hit", np.uint8),
                            ("_pad", np.uint8, 3), ("geom", np.int32), ("tri", np.int32)])
assert INTERSECT_DTYPE.itemsize == 40


def intersect_array(buf, n):
    return np.frombuffer(bytes(buf), dtype=INTERSECT_DTYPE, count=n).copy()
