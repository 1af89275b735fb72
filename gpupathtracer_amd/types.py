"""ctypes mirrors of include/firefly/ff_types.h (layout-compatible with the reference's structs,
PathTracer/FireflyEngine/utilities.h:57-66,77-88,148-171,219-233,257-267,269-291)."""
import ctypes as C

import numpy as np


class FfVec2(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class FfVec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def __init__(self, x=0.0, y=0.0, z=0.0):
        super().__init__(float(x), float(y), float(z))

    def tuple(self):
        return (self.x, self.y, self.z)


class FfMat4(C.Structure):
    _fields_ = [("m", C.c_float * 16)]

    def numpy(self):
        """4x4 array indexed [col][row] (glm storage order)."""
        return np.frombuffer(bytes(self), dtype=np.float32).reshape(4, 4).copy()


class FfBXDF(C.Structure):
    _fields_ = [
        ("m_type", C.c_int32),
        ("m_albedo", FfVec3),
        ("m_specularColor", FfVec3),
        ("m_refractiveIndex", C.c_float),
        ("m_emissiveColor", FfVec3),
        ("m_intensity", C.c_float),
        ("m_transmittanceColor", FfVec3),
    ]


class FfTriangle(C.Structure):
    _fields_ = [
        ("m_v0", FfVec3), ("m_v1", FfVec3), ("m_v2", FfVec3),
        ("m_uv0", FfVec2), ("m_uv1", FfVec2), ("m_uv2", FfVec2),
        ("m_n0", FfVec3), ("m_n1", FfVec3), ("m_n2", FfVec3),
    ]


class FfGeometry(C.Structure):
    _fields_ = [
        ("m_geometryType", C.c_int32),
        ("m_position", FfVec3),
        ("m_rotation", FfVec3),
        ("m_scale", FfVec3),
        ("m_modelMatrix", FfMat4),
        ("m_inverseModelMatrix", FfMat4),
        ("m_sphereRadius", C.c_float),
        ("m_normal", FfVec3),
        ("m_triangles", C.POINTER(FfTriangle)),
        ("m_numberOfTriangles", C.c_int32),
        ("m_bxdf", C.POINTER(FfBXDF)),
    ]


class FfRay(C.Structure):
    _fields_ = [("m_origin", FfVec3), ("m_direction", FfVec3)]


class FfIntersect(C.Structure):
    _fields_ = [
        ("m_intersectionPoint", FfVec3),
        ("m_normal", FfVec3),
        ("m_t", C.c_float),
        ("m_hit", C.c_uint8),
        ("_pad", C.c_uint8 * 3),
        ("geometryIndex", C.c_int32),
        ("triangleIndex", C.c_int32),
    ]


class FfCamera(C.Structure):
    _fields_ = [
        ("m_position", FfVec3), ("m_up", FfVec3), ("m_right", FfVec3), ("m_forward", FfVec3), ("m_worldUp", FfVec3),
        ("m_yaw", C.c_float), ("m_pitch", C.c_float),
        ("m_screenWidth", C.c_float), ("m_screenHeight", C.c_float),
        ("m_fov", C.c_float), ("m_nearClip", C.c_float), ("m_farClip", C.c_float),
        ("m_cameraMovementSpeed", C.c_float), ("m_cameraMouseSensitivity", C.c_float),
        ("m_cameraFirstMouseInput", C.c_uint8), ("_pad", C.c_uint8 * 3),
        ("m_xDelta", C.c_float), ("m_yDelta", C.c_float),
    ]


class FfRenderParams(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("bounces", C.c_int32), ("spp", C.c_int32),
        ("seed", C.c_uint64),
        ("trace_mode", C.c_int32), ("shade_mode", C.c_int32), ("grid_mode", C.c_int32), ("spp_per_launch", C.c_int32),
    ]


FF_STATS_TAIL_ITEMS = 1
FF_STATS_TAIL_SKIPPED_TOO_LARGE = 2


class FfStats(C.Structure):
    _fields_ = [
        ("rays_traced", C.c_uint64), ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64), ("planes_tested", C.c_uint64),
        ("kernel_ms", C.c_double), ("total_ms", C.c_double),
        ("kernel_launches", C.c_uint32), ("flags", C.c_uint32),
        ("scene_bytes_nodes", C.c_uint64), ("scene_bytes_tris", C.c_uint64), ("rays_answered", C.c_uint64), ("rays_cut_short", C.c_uint64),
    ]


class FfBuildStats(C.Structure):
    _fields_ = [
        ("builder", C.c_int32), ("bvh_nodes", C.c_int32), ("bvh_max_depth", C.c_int32), ("last_operation", C.c_int32),
        ("num_triangles", C.c_uint64), ("total_ms", C.c_double), ("copy_ms", C.c_double), ("build_ms", C.c_double),
    ]


BUILD_HOST_SAH, BUILD_GPU_LBVH, BUILD_GPU_PLOC = 0, 1, 2
UPDATE_REFIT, UPDATE_REBUILD = 0, 1

# device records as ff_debug_download_bvh returns them (gpupathtracer_amd/csrc/ff_internal.h)
import numpy as _np
BVH_NODE_DTYPE = _np.dtype([("lmin", _np.float32, 3), ("left", _np.int32), ("lmax", _np.float32, 3), ("right", _np.int32),
                            ("rmin", _np.float32, 3), ("pad0", _np.int32), ("rmax", _np.float32, 3), ("pad1", _np.int32)])
TRI_RECORD_DTYPE = _np.dtype([("v0", _np.float32, 3), ("orig_index", _np.int32), ("e1", _np.float32, 3), ("cull_margin", _np.float32),
                              ("e2", _np.float32, 3), ("pad1", _np.int32)])
assert BVH_NODE_DTYPE.itemsize == 64 and TRI_RECORD_DTYPE.itemsize == 48
# 4-wide traversal node (csrc/ff_internal.h Bvh4Node): box planes [axis][slot] and links (>= 0: node relative to the mesh's root, < 0: leaf)
BVH4_NODE_DTYPE = _np.dtype([("mn", _np.float32, (3, 4)), ("mx", _np.float32, (3, 4)), ("link", _np.int32, 4)])
assert BVH4_NODE_DTYPE.itemsize == 112
BVH4_EMPTY_LINK = 0x7fffffff


class FfSceneInfo(C.Structure):
    _fields_ = [
        ("num_geometries", C.c_int32), ("num_meshes", C.c_int32), ("num_planes", C.c_int32),
        ("bvh_nodes", C.c_int32), ("bvh_max_depth", C.c_int32), ("bvh_max_leaf", C.c_int32),
        ("lds_nodes", C.c_int32), ("lds_bytes", C.c_int32),
        ("num_triangles", C.c_uint64), ("device_bytes", C.c_uint64),
        ("valid", C.c_int32), ("bvh_child_area", C.c_float),
    ]


assert C.sizeof(FfBXDF) == 60
assert C.sizeof(FfTriangle) == 96
assert C.sizeof(FfGeometry) == 208
assert C.sizeof(FfRay) == 24
assert C.sizeof(FfIntersect) == 40
assert C.sizeof(FfCamera) == 108

# enums (ff_types.h)
BXDF_EMITTER, BXDF_DIFFUSE, BXDF_MIRROR, BXDF_GLASS, BXDF_COUNT = range(5)
GEOM_SPHERE, GEOM_PLANE, GEOM_TRIANGLEMESH = range(3)
TRACE_BRUTE_FORCE, TRACE_BVH = 0, 1
SHADE_NORMAL_DEBUG, SHADE_DIFFUSE_PATH, SHADE_DIFFUSE_PATH_SMOOTH = 0, 1, 2
GRID_FULL, GRID_REFERENCE_FLOOR = 0, 1

# status codes (ff_api.h)
FF_OK, FF_ERR_INVALID_ARG, FF_ERR_NO_DEVICE, FF_ERR_HIP, FF_ERR_NO_SCENE, FF_ERR_UNSUPPORTED, FF_ERR_GL_UNAVAILABLE, FF_ERR_IO, FF_ERR_OOM, FF_ERR_COMM = range(10)

TRIANGLE_DTYPE = np.dtype((np.float32, (24,)))


def triangles_from_array(arr):
    """float32 [n, 24] (FfTriangle field order) -> ctypes array of FfTriangle sharing a private copy."""
    a = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1, 24)
    buf = (FfTriangle * a.shape[0])()
    C.memmove(buf, a.ctypes.data, a.nbytes)
    return buf


def triangles_to_array(ptr, count):
    out = np.empty((count, 24), dtype=np.float32)
    C.memmove(out.ctypes.data, ptr, out.nbytes)
    return out
