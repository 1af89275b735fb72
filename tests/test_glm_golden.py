"""The oracle's and the library's glm restatements against golden vectors produced by the reference's OWN vendored glm
0.9.9.7 headers (generator: oracle/ref_glm_vectors.cpp, compiled in place from /root/reference).  Bit-exact."""
import ctypes as C
import os

import numpy as np
import pytest

from gpupathtracer_amd import types as T
from oracle_lib import farr

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "glm_vectors.bin")


def load_vectors():
    with open(GOLDEN, "rb") as f:
        hdr = np.frombuffer(f.read(16), dtype=np.int32)
        assert hdr[0] == 0x564D4C47
        n, nin, nout = int(hdr[1]), int(hdr[2]), int(hdr[3])
        data = np.frombuffer(f.read(), dtype=np.float32).reshape(n, nin + nout)
    return data[:, :nin].copy(), data[:, nin:].copy()


IN, OUT = load_vectors()
# input slices
A, B, V, VA, VB, ANG, PERSP = slice(0, 16), slice(16, 32), slice(32, 36), slice(36, 39), slice(39, 42), 42, slice(43, 48)
# output slices
_o = 0


def _take(n):
    global _o
    s = slice(_o, _o + n)
    _o += n
    return s


O_AV, O_AB, O_INV, O_TR, O_TRANSL, O_ROT, O_SCALE, O_LOOK, O_PERSP = (_take(4), _take(16), _take(16), _take(16), _take(16), _take(16),
                                                                       _take(16), _take(16), _take(16))
O_NORM, O_CROSS, O_DOT, O_DIST, O_RAD, O_NRMMAT, O_NORM4, O_MODEL, O_IMODEL, O_CAM = (_take(3), _take(3), _take(1), _take(1), _take(1),
                                                                                        _take(4), _take(4), _take(16), _take(16), _take(16))
assert _o == OUT.shape[1]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def call_out(fn, n, *args):
    out = np.zeros(n, dtype=np.float32)
    fn(*args, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def test_fixture_shape():
    assert IN.shape == (256, 48) and OUT.shape == (256, 197)


def test_oracle_glm_bit_exact(oracle):
    for k in range(IN.shape[0]):
        i, o = IN[k], OUT[k]
        a, pa = farr(i[A]); b, pb = farr(i[B]); v, pv = farr(i[V]); va, pva = farr(i[VA]); vb, pvb = farr(i[VB])
        assert np.array_equal(bits(call_out(oracle.orc_mat4_mul_vec4, 4, pa, pv)), bits(o[O_AV]))
        assert np.array_equal(bits(call_out(oracle.orc_mat4_mul, 16, pa, pb)), bits(o[O_AB]))
        assert np.array_equal(bits(call_out(oracle.orc_mat4_inverse, 16, pa)), bits(o[O_INV]))
        assert np.array_equal(bits(call_out(oracle.orc_mat4_transpose, 16, pa)), bits(o[O_TR]))
        assert np.array_equal(bits(call_out(oracle.orc_translate, 16, pa, pva)), bits(o[O_TRANSL]))
        assert np.array_equal(bits(call_out(oracle.orc_rotate, 16, pa, C.c_float(i[ANG]), pvb)), bits(o[O_ROT]))
        assert np.array_equal(bits(call_out(oracle.orc_scale, 16, pa, pva)), bits(o[O_SCALE]))
        up, pup = farr(i[V][:3])
        assert np.array_equal(bits(call_out(oracle.orc_look_at_rh, 16, pva, pvb, pup)), bits(o[O_LOOK]))
        p = i[PERSP]
        persp = call_out(oracle.orc_perspective_fov_rh_no, 16, C.c_float(oracle.orc_radians(C.c_float(p[0]))), C.c_float(p[1]),
                         C.c_float(p[2]), C.c_float(p[3]), C.c_float(p[4]))
        assert np.array_equal(bits(persp), bits(o[O_PERSP]))
        assert np.array_equal(bits(call_out(oracle.orc_normalize3, 3, pva)), bits(o[O_NORM]))
        assert np.array_equal(bits(call_out(oracle.orc_cross3, 3, pva, pvb)), bits(o[O_CROSS]))
        assert bits(np.float32(oracle.orc_dot3(pva, pvb))) == bits(o[O_DOT])[0]
        assert bits(np.float32(oracle.orc_distance3(pva, pvb))) == bits(o[O_DIST])[0]
        assert bits(np.float32(oracle.orc_radians(C.c_float(i[ANG])))) == bits(o[O_RAD])[0]
        # inverse(transpose(A)) * vec4(a, 0)  (kernel.cu:117)
        tr = call_out(oracle.orc_mat4_transpose, 16, pa)
        inv = call_out(oracle.orc_mat4_inverse, 16, tr.ctypes.data_as(C.POINTER(C.c_float)))
        n4, pn4 = farr(np.append(i[VA], 0.0))
        assert np.array_equal(bits(call_out(oracle.orc_mat4_mul_vec4, 4, inv.ctypes.data_as(C.POINTER(C.c_float)), pn4)), bits(o[O_NRMMAT]))


def test_geometry_ctor_and_camera_matrix_bit_exact(oracle, ff):
    """Geometry::Geometry (utilities.h:176-213) and the camera matrix of kernel.cu:203 — oracle AND library vs glm."""
    lib = ff.load()
    for k in range(IN.shape[0]):
        i, o = IN[k], OUT[k]
        pos = T.FfVec3(*i[VA])
        rot = T.FfVec3(*(np.float32(90.0) * i[VB]))
        scl = T.FfVec3(*(np.abs(i[V][:3]) + np.float32(0.5)))
        for init in (oracle.orc_geometry_init, lib.ff_geometry_init):
            g = T.FfGeometry()
            init(C.byref(g), T.GEOM_PLANE, pos, rot, scl, None, 0, 0.0)
            assert np.array_equal(bits(np.frombuffer(bytes(g.m_modelMatrix), dtype=np.float32)), bits(o[O_MODEL]))
            assert np.array_equal(bits(np.frombuffer(bytes(g.m_inverseModelMatrix), dtype=np.float32)), bits(o[O_IMODEL]))
            assert g.m_normal.tuple() == (0.0, 0.0, 1.0)
        # camera: position a, forward normalize(b), worldUp (0,1,0) -> right/up as UpdateBasisAxis does (utilities.h:414-417)
        fwd = call_out(oracle.orc_normalize3, 3, farr(i[VB])[1])
        right = call_out(oracle.orc_normalize3, 3, farr(call_out(oracle.orc_cross3, 3, farr(fwd)[1], farr([0, 1, 0])[1]))[1])
        up = call_out(oracle.orc_normalize3, 3, farr(call_out(oracle.orc_cross3, 3, farr(right)[1], farr(fwd)[1]))[1])
        cam = T.FfCamera()
        cam.m_position = T.FfVec3(*i[VA]); cam.m_forward = T.FfVec3(*fwd); cam.m_up = T.FfVec3(*up)
        p = i[PERSP]
        cam.m_fov, cam.m_screenWidth, cam.m_screenHeight, cam.m_nearClip, cam.m_farClip = p[0], p[1], p[2], p[3], p[4]
        m_or = call_out(oracle.orc_camera_ray_matrix, 16, C.byref(cam))
        m_lib = T.FfMat4()
        lib.ff_camera_ray_matrix(C.byref(cam), C.byref(m_lib))
        assert np.array_equal(bits(m_or), bits(o[O_CAM]))
        assert np.array_equal(bits(np.frombuffer(bytes(m_lib), dtype=np.float32)), bits(o[O_CAM]))


def test_camera_basis_matches_between_oracle_and_library(oracle, ff):
    lib = ff.load()
    rng = np.random.default_rng(3)
    for _ in range(64):
        a, b = T.FfCamera(), T.FfCamera()
        oracle.orc_camera_init_default(C.byref(a), 1920, 1080)
        lib.ff_camera_init_default(C.byref(b), 1920, 1080)
        a.m_yaw = b.m_yaw = float(rng.uniform(-180, 180))
        a.m_pitch = b.m_pitch = float(rng.uniform(-89, 89))
        oracle.orc_camera_update_basis(C.byref(a))
        lib.ff_camera_update_basis(C.byref(b))
        assert bytes(a) == bytes(b)
    d = T.FfCamera()
    lib.ff_camera_init_default(C.byref(d), 800, 800)
    # kernel.cu:312-322 literals; yaw -90 / pitch 0 -> forward ~ (0,0,-1)
    assert d.m_position.tuple() == (0.0, 0.0, 15.0) and d.m_fov == 70.0 and d.m_farClip == 1000.0
    assert abs(d.m_forward.z + 1.0) < 1e-6 and abs(d.m_right.x - 1.0) < 1e-6 and abs(d.m_up.y - 1.0) < 1e-6
