"""Scene description file (next-row scope, the reference's "TODO: Load scene from file", kernel.cu:261): the loader
builds exactly what the programmatic path builds, and malformed files are reported with file:line messages."""
import ctypes as C
import os

import numpy as np
import pytest

from gpupathtracer_amd import lib, scenes
from gpupathtracer_amd import types as T
from oracle_lib import oracle_render

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def _programmatic():
    tris = lib.load_obj(os.path.join(DATA, "quad_mixed.obj"))
    red = scenes.make_bxdf(T.BXDF_DIFFUSE, albedo=(1, 0, 0))
    light = scenes.make_bxdf(T.BXDF_EMITTER, emissive=(1, 1, 1), intensity=2.0)
    s = scenes.Scene()
    s.add_mesh(tris, (0, 0, 0), (0, 30, 10), (2, 2, 2), red)
    s.add_plane((0, -2.5, 0), (90, 0, 0), (9, 9, 9), red)
    s.add_plane((0, 3, 0), (90, 0, 0), (4, 4, 4), light)
    return s.finalize()


def test_scene_file_equals_programmatic_scene():
    sf = lib.SceneFile(os.path.join(DATA, "box.scene"))
    ref = _programmatic()
    assert len(sf) == len(ref) == 3 and sf.triangle_count == ref.triangle_count == 5
    for i in range(3):
        a, b = sf.geometries[i], ref.geometries[i]
        assert a.m_geometryType == b.m_geometryType
        assert bytes(a.m_modelMatrix) == bytes(b.m_modelMatrix) and bytes(a.m_inverseModelMatrix) == bytes(b.m_inverseModelMatrix)
        assert bytes(a.m_bxdf.contents) == bytes(b.m_bxdf.contents)
        assert a.m_numberOfTriangles == b.m_numberOfTriangles
        if a.m_numberOfTriangles:
            assert np.array_equal(T.triangles_to_array(a.m_triangles, 5), T.triangles_to_array(b.m_triangles, 5))
    cam = sf.camera(64, 48)
    exp = scenes.posed_camera(64, 48, position=(0.5, 0.5, 6.0), yaw=-90.0, pitch=0.0)
    exp.m_fov, exp.m_farClip = 60.0, 500.0
    assert bytes(cam) == bytes(exp)
    # same pixels through the oracle
    params = lib.render_params(64, 48, 3, 2)
    a8, arad = oracle_render(sf, cam, params, threads=4)
    b8, brad = oracle_render(ref, exp, params, threads=4)
    assert np.array_equal(a8, b8) and np.array_equal(arad, brad) and arad.max() > 0
    info = lib.scene_info(sf)
    assert info.valid == 1 and info.num_planes == 2 and info.num_meshes == 1
    sf.close()


@pytest.mark.parametrize("text,needle", [
    ("plane position 0 0 0 bxdf nope\n", "not defined"),
    ("bxdf a diffuse albedo 1 0\n", "albedo needs 3"),
    ("bxdf a shiny\n", "unknown bxdf type"),
    ("bxdf a diffuse\nplane position 1 2\n", "position needs 3"),
    ("bxdf a diffuse\nplane bxdf a spin 3\n", "unknown geometry key"),
    ("bxdf a diffuse\nmesh missing.obj bxdf a\n", "cannot open"),
    ("frobnicate\n", "unknown statement"),
    ("# only a comment\ncamera fov 50\n", "no geometries"),
])
def test_scene_file_errors(tmp_path, text, needle):
    p = tmp_path / "bad.scene"
    p.write_text(text)
    with pytest.raises(lib.FireflyError) as e:
        lib.SceneFile(str(p))
    assert e.value.status == T.FF_ERR_IO and needle in e.value.message, e.value.message


def test_scene_file_missing():
    with pytest.raises(lib.FireflyError) as e:
        lib.SceneFile("/nonexistent/x.scene")
    assert e.value.status == T.FF_ERR_IO


def test_scene_file_mirror_bxdf(tmp_path):
    p = tmp_path / "mirror.scene"
    p.write_text("bxdf m mirror specular 0.9 0.8 0.7\nbxdf l emitter color 1 1 1 intensity 3\n"
                 "plane position 0 0 -2 scale 4 4 4 bxdf m\nplane position 0 2 0 rotation 90 0 0 bxdf l\n")
    sf = lib.SceneFile(str(p))
    b = sf.geometries[0].m_bxdf.contents
    assert b.m_type == T.BXDF_MIRROR
    assert (b.m_specularColor.x, b.m_specularColor.y, b.m_specularColor.z) == pytest.approx((0.9, 0.8, 0.7))
    sf.close()


def test_scene_file_sphere(tmp_path):
    p = tmp_path / "s.scene"
    p.write_text("bxdf d diffuse albedo 0.5 0.5 0.5\nsphere radius 0.75 position 1 2 3 scale 1 2 1 bxdf d\n")
    sf = lib.SceneFile(str(p))
    g = sf.geometries[0]
    assert g.m_geometryType == T.GEOM_SPHERE and g.m_sphereRadius == 0.75
    assert (g.m_position.x, g.m_position.y, g.m_position.z) == (1.0, 2.0, 3.0)
    assert lib.scene_info(sf).valid == 1
    sf.close()
    p.write_text("bxdf d diffuse\nsphere position 0 0 0 bxdf d\n")
    with pytest.raises(lib.FireflyError) as e:
        lib.SceneFile(str(p))
    assert "sphere needs a radius" in e.value.message


def test_scene_file_glass_bxdf(tmp_path):
    p = tmp_path / "g.scene"
    p.write_text("bxdf g glass specular 1 1 1 transmittance 0.9 0.95 1 ior 1.5\nsphere radius 1 bxdf g\n")
    sf = lib.SceneFile(str(p))
    b = sf.geometries[0].m_bxdf.contents
    assert b.m_type == T.BXDF_GLASS and b.m_refractiveIndex == 1.5
    assert (b.m_transmittanceColor.x, b.m_transmittanceColor.y, b.m_transmittanceColor.z) == pytest.approx((0.9, 0.95, 1.0))
    sf.close()
