"""Scene presets for the BASELINE configs built through the library's own host helpers.

The reference has exactly one hard-coded scene (kernel.cu:227-259); `reference_scene` reproduces it.  The other
presets are build-defined (the reference has no scene file or second scene) and use the reference's literals where
they exist (camera kernel.cu:312-321, BXDFs kernel.cu:237-244).
"""
import ctypes as C
import os

import numpy as np

from . import lib as L
from . import types as T

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MESH_DIR = os.path.join(_ROOT, "assets", "meshes")


def load_fftri(path):
    """Read a flat triangle file written by oracle/ref_tinyobj_dump.cpp: 'FFTR', n, n*24 float32."""
    with open(path, "rb") as f:
        hdr = np.frombuffer(f.read(8), dtype=np.int32)
        if hdr.size != 2 or hdr[0] != 0x52544646:
            raise ValueError(f"{path}: not an FFTR triangle file")
        arr = np.frombuffer(f.read(), dtype=np.float32)
    return arr.reshape(int(hdr[1]), 24).copy()


def load_mesh(name):
    """Mesh fixture by name (cube, wahoo, rocketman, sphereBlender, sphere): the reference's sceneResources/*.obj
    flattened with LoadMesh semantics by the reference's own tinyobj (generator: oracle/ref_tinyobj_dump.cpp)."""
    return load_fftri(os.path.join(MESH_DIR, name + ".fftri"))


def subdivide_sphere(tris, levels):
    """Midpoint-subdivide every triangle `levels` times, re-projecting new vertices to the unit sphere (C4 preset).
    float32 arithmetic in a fixed order, so the result is deterministic.  Only positions are kept; uv/normals are 0."""
    v = np.ascontiguousarray(tris[:, :9], dtype=np.float32).reshape(-1, 3, 3)

    def unit(p):
        n = np.sqrt((p * p).sum(axis=1, dtype=np.float32), dtype=np.float32)
        return (p / n[:, None]).astype(np.float32)

    v = np.stack([unit(v[:, 0]), unit(v[:, 1]), unit(v[:, 2])], axis=1)
    for _ in range(levels):
        a, b, c = v[:, 0], v[:, 1], v[:, 2]
        ab = unit(((a + b) * np.float32(0.5)).astype(np.float32))
        bc = unit(((b + c) * np.float32(0.5)).astype(np.float32))
        ca = unit(((c + a) * np.float32(0.5)).astype(np.float32))
        v = np.concatenate([
            np.stack([a, ab, ca], axis=1), np.stack([ab, b, bc], axis=1),
            np.stack([ca, bc, c], axis=1), np.stack([ab, bc, ca], axis=1)], axis=0)
    out = np.zeros((v.shape[0], 24), dtype=np.float32)
    out[:, :9] = v.reshape(-1, 9)
    return out


def make_bxdf(kind, albedo=(-1, -1, -1), emissive=(-1, -1, -1), intensity=-1.0, specular=(-1, -1, -1),
              transmittance=(-1, -1, -1), ior=-1.0):
    b = T.FfBXDF()
    L.load().ff_bxdf_init(C.byref(b))  # utilities.h:81-88 defaults
    b.m_specularColor = T.FfVec3(*specular)
    b.m_transmittanceColor = T.FfVec3(*transmittance)
    b.m_refractiveIndex = ior
    b.m_type = kind
    b.m_albedo = T.FfVec3(*albedo)
    b.m_emissiveColor = T.FfVec3(*emissive)
    b.m_intensity = intensity
    return b


class Scene:
    """Host scene: an FfGeometry array plus the triangle/BXDF buffers its pointers refer to."""

    def __init__(self):
        self._specs = []
        self.geometries = None
        self._keep = []

    def add_mesh(self, triangles, position=(0, 0, 0), rotation=(0, 0, 0), scale=(1, 1, 1), bxdf=None):
        self._specs.append((T.GEOM_TRIANGLEMESH, position, rotation, scale, np.asarray(triangles, dtype=np.float32), bxdf))
        return self

    def add_plane(self, position=(0, 0, 0), rotation=(0, 0, 0), scale=(1, 1, 1), bxdf=None):
        self._specs.append((T.GEOM_PLANE, position, rotation, scale, None, bxdf))
        return self

    def add_sphere(self, radius, position=(0, 0, 0), rotation=(0, 0, 0), scale=(1, 1, 1), bxdf=None):
        """SPHERE geometry (utilities.h:193-195): object-space radius, placed by the same T*R*S transform."""
        self._specs.append((T.GEOM_SPHERE, position, rotation, scale, float(radius), bxdf))
        return self

    def finalize(self):
        lib = L.load()
        n = len(self._specs)
        self.geometries = (T.FfGeometry * n)()
        self._keep = []
        for i, (kind, pos, rot, scl, tris, bxdf) in enumerate(self._specs):
            tbuf, cnt, radius = None, 0, 0.0
            if kind == T.GEOM_SPHERE:
                radius = tris  # the spec's payload slot carries the radius
            elif tris is not None and len(tris):
                tbuf = T.triangles_from_array(tris)
                cnt = len(tbuf)
            lib.ff_geometry_init(C.byref(self.geometries[i]), kind, T.FfVec3(*pos), T.FfVec3(*rot), T.FfVec3(*scl),
                                 tbuf, cnt, radius)
            if bxdf is not None:
                self.geometries[i].m_bxdf = C.pointer(bxdf)
            self._keep.append((tbuf, bxdf))
        return self

    def __len__(self):
        return len(self._specs)

    @property
    def triangle_count(self):
        return int(sum(len(s[4]) for s in self._specs if s[0] == T.GEOM_TRIANGLEMESH and s[4] is not None))


def default_camera(width, height):
    """Camera literals of kernel.cu:311-322 for a width x height image (assigned un-swapped)."""
    cam = T.FfCamera()
    L.load().ff_camera_init_default(C.byref(cam), width, height)
    return cam


def posed_camera(width, height, position, yaw, pitch):
    cam = default_camera(width, height)
    cam.m_position = T.FfVec3(*position)
    cam.m_yaw = yaw
    cam.m_pitch = pitch
    L.load().ff_camera_update_basis(C.byref(cam))
    return cam


def reference_scene(mesh_triangles):
    """The reference's only scene, kernel.cu:229-258: a mesh at the origin rotated (0,90,180) and four scale-5 planes
    at z=+-2.5 and y=-+2.5, every geometry sharing one red diffuse BXDF (the emitter BXDF of :241-244 is never attached)."""
    red = make_bxdf(T.BXDF_DIFFUSE, albedo=(1, 0, 0))          # :237-239
    s = Scene()
    s.add_mesh(mesh_triangles, (0, 0, 0), (0, 90, 180), (1, 1, 1), red)  # :229
    s.add_plane((0, 0, 2.5), (0, 0, 0), (5, 5, 5), red)         # :231
    s.add_plane((0, 0, -2.5), (0, 0, 0), (5, 5, 5), red)        # :232
    s.add_plane((0, -2.5, 0), (90, 0, 0), (5, 5, 5), red)       # :233
    s.add_plane((0, 2.5, 0), (90, 0, 0), (5, 5, 5), red)        # :234
    return s.finalize()


def _box(scene):
    grey = make_bxdf(T.BXDF_DIFFUSE, albedo=(0.75, 0.75, 0.75))
    red = make_bxdf(T.BXDF_DIFFUSE, albedo=(0.75, 0.1, 0.1))
    green = make_bxdf(T.BXDF_DIFFUSE, albedo=(0.1, 0.75, 0.1))
    light = make_bxdf(T.BXDF_EMITTER, emissive=(1, 1, 1), intensity=2.0)  # kernel.cu:241-244
    scene.add_plane((0, 0, -2.5), (0, 0, 0), (5, 5, 5), grey)      # back
    scene.add_plane((0, -2.5, 0), (90, 0, 0), (5, 5, 5), grey)     # floor
    scene.add_plane((0, 2.5, 0), (90, 0, 0), (5, 5, 5), grey)      # ceiling
    scene.add_plane((-2.5, 0, 0), (0, 90, 0), (5, 5, 5), red)      # left
    scene.add_plane((2.5, 0, 0), (0, 90, 0), (5, 5, 5), green)     # right
    scene.add_plane((0, 2.49, 0), (90, 0, 0), (2, 2, 2), light)    # area light just under the ceiling
    return scene


def cornell_wahoo_scene(wahoo=None, cube=None):
    """C2 / C5: wahoo.obj (scale 0.28) and cube.obj in an open-front box of five planes with one emitter plane."""
    wahoo = load_mesh("wahoo") if wahoo is None else wahoo
    cube = load_mesh("cube") if cube is None else cube
    s = Scene()
    s.add_mesh(wahoo, (0, -2.4, 0), (0, 0, 0), (0.28, 0.28, 0.28), make_bxdf(T.BXDF_DIFFUSE, albedo=(1, 0, 0)))  # kernel.cu:239 albedo
    s.add_mesh(cube, (1.5, -2.0, 1.0), (0, 0, 0), (1, 1, 1), make_bxdf(T.BXDF_DIFFUSE, albedo=(0.75, 0.75, 0.75)))
    return _box(s).finalize()


def cornell_mirror_scene(sphere=None, cube=None):
    """The C2 box with a MIRROR back wall and a MIRROR cube next to a diffuse sphere (BXDFTyp::MIRROR, utilities.h:68-75)."""
    sphere = load_mesh("sphereBlender") if sphere is None else sphere
    cube = load_mesh("cube") if cube is None else cube
    s = Scene()
    s.add_mesh(sphere, (-0.9, -1.5, -0.3), (0, 0, 0), (1, 1, 1), make_bxdf(T.BXDF_DIFFUSE, albedo=(0.2, 0.4, 0.9)))
    s.add_mesh(cube, (1.2, -1.9, 0.4), (0, 30, 0), (1.2, 1.2, 1.2), make_bxdf(T.BXDF_MIRROR, specular=(0.95, 0.9, 0.8)))
    _box(s)
    kind, pos, rot, scl, tris, _ = s._specs[2]  # the back wall becomes a mirror
    s._specs[2] = (kind, pos, rot, scl, tris, make_bxdf(T.BXDF_MIRROR, specular=(0.9, 0.9, 0.9)))
    return s.finalize()


def cornell_spheres_scene(cube=None):
    """The C2 box with SPHERE geometries (utilities.h:193-195): a diffuse one, a squashed and rotated diffuse one, a mirror
    one, and the cube mesh between them."""
    cube = load_mesh("cube") if cube is None else cube
    s = Scene()
    s.add_mesh(cube, (0.1, -2.0, -0.6), (0, 20, 0), (1, 1, 1), make_bxdf(T.BXDF_DIFFUSE, albedo=(0.75, 0.75, 0.75)))
    _box(s)
    s.add_sphere(0.8, (-1.3, -1.7, 0.3), (0, 0, 0), (1, 1, 1), make_bxdf(T.BXDF_DIFFUSE, albedo=(0.2, 0.5, 0.9)))
    s.add_sphere(0.5, (1.4, -1.9, 0.8), (20, 0, 35), (1.6, 1.0, 1.2), make_bxdf(T.BXDF_DIFFUSE, albedo=(0.9, 0.6, 0.1)))
    s.add_sphere(0.6, (0.9, 0.4, -1.2), (0, 0, 0), (1, 1, 1), make_bxdf(T.BXDF_MIRROR, specular=(0.95, 0.95, 0.95)))
    return s.finalize()


def cornell_glass_scene(cube=None):
    """The C2 box with GLASS surfaces (BXDFTyp::GLASS, utilities.h:68-75): a glass sphere, a glass pane (a two-sided
    plane) in front of a diffuse cube, and a mirror sphere behind."""
    cube = load_mesh("cube") if cube is None else cube
    glass = dict(specular=(1.0, 1.0, 1.0), transmittance=(0.95, 0.98, 0.95), ior=1.5)
    s = Scene()
    s.add_mesh(cube, (1.1, -2.0, -0.8), (0, 25, 0), (1, 1, 1), make_bxdf(T.BXDF_DIFFUSE, albedo=(0.8, 0.3, 0.2)))
    _box(s)
    s.add_sphere(0.85, (-0.9, -1.65, 0.4), (0, 0, 0), (1, 1, 1), make_bxdf(T.BXDF_GLASS, **glass))
    s.add_plane((1.0, -1.2, 0.6), (0, 20, 0), (1.8, 2.4, 1), make_bxdf(T.BXDF_GLASS, specular=(1, 1, 1), transmittance=(0.8, 0.9, 1.0), ior=1.33))
    s.add_sphere(0.5, (-1.2, 0.9, -1.4), (0, 0, 0), (1, 1, 1), make_bxdf(T.BXDF_MIRROR, specular=(0.9, 0.9, 0.9)))
    return s.finalize()


def blooper_scene(rocketman=None, cube=None):
    """C3: kernel.cu:229-258 with rocketman + cube.obj and only the two +-y planes so the interior is visible; the +y
    plane carries the emitter BXDF the reference builds at :241-244 but never attaches (without it every path is black)."""
    rocketman = load_mesh("rocketman") if rocketman is None else rocketman
    cube = load_mesh("cube") if cube is None else cube
    red = make_bxdf(T.BXDF_DIFFUSE, albedo=(1, 0, 0))
    light = make_bxdf(T.BXDF_EMITTER, emissive=(1, 1, 1), intensity=2.0)
    s = Scene()
    s.add_mesh(rocketman, (0, 0, 0), (0, 90, 180), (1, 1, 1), red)
    s.add_mesh(cube, (1.5, -2.0, 0), (0, 0, 0), (1, 1, 1), red)
    s.add_plane((0, -2.5, 0), (90, 0, 0), (5, 5, 5), red)
    s.add_plane((0, 2.5, 0), (90, 0, 0), (5, 5, 5), light)
    return s.finalize()


def sphere_stress_scene(levels=5, base=None):
    """C4: sphereBlender.obj (960 tris) subdivided `levels` times (5 -> 983 040 tris), scale 2, inside the C2 box."""
    base = load_mesh("sphereBlender") if base is None else base
    s = Scene()
    s.add_mesh(subdivide_sphere(base, levels), (0, 0, 0), (0, 0, 0), (2, 2, 2), make_bxdf(T.BXDF_DIFFUSE, albedo=(0.75, 0.75, 0.75)))
    return _box(s).finalize()
