"""Named parity cases: scene preset + camera pose + render params.  Shared by tools/make_golden_frames.py (which
renders them with the CPU oracle into tests/golden/frames.npz) and by the GPU parity tests."""
from gpupathtracer_amd import lib, scenes
from gpupathtracer_amd import types as T

POSES = {
    "default": dict(position=(0.0, 0.0, 15.0), yaw=-90.0, pitch=0.0),          # kernel.cu:312-321
    "oblique": dict(position=(7.0, 3.0, 9.0), yaw=-128.0, pitch=-14.0),        # mesh visible past the z=+2.5 plane
    "inside": dict(position=(0.0, 0.0, 2.2), yaw=-90.0, pitch=0.0),            # inside the four planes
}


def _dbg(w, h, grid=T.GRID_FULL):
    return lib.render_params(w, h, 1, 1, 1234, T.TRACE_BVH, T.SHADE_NORMAL_DEBUG, grid, 0)


def _path(w, h, bounces, spp, seed=1234, shade=T.SHADE_DIFFUSE_PATH):
    return lib.render_params(w, h, bounces, spp, seed, T.TRACE_BVH, shade, T.GRID_FULL, 0)


# name -> (scene builder, pose, params builder)
CASES = {
    # BASELINE config #1: the reference's scene with cube.obj, 256x256, primary hit + normal shade
    "c1_cube_256": (lambda: scenes.reference_scene(scenes.load_mesh("cube")), "default", lambda: _dbg(256, 256)),
    "ref_cube_256_oblique": (lambda: scenes.reference_scene(scenes.load_mesh("cube")), "oblique", lambda: _dbg(256, 256)),
    "ref_wahoo_256_default": (lambda: scenes.reference_scene(scenes.load_mesh("wahoo")), "default", lambda: _dbg(256, 256)),
    "ref_wahoo_256_oblique": (lambda: scenes.reference_scene(scenes.load_mesh("wahoo")), "oblique", lambda: _dbg(256, 256)),
    "ref_rocketman_256_oblique": (lambda: scenes.reference_scene(scenes.load_mesh("rocketman")), "oblique", lambda: _dbg(256, 256)),
    "ref_rocketman_256_inside": (lambda: scenes.reference_scene(scenes.load_mesh("rocketman")), "inside", lambda: _dbg(256, 256)),
    # the shipped first frame (kernel.cu:229-258, 800x800): 52441 x (0,0,51) + 1 x (34,129,216), SURVEY.md §8c
    "ref_rocketman_800_default": (lambda: scenes.reference_scene(scenes.load_mesh("rocketman")), "default", lambda: _dbg(800, 800)),
    # non-square, height not a multiple of 16: the reference's floor-division grid leaves the last rows untraced
    "ref_sphere_200x150_floorgrid": (lambda: scenes.reference_scene(scenes.load_mesh("sphere")), "oblique",
                                     lambda: _dbg(200, 150, T.GRID_REFERENCE_FLOOR)),
    # build-defined integrator
    "path_cornell_96x64_b4_s4": (scenes.cornell_wahoo_scene, "default", lambda: _path(96, 64, 4, 4)),
    "path_cornell_64x48_b8_s2_seed7": (scenes.cornell_wahoo_scene, "inside", lambda: _path(64, 48, 8, 2, seed=7)),
    "path_blooper_64x64_b3_s3": (scenes.blooper_scene, "oblique", lambda: _path(64, 64, 3, 3)),
    "path_cornell_40x30_b1_s1": (scenes.cornell_wahoo_scene, "default", lambda: _path(40, 30, 1, 1)),
    # BXDFTyp::MIRROR (build-defined): mirror back wall + mirror cube, long specular chains
    "path_mirror_80x60_b6_s3": (scenes.cornell_mirror_scene, "inside", lambda: _path(80, 60, 6, 3, seed=77)),
    # SPHERE geometry (build-defined): diffuse, non-uniformly scaled and mirror spheres; also through the reference's shade
    "path_spheres_80x60_b5_s3": (scenes.cornell_spheres_scene, "inside", lambda: _path(80, 60, 5, 3, seed=5)),
    "dbg_spheres_96x72": (scenes.cornell_spheres_scene, "inside", lambda: _dbg(96, 72)),
    # BXDFTyp::GLASS (build-defined): glass sphere (enter, leave, total internal reflection), glass pane, mirror sphere
    "path_glass_80x60_b8_s4": (scenes.cornell_glass_scene, "inside", lambda: _path(80, 60, 8, 4, seed=31)),
    # interpolated vertex normals (build-defined use of utilities.h:163-170 and the barycentrics of kernel.cu:80-81)
    "smooth_cornell_96x64_b4_s4": (scenes.cornell_wahoo_scene, "inside", lambda: _path(96, 64, 4, 4, seed=9, shade=T.SHADE_DIFFUSE_PATH_SMOOTH)),
    "smooth_blooper_64x64_b3_s3": (scenes.blooper_scene, "oblique", lambda: _path(64, 64, 3, 3, shade=T.SHADE_DIFFUSE_PATH_SMOOTH)),
}


def build_case(name):
    scene_fn, pose, params_fn = CASES[name]
    params = params_fn()
    scene = scene_fn()
    cam = scenes.posed_camera(params.width, params.height, **POSES[pose])
    return scene, cam, params
