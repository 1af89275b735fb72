"""GPU property tests at BASELINE sizes (run with `-m gpu`): size-independent properties the path offers, where the CPU
oracle is too slow to be the checker — BVH == brute force, strip union == whole frame, spp chunking == one launch,
determinism, ray-count bounds — plus edge cases (empty / one-triangle meshes, plane-only scenes, ragged sizes)."""
import numpy as np
import pytest

from gpupathtracer_amd import dist as ffdist  # noqa: F401
from gpupathtracer_amd import lib, scenes
from gpupathtracer_amd import types as T
from oracle_lib import oracle_render

pytestmark = pytest.mark.gpu


def _inside(w, h):
    return scenes.posed_camera(w, h, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)


def test_bvh_equals_brute_force_full_hd(tracer):
    """1920x1080, 8 bounces (BASELINE config #2 geometry), 1 spp: the BVH kernel and the brute-force reference loop
    produce the same bits and the same number of rays."""
    scene = scenes.cornell_wahoo_scene()
    cam = _inside(1920, 1080)
    tracer.upload_scene(scene)
    out = {}
    for mode in (T.TRACE_BVH, T.TRACE_BRUTE_FORCE):
        rgb8, rad = tracer.render(cam, lib.render_params(1920, 1080, 8, 1, trace_mode=mode))
        out[mode] = (rgb8, rad, tracer.stats().rays_traced)
    assert out[T.TRACE_BVH][2] == out[T.TRACE_BRUTE_FORCE][2]
    assert 1920 * 1080 <= out[T.TRACE_BVH][2] <= 1920 * 1080 * 8
    assert np.array_equal(out[T.TRACE_BVH][0], out[T.TRACE_BRUTE_FORCE][0])
    assert np.array_equal(out[T.TRACE_BVH][1].view(np.uint32), out[T.TRACE_BRUTE_FORCE][1].view(np.uint32))
    assert out[T.TRACE_BVH][1].max() > 0


def test_bvh_equals_brute_force_blooper_scene(tracer):
    scene = scenes.blooper_scene()
    cam = scenes.posed_camera(960, 540, position=(4.0, 1.0, 7.0), yaw=-118.0, pitch=-8.0)
    tracer.upload_scene(scene)
    a = tracer.render(cam, lib.render_params(960, 540, 6, 2, seed=77, trace_mode=T.TRACE_BVH))
    b = tracer.render(cam, lib.render_params(960, 540, 6, 2, seed=77, trace_mode=T.TRACE_BRUTE_FORCE))
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))


def test_million_triangle_sphere_matches_coarse_oracle_crop(tracer):
    """C4 (983 040 triangles): BVH on the GPU vs the brute-force CPU oracle on a small crop."""
    scene = scenes.sphere_stress_scene(5)
    info = lib.scene_info(scene)
    assert info.num_triangles == 983040 and info.valid == 1
    cam = _inside(1920, 1080)
    tracer.upload_scene(scene)
    params = lib.render_params(1920, 1080, 1, 1, shade_mode=T.SHADE_NORMAL_DEBUG)
    rgb8, rad = tracer.render(cam, params)
    x0, y0, w, h = 940, 520, 24, 16
    exp8, exprad = oracle_render(scene, cam, params, window=(x0, y0, w, h), threads=16)
    assert np.array_equal(rgb8[y0:y0 + h, x0:x0 + w], exp8)
    assert np.array_equal(rad[y0:y0 + h, x0:x0 + w].view(np.uint32), exprad.view(np.uint32))
    assert rgb8[y0:y0 + h, x0:x0 + w].any()


@pytest.mark.parametrize("parts", [2, 3, 8])
def test_strip_union_is_the_whole_frame(tracer, parts):
    """Multi-GPU partition on one GPU: rendering the strips part by part and de-interleaving them reproduces the
    single-launch frame bit for bit (the RNG is keyed on the global pixel index)."""
    import torch
    scene = scenes.cornell_wahoo_scene()
    w, h = 640, 360 + 8  # a short last strip
    cam = _inside(w, h)
    tracer.upload_scene(scene)
    params = lib.render_params(w, h, 5, 2, seed=4321)
    full8, fullr = tracer.render(cam, params)
    blocks8, blocksr = [], []
    for p in range(parts):
        r8, rr = tracer.render_strips(cam, params, ffdist.STRIP_ROWS, p, parts)
        assert r8.shape[0] == ffdist.strip_layout(h, ffdist.STRIP_ROWS, parts)[p]
        blocks8.append(r8)
        blocksr.append(rr)
    img8 = ffdist.deinterleave_host(torch.from_numpy(np.concatenate(blocks8)), h, ffdist.STRIP_ROWS, parts).numpy()
    imgr = ffdist.deinterleave_host(torch.from_numpy(np.concatenate(blocksr)), h, ffdist.STRIP_ROWS, parts).numpy()
    assert np.array_equal(img8, full8)
    assert np.array_equal(imgr.view(np.uint32), fullr.view(np.uint32))


def test_device_deinterleave_kernel(tracer):
    import torch
    h, w, parts = 100, 37, 3
    y, x = np.mgrid[0:h, 0:w]
    full = np.stack([(y * w + x) % 251, y % 256, x % 256], axis=2).astype(np.uint8)
    packed = np.concatenate([full[ffdist.strip_row_indices(h, 16, p, parts)] for p in range(parts)])
    src = torch.from_numpy(packed).cuda()
    dst = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
    tracer.deinterleave_strips(src.data_ptr(), dst.data_ptr(), w, h, 16, parts, 3)
    assert np.array_equal(dst.cpu().numpy(), full)
    srcf = torch.from_numpy(packed.astype(np.float32)).cuda()
    dstf = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    tracer.deinterleave_strips(srcf.data_ptr(), dstf.data_ptr(), w, h, 16, parts, 12)
    assert np.array_equal(dstf.cpu().numpy(), full.astype(np.float32))


def test_spp_chunking_and_determinism(tracer):
    scene = scenes.cornell_wahoo_scene()
    cam = _inside(320, 180)
    tracer.upload_scene(scene)
    one = tracer.render(cam, lib.render_params(320, 180, 8, 200))
    assert tracer.stats().kernel_launches == 1
    again = tracer.render(cam, lib.render_params(320, 180, 8, 200))
    chunks = tracer.render(cam, lib.render_params(320, 180, 8, 200, spp_per_launch=64))  # 200 spp = 4 sample blocks
    assert tracer.stats().kernel_launches == 4
    for other in (again, chunks):
        assert np.array_equal(one[0], other[0]) and np.array_equal(one[1].view(np.uint32), other[1].view(np.uint32))
    different_seed = tracer.render(cam, lib.render_params(320, 180, 8, 200, seed=1))
    assert not np.array_equal(one[1], different_seed[1])


def test_device_output_pointers(tracer):
    import torch
    scene = scenes.cornell_wahoo_scene()
    cam = _inside(200, 120)
    tracer.upload_scene(scene)
    params = lib.render_params(200, 120, 4, 3)
    host8, hostr = tracer.render(cam, params)
    d8 = torch.zeros((120, 200, 3), dtype=torch.uint8, device="cuda")
    dr = torch.zeros((120, 200, 3), dtype=torch.float32, device="cuda")
    tracer.set_stream(torch.cuda.current_stream().cuda_stream)
    tracer.render_device(cam, params, d8.data_ptr(), dr.data_ptr())
    tracer.set_stream(0)
    assert np.array_equal(d8.cpu().numpy(), host8) and np.array_equal(dr.cpu().numpy().view(np.uint32), hostr.view(np.uint32))
    st = tracer.stats()
    assert st.kernel_ms > 0 and st.total_ms >= st.kernel_ms * 0.5 and st.kernel_launches == 1


def test_stats_counters(tracer):
    scene = scenes.cornell_wahoo_scene()
    cam = _inside(256, 144)
    tracer.upload_scene(scene)
    params = lib.render_params(256, 144, 8, 4)
    tracer.set_collect_stats(True)
    tracer.render(cam, params)
    st = tracer.stats()
    tracer.set_collect_stats(False)
    assert 256 * 144 * 4 <= st.rays_traced <= 256 * 144 * 4 * 8
    assert st.nodes_visited > st.rays_traced and st.tris_tested > 0 and st.planes_tested > 0
    assert st.scene_bytes_tris == 5184 * 48 and st.scene_bytes_nodes % 112 == 0 and st.scene_bytes_nodes > 0
    tracer.render(cam, params)
    assert tracer.stats().nodes_visited == 0  # counters are off again


def test_edge_case_scenes(tracer):
    red = scenes.make_bxdf(T.BXDF_DIFFUSE, albedo=(1, 0, 0))
    light = scenes.make_bxdf(T.BXDF_EMITTER, emissive=(1, 1, 1), intensity=2.0)
    one = np.zeros((1, 24), dtype=np.float32)
    one[0, :9] = [-1, -1, 0, 1, -1, 0, 0, 1, 0]
    cam = scenes.default_camera(97, 61)  # ragged: neither a multiple of 8 nor of 16
    cases = {
        "plane_only": scenes.Scene().add_plane((0, 0, 0), (0, 0, 0), (4, 4, 4), red).add_plane((0, 3, 3), (60, 0, 0), (6, 6, 6), light).finalize(),
        "empty_mesh": scenes.Scene().add_mesh(np.zeros((0, 24), np.float32), bxdf=red).add_plane((0, 0, 0), (0, 0, 0), (4, 4, 4), red).finalize(),
        "one_triangle": scenes.Scene().add_mesh(one, (0, 0, 1), (0, 0, 0), (3, 3, 3), red).add_plane((0, 3, 3), (60, 0, 0), (6, 6, 6), light).finalize(),
    }
    for name, scene in cases.items():
        tracer.upload_scene(scene)
        for shade, b, s in ((T.SHADE_NORMAL_DEBUG, 1, 1), (T.SHADE_DIFFUSE_PATH, 4, 3)):
            for mode in (T.TRACE_BVH, T.TRACE_BRUTE_FORCE):
                params = lib.render_params(97, 61, b, s, trace_mode=mode, shade_mode=shade)
                rgb8, rad = tracer.render(cam, params)
                exp8, exprad = oracle_render(scene, cam, params, threads=8)
                assert np.array_equal(rgb8, exp8), (name, shade, mode)
                assert np.array_equal(rad.view(np.uint32), exprad.view(np.uint32)), (name, shade, mode)
    assert exp8.any()


def test_reference_floor_grid_and_errors(tracer):
    scene = scenes.reference_scene(scenes.load_mesh("cube"))
    tracer.upload_scene(scene)
    cam = scenes.default_camera(1920, 1080)
    rgb8, _ = tracer.render(cam, lib.render_params(1920, 1080, 1, 1, shade_mode=T.SHADE_NORMAL_DEBUG, grid_mode=T.GRID_REFERENCE_FLOOR))
    full, _ = tracer.render(cam, lib.render_params(1920, 1080, 1, 1, shade_mode=T.SHADE_NORMAL_DEBUG))
    assert not rgb8[1072:].any()                     # kernel.cu:308-309: rows 1072..1079 are never traced
    assert np.array_equal(rgb8[:1072], full[:1072])
    for bad in (lib.render_params(0, 10), lib.render_params(16, 16, bounces=0), lib.render_params(16, 16, spp=0),
                lib.render_params(16, 16, trace_mode=7)):
        with pytest.raises(lib.FireflyError) as e:
            tracer.render(cam, bad)
        assert e.value.status == T.FF_ERR_INVALID_ARG
    fresh = lib.Tracer(0)
    with pytest.raises(lib.FireflyError) as e:
        fresh.render(cam, lib.render_params(16, 16))
    assert e.value.status == T.FF_ERR_NO_SCENE
    fresh.close()


def test_non_unit_and_degenerate_rays(tracer):
    """intersectRays takes arbitrary rays: non-normalised and axis-aligned directions must match the oracle too."""
    from oracle_lib import oracle_intersect
    scene = scenes.cornell_wahoo_scene()
    tracer.upload_scene(scene)
    rng = np.random.default_rng(5)
    o = rng.uniform(-2, 2, size=(600, 3)).astype(np.float32)
    d = rng.normal(size=(600, 3)).astype(np.float32) * rng.uniform(0.05, 20.0, size=(600, 1)).astype(np.float32)
    d[:60] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 60)] * rng.choice([-1.0, 1.0], size=(60, 1)).astype(np.float32)
    for mode in (T.TRACE_BVH, T.TRACE_BRUTE_FORCE):
        got = tracer.intersect_rays(o, d, mode)
        exp = oracle_intersect(scene, o, d)
        assert np.array_equal(got["hit"], exp["hit"])
        h = exp["hit"] == 1
        assert np.array_equal(got["geom"][h], exp["geom"][h]) and np.array_equal(got["tri"][h], exp["tri"][h])
        assert np.array_equal(got["t"][h].view(np.uint32), exp["t"][h].view(np.uint32))
        assert np.array_equal(got["point"][h].view(np.uint32), exp["point"][h].view(np.uint32))


def test_gl_interop_reports_unavailable_on_a_headless_box(tracer):
    """ff_register_gl_pbo (utilities.h:618 twin) without a current GL context: a status code, not a crash."""
    import ctypes as C
    l = lib.load()
    st = C.c_void_p()
    assert l.ff_create(C.byref(st), 0) == T.FF_OK
    assert l.ff_register_gl_pbo(st, 1, 64, 64) == T.FF_ERR_GL_UNAVAILABLE
    assert b"hipGraphicsGLRegisterBuffer" in l.ff_last_error()
    cam = scenes.default_camera(64, 64)
    params = lib.render_params(64, 64)
    assert l.ff_render_to_pbo(st, C.byref(cam), C.byref(params)) == T.FF_ERR_GL_UNAVAILABLE
    assert l.ff_unregister_gl_pbo(st) == T.FF_OK
    l.ff_destroy(st)


def test_scene_file_renders_like_the_oracle(tracer):
    import os
    sf = lib.SceneFile(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "box.scene"))
    cam = sf.camera(120, 90)
    tracer.upload_scene(sf)
    params = lib.render_params(120, 90, 4, 3)
    rgb8, rad = tracer.render(cam, params)
    exp8, exprad = oracle_render(sf, cam, params, threads=8)
    assert np.array_equal(rgb8, exp8) and np.array_equal(rad.view(np.uint32), exprad.view(np.uint32)) and exprad.max() > 0


def test_c_example_through_the_abi(tracer, tmp_path):
    """examples/headless_render.c: plain C against include/firefly/*.h + libfirefly_hip.so (no Python, no HIP headers)
    produces the same framebuffer as the ctypes path."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "headless_render")
    libdir = os.path.join(root, "gpupathtracer_amd")
    subprocess.check_call(["gcc", "-O2", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "headless_render.c"),
                           "-L" + libdir, "-lfirefly_hip", "-Wl,-rpath," + libdir, "-o", exe])
    scene_path = os.path.join(root, "tests", "data", "box.scene")
    out = str(tmp_path / "out.ppm")
    subprocess.check_call([exe, scene_path, out, "96", "64", "3", "5"])
    with open(out, "rb") as f:
        assert f.readline() == b"P6\n" and f.readline() == b"96 64\n" and f.readline() == b"255\n"
        img = np.frombuffer(f.read(), dtype=np.uint8).reshape(64, 96, 3)
    sf = lib.SceneFile(scene_path)
    tracer.upload_scene(sf)
    rgb8, _ = tracer.render(sf.camera(96, 64), lib.render_params(96, 64, 3, 5))
    assert np.array_equal(img, rgb8) and img.any()


def test_lean_ieee_sequences_are_exact_for_every_float(tracer):
    """The kernels replace the IEEE expansions of 1/x and sqrt(x) by one Newton step on the hardware estimates inside
    2^-60..2^60 (full expansion outside): bit-identical for all 2^32 inputs, or the parity claims would not hold."""
    assert tracer.check_ieee() == (0, 0)


def test_fine_grained_tail_is_bit_identical(monkeypatch):
    """From four sample blocks on, the frame's last block is traced as 16-sample items with per-sample storage and summed in
    order by the combine pass: same bits as the all-in-registers path (FF_TAIL_GROUP=0) and as the oracle, including a
    partial last block."""
    scene = scenes.cornell_wahoo_scene()
    cam = scenes.posed_camera(160, 90, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
    out = {}
    for group in ("0", "32", "16", "8"):
        monkeypatch.setenv("FF_TAIL_GROUP", group)
        with lib.Tracer(0) as t:
            t.upload_scene(scene)
            for spp in (256, 300, 1024 + 40):
                out[(group, spp)] = t.render(cam, lib.render_params(160, 90, 5, spp, 77))
    for spp in (256, 300, 1024 + 40):
        for group in ("32", "16", "8"):
            assert np.array_equal(out[(group, spp)][0], out[("0", spp)][0])
            assert np.array_equal(out[(group, spp)][1].view(np.uint32), out[("0", spp)][1].view(np.uint32)), (group, spp)
    monkeypatch.delenv("FF_TAIL_GROUP")
    # a frame split into several launches (the tail then belongs to the last launch only, if it has four blocks)
    with lib.Tracer(0) as t:
        t.upload_scene(scene)
        for per_launch in (64, 128, 600):
            chunked = t.render(cam, lib.render_params(160, 90, 5, 1024 + 40, 77, spp_per_launch=per_launch))[1]
            assert np.array_equal(chunked.view(np.uint32), out[("0", 1024 + 40)][1].view(np.uint32)), per_launch
    # the strips of a multi-GPU rank (their own launches, their own tail) equal the rows of the full frame
    full = out[("0", 300)][1]
    p300 = lib.render_params(160, 90, 5, 300, 77)
    with lib.Tracer(0) as t:
        t.upload_scene(scene)
        for part in range(3):
            _, rad = t.render_strips(cam, p300, 4, part, 3)
            rows = ffdist.strip_row_indices(90, 4, part, 3)
            assert np.array_equal(rad.view(np.uint32), full[rows].view(np.uint32)), part
    # ... whether the frame's last one, two or three sample blocks go out as short items (csrc/ff_api.cpp: two where a rank's launch is
    # short, FF_TAIL_BLOCKS forces a number; a partial last block, and a frame all of whose blocks are tail blocks)
    for blocks in ("1", "2", "3"):
        monkeypatch.setenv("FF_TAIL_BLOCKS", blocks)
        with lib.Tracer(0) as t:
            t.upload_scene(scene)
            for spp in (300, 256, 100):
                ref = out[("0", spp)][1] if ("0", spp) in out else None
                if ref is None:
                    monkeypatch.setenv("FF_TAIL_GROUP", "0")
                    with lib.Tracer(0) as t0:
                        t0.upload_scene(scene)
                        ref = t0.render(cam, lib.render_params(160, 90, 5, spp, 77))[1]
                    monkeypatch.delenv("FF_TAIL_GROUP")
                for part in range(2):
                    _, rad = t.render_strips(cam, lib.render_params(160, 90, 5, spp, 77), 8, part, 2)
                    assert t.stats().flags & T.FF_STATS_TAIL_ITEMS or spp == 100, (blocks, spp)
                    rows = ffdist.strip_row_indices(90, 8, part, 2)
                    assert np.array_equal(rad.view(np.uint32), ref[rows].view(np.uint32)), (blocks, spp, part)
    monkeypatch.delenv("FF_TAIL_BLOCKS")
    small = scenes.posed_camera(12, 9, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
    p = lib.render_params(12, 9, 4, 300, 5)  # five blocks, the last one holds 44 samples
    with lib.Tracer(0) as t:
        t.upload_scene(scene)
        rgb8, rad = t.render(small, p)
    o_rgb8, o_rad = oracle_render(scene, small, p, threads=16)
    assert np.array_equal(rgb8, o_rgb8) and np.array_equal(rad.view(np.uint32), o_rad.view(np.uint32))


def test_work_queue_knobs_do_not_change_a_bit(tracer, monkeypatch):
    """The work queue (csrc/ff_kernels.hip acquire_pixel: several counters that each own every n-th stripe of 64 items, waves that
    take a chunk per atomic and deal it to their lanes) only decides WHO traces an item: any number of counters and any chunk size
    give the same frame - every item exactly once - on frames of one and of several launches, with the fine-grained tail, on the
    strips of a multi-GPU rank and in brute-force mode."""
    scene = scenes.cornell_wahoo_scene()
    w, h = 328, 187  # 41 x 24 tiles of 8x8 (the last column and row are ragged), 984 stripes
    cam = _inside(w, h)
    tracer.upload_scene(scene)
    cases = {
        "1spp": (lib.render_params(w, h, 6, 1, 9), None),
        "3 blocks": (lib.render_params(w, h, 3, 130, 9), None),
        "launches": (lib.render_params(w, h, 3, 200, 9, spp_per_launch=64), None),
        "brute": (lib.render_params(w, h, 2, 1, 9, trace_mode=T.TRACE_BRUTE_FORCE), None),
        "strips": (lib.render_params(w, h, 3, 70, 9), (8, 1, 3)),
    }

    def frames():
        out = {}
        for name, (p, strips) in cases.items():
            rgb8, rad = tracer.render(cam, p) if strips is None else tracer.render_strips(cam, p, *strips)
            out[name] = (rgb8.copy(), rad.view(np.uint32).copy(), tracer.stats().rays_traced)
        return out

    ref = frames()
    assert ref["1spp"][1].any()
    for counters, chunk in (("1", "1"), ("3", "7"), ("16", "64"), ("64", "4096"), ("5", "100")):
        monkeypatch.setenv("FF_QUEUE_COUNTERS", counters)
        monkeypatch.setenv("FF_QUEUE_CHUNK", chunk)
        tracer.reload_switches()
        got = frames()
        for name in cases:
            assert got[name][2] == ref[name][2], (counters, chunk, name)
            assert np.array_equal(got[name][0], ref[name][0]) and np.array_equal(got[name][1], ref[name][1]), (counters, chunk, name)
    monkeypatch.setenv("FF_TAIL_GROUP", "16")  # (read at ff_create)
    with lib.Tracer(0) as t:
        t.upload_scene(scene)
        p = lib.render_params(w, h, 3, 300, 9)
        monkeypatch.delenv("FF_QUEUE_COUNTERS")
        monkeypatch.delenv("FF_QUEUE_CHUNK")
        t.reload_switches()
        a = t.render(cam, p)[1].view(np.uint32).copy()
        monkeypatch.setenv("FF_QUEUE_COUNTERS", "7")
        monkeypatch.setenv("FF_QUEUE_CHUNK", "33")
        t.reload_switches()
        b = t.render(cam, p)[1].view(np.uint32).copy()
        assert t.stats().flags & T.FF_STATS_TAIL_ITEMS
    assert np.array_equal(a, b)


def test_stats_flags_report_the_tail_policy(tracer):
    """FfStats::flags: short one-GPU frames use the fine-grained tail, the 1 024-spp frame does not, and a frame whose
    per-sample buffer would pass 16 GiB says that it went without (31 M pixels x 64 samples x 16 B)."""
    tracer.upload_scene(scenes.cornell_wahoo_scene())
    tracer.render(_inside(320, 180), lib.render_params(320, 180, 3, 8, 1), want_rgb8=False, want_radiance=False)
    assert tracer.stats().flags == T.FF_STATS_TAIL_ITEMS
    tracer.render(_inside(64, 36), lib.render_params(64, 36, 2, 1024, 1), want_rgb8=False, want_radiance=False)
    assert tracer.stats().flags == 0
    tracer.render(_inside(64, 36), lib.render_params(64, 36, 2, 1, 1), want_rgb8=False, want_radiance=False)
    assert tracer.stats().flags == 0  # one sample: nothing to split
    tracer.render(_inside(7680, 4096), lib.render_params(7680, 4096, 1, 64, 1), want_rgb8=False, want_radiance=False)
    assert tracer.stats().flags == T.FF_STATS_TAIL_SKIPPED_TOO_LARGE
    assert tracer.stats().rays_traced == 7680 * 4096 * 64


def test_launch_timeline_counts_every_ray(monkeypatch):
    """FF_DEBUG_TIMELINE_US: instrumented launches histogram ray completions over the launch's wall clock (one row per wave,
    added up by ff_debug_timeline).  Every ray is counted exactly once, and the instrumented frame is the plain frame."""
    monkeypatch.setenv("FF_DEBUG_TIMELINE_US", "20")
    scene = scenes.cornell_wahoo_scene()
    cam = _inside(320, 180)
    p = lib.render_params(320, 180, 6, 8, 3)
    with lib.Tracer(0) as t:
        t.upload_scene(scene)
        plain = t.render(cam, p)[1].view(np.uint32).copy()
        rays = t.stats().rays_traced
        t.set_collect_stats(True)
        inst = t.render(cam, p)[1].view(np.uint32).copy()
        inst_rays = t.stats().rays_traced
        us, counts = t.debug_timeline()
        t.set_collect_stats(False)
    assert us == 20 and int(counts.sum()) == rays == inst_rays
    # (where in the launch's wall clock the rays complete depends on the box: tools/timeline_probe.py reports it, nothing here
    # asserts on it - a correctness suite under -x must not depend on the speed of the machine)
    assert np.array_equal(plain, inst)
    monkeypatch.delenv("FF_DEBUG_TIMELINE_US")
    with lib.Tracer(0) as t:
        t.upload_scene(scene)
        t.set_collect_stats(True)
        t.render(cam, p)
        us, counts = t.debug_timeline()
    assert us == 0 and not counts.any()


def test_tiles_equal_the_full_frame(tracer):
    """ff_render_tile: arbitrary rectangles (ragged sizes, image corners) carry the pixels of the full frame, bit for bit,
    in both shade modes and with enough samples for the fine-grained tail."""
    scene = scenes.cornell_wahoo_scene()
    cam = scenes.posed_camera(101, 67, position=(0.2, -0.3, 2.3), yaw=-95.0, pitch=-4.0)
    tracer.upload_scene(scene)
    for params in (lib.render_params(101, 67, 4, 3, 9), lib.render_params(101, 67, 3, 260, 9),
                   lib.render_params(101, 67, 1, 1, 9, T.TRACE_BVH, T.SHADE_NORMAL_DEBUG, T.GRID_FULL, 0)):
        full8, full = tracer.render(cam, params)
        for (x0, y0, w, h) in ((0, 0, 101, 67), (0, 0, 17, 9), (84, 58, 17, 9), (33, 20, 40, 31), (100, 66, 1, 1)):
            rgb8, rad = tracer.render_tile(cam, params, x0, y0, w, h)
            assert np.array_equal(rgb8, full8[y0:y0 + h, x0:x0 + w]), (x0, y0, w, h)
            assert np.array_equal(rad.view(np.uint32), full[y0:y0 + h, x0:x0 + w].view(np.uint32)), (x0, y0, w, h)
    with pytest.raises(lib.FireflyError) as e:
        tracer.render_tile(cam, lib.render_params(101, 67, 1, 1), 90, 0, 20, 5)
    assert e.value.status == T.FF_ERR_INVALID_ARG


@pytest.mark.parametrize("builder", [T.BUILD_HOST_SAH, T.BUILD_GPU_LBVH])
def test_failed_scene_allocation_leaves_no_half_uploaded_scene(monkeypatch, builder):
    """FF_DEBUG_FAIL_ALLOC=k makes the k-th scene allocation of every upload report out-of-memory: the upload returns
    FF_ERR_OOM, frees what it had allocated and the state reads 'no scene' — whichever allocation it was."""
    scene = scenes.cornell_wahoo_scene()
    cam = _inside(64, 36)
    params = lib.render_params(64, 36, 2, 1)
    with lib.Tracer(0) as good:
        good.upload_scene(scene)
        ref = good.render(cam, params)
    failures = 0
    for k in range(8):
        monkeypatch.setenv("FF_DEBUG_FAIL_ALLOC", str(k))
        with lib.Tracer(0) as t:
            t.set_builder(builder)
            try:
                t.upload_scene(scene)
            except lib.FireflyError as e:
                failures += 1
                assert e.status == T.FF_ERR_OOM, (k, e)
                with pytest.raises(lib.FireflyError) as e2:
                    t.render(cam, params)
                assert e2.value.status == T.FF_ERR_NO_SCENE
            else:  # past the last allocation: an ordinary upload
                got = t.render(cam, params)
                assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1].view(np.uint32), ref[1].view(np.uint32))
    assert failures >= 5  # records, triangles, normals, nodes, 4-wide nodes, parents


def test_primary_rays_that_miss_the_scene_box_are_dropped_at_the_queue(tracer, monkeypatch):
    """A camera outside the padded box around all geometries (the reference's default one, kernel.cu:312-321): pixels whose primary
    ray misses that box are written as zero sums when their work item is fetched (csrc/ff_kernels.hip acquire_pixel) instead of
    being traced sample by sample.  The frame, and the number of rays counted, equal the brute-force kernel's (which takes no such
    shortcut) and the same kernel's with the shortcut off; in path mode, in the reference's own shade mode, on tiles and strips,
    and with enough samples for several blocks per pixel."""
    scene = scenes.cornell_wahoo_scene()
    for (w, h), cam_args in (((320, 180), None), ((200, 150), dict(position=(9.0, 4.0, 11.0), yaw=-130.0, pitch=-15.0))):
        cam = scenes.default_camera(w, h) if cam_args is None else scenes.posed_camera(w, h, **cam_args)
        tracer.upload_scene(scene)
        for params in (lib.render_params(w, h, 6, 5, 3), lib.render_params(w, h, 3, 130, 4), lib.render_params(w, h, 1, 1, shade_mode=T.SHADE_NORMAL_DEBUG)):
            bvh = tracer.render(cam, params)
            rays = tracer.stats().rays_traced
            st = tracer.stats()
            assert st.rays_answered <= rays
            if st.flags & T.FF_STATS_TAIL_ITEMS and params.spp <= 64:
                # the frame's only block went out sample by sample: nothing to drop, no mask pass; only repeated primaries count
                assert st.rays_answered <= w * h * params.spp
            assert (bvh[1] == 0).all(axis=2).mean() > 0.5  # most of the frame is background
            params.trace_mode = T.TRACE_BRUTE_FORCE
            brute = tracer.render(cam, params)
            assert rays == tracer.stats().rays_traced
            assert np.array_equal(bvh[0], brute[0]) and np.array_equal(bvh[1].view(np.uint32), brute[1].view(np.uint32))
            params.trace_mode = T.TRACE_BVH
            monkeypatch.setenv("FF_NO_PRIMARY_CULL", "1")
            tracer.reload_switches()
            off = tracer.render(cam, params)
            monkeypatch.delenv("FF_NO_PRIMARY_CULL")
            tracer.reload_switches()
            assert rays == tracer.stats().rays_traced and np.array_equal(bvh[1].view(np.uint32), off[1].view(np.uint32))
        # a tile across the silhouette of the box and the strips of a three-part frame carry the full frame's pixels
        params = lib.render_params(w, h, 4, 3, 8)
        full = tracer.render(cam, params)
        tile = tracer.render_tile(cam, params, w // 4, h // 4, w // 2, h // 2)
        assert np.array_equal(tile[1].view(np.uint32), full[1][h // 4:h // 4 + h // 2, w // 4:w // 4 + w // 2].view(np.uint32))


def test_culled_pixels_have_a_stored_primary_hit_for_their_tail_items():
    """A frame of a few sample blocks from a camera outside the scene: the whole-block items of the pixels that see nothing are
    dropped at the queue, but their LAST block goes out sample by sample (fine-grained tail) and every one of those samples starts
    from the pixel's stored primary hit - which the pre-pass must therefore have written for culled pixels too.  On a fresh state
    (the hit buffer holds nothing from earlier frames), against the brute-force kernel."""
    scene = scenes.cornell_wahoo_scene()
    w, h = 200, 150
    cam = scenes.posed_camera(w, h, position=(9.0, 4.0, 11.0), yaw=-130.0, pitch=-15.0)
    for spp in (192, 512):  # (three and eight blocks of 64: the last one is a full block, handed out in groups of 16 samples)
        with lib.Tracer(0) as t:
            t.upload_scene(scene)
            params = lib.render_params(w, h, 4, spp, 11)
            bvh = t.render(cam, params)
            st = t.stats()
            assert st.flags & T.FF_STATS_TAIL_ITEMS and st.rays_answered > 0
            params.trace_mode = T.TRACE_BRUTE_FORCE
            brute = t.render(cam, params)
            assert st.rays_traced == t.stats().rays_traced
            assert np.array_equal(bvh[0], brute[0]) and np.array_equal(bvh[1].view(np.uint32), brute[1].view(np.uint32)), spp


def test_primary_rays_are_answered_from_the_pixel_s_stored_hit(tracer, monkeypatch):
    """Every sample of a pixel starts with the same ray (kernel.cu:200-205 has no jitter): a pre-pass of the frame traces every
    pixel's primary ray once and stores its closest hit per pixel; every sample of the frame starts from there (csrc/ff_kernels.hip
    trace_bvh_kernel, settle_hit; csrc/ff_api.cpp render_enqueue).  Same bits and the same number of path segments as with the
    reuse off (FF_NO_PRIMARY_REUSE=1) and as the brute-force kernel, which traces every one of them; FfStats::rays_answered counts
    exactly the primary segments: one per sample (camera inside the box: no pixel is culled); with diffuse, mirror and glass
    surfaces, one bounce (every path is its primary segment), partial last blocks, several launches per frame and interpolated
    normals."""
    monkeypatch.setenv("FF_NO_PRIMARY_CULL", "1")  # (its pixels would count as answered as well)
    tracer.reload_switches()
    inside = scenes.posed_camera(96, 64, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
    # the open scene from outside: most primary rays hit nothing - the first sample of a block finds that out and is the block's
    # last (a block of zeros adds up to zero); the others count as answered all the same
    outside = scenes.posed_camera(96, 64, position=(4.0, 1.0, 7.0), yaw=-118.0, pitch=-8.0)
    for scene, cam in ((scenes.cornell_wahoo_scene(), inside), (scenes.cornell_glass_scene(), inside), (scenes.blooper_scene(), outside)):
        tracer.upload_scene(scene)
        for bounces, spp, per_launch, shade in ((6, 70, 0, T.SHADE_DIFFUSE_PATH), (1, 130, 0, T.SHADE_DIFFUSE_PATH), (4, 200, 64, T.SHADE_DIFFUSE_PATH_SMOOTH), (3, 1100, 0, T.SHADE_DIFFUSE_PATH)):
            params = lib.render_params(96, 64, bounces, spp, 21, T.TRACE_BVH, shade, T.GRID_FULL, per_launch)
            on = tracer.render(cam, params)
            st = tracer.stats()
            block_spp = 64 * ((spp + 1023) // 1024)
            blocks = (spp + block_spp - 1) // block_spp
            tail = st.flags & T.FF_STATS_TAIL_ITEMS
            if not tail:
                assert st.rays_answered == 96 * 64 * spp, (bounces, spp, st.rays_answered, blocks)
            else:
                assert 0 < st.rays_answered <= 96 * 64 * spp
            monkeypatch.setenv("FF_NO_PRIMARY_REUSE", "1")
            tracer.reload_switches()
            off = tracer.render(cam, params)
            st_off = tracer.stats()
            monkeypatch.delenv("FF_NO_PRIMARY_REUSE")
            tracer.reload_switches()
            assert st_off.rays_answered == 0 and st_off.rays_traced == st.rays_traced
            assert np.array_equal(on[0], off[0]) and np.array_equal(on[1].view(np.uint32), off[1].view(np.uint32))
            # FF_REUSE_QUORUM: lanes with a parked hit wait for the next iteration's shading unless that many of them are ready
            # (65: they always wait, also when no other lane of the wave has a query left)
            for quorum in ("5", "65"):
                monkeypatch.setenv("FF_REUSE_QUORUM", quorum)
                tracer.reload_switches()
                held = tracer.render(cam, params)
                st_q = tracer.stats()
                monkeypatch.delenv("FF_REUSE_QUORUM")
                tracer.reload_switches()
                assert st_q.rays_answered == st.rays_answered and st_q.rays_traced == st.rays_traced
                assert np.array_equal(on[1].view(np.uint32), held[1].view(np.uint32))
            if spp <= 200:
                params.trace_mode = T.TRACE_BRUTE_FORCE
                brute = tracer.render(cam, params)
                assert tracer.stats().rays_traced == st.rays_traced and tracer.stats().rays_answered == 0
                assert np.array_equal(on[1].view(np.uint32), brute[1].view(np.uint32))


def test_last_bounce_queries_end_after_the_planes_when_no_emitter_is_held(tracer, monkeypatch):
    """The last segment of a path (bounce index bounces - 1) adds radiance only when it ends on an emitter
    (csrc/ff_kernels.hip shade_and_advance; oracle/ff_oracle.c restates the loop).  When every emitter of the scene is a plane
    or a sphere - records every query screens before any mesh - a last-bounce query that holds no emitter after that screening
    ends there (scan_records; FfStats::rays_cut_short).  Same bits and the same number of path segments as with the shortcut off
    (FF_NO_LAST_BOUNCE_CUT=1) and as the brute-force kernel; an emitter MESH switches it off; so does the normal-debug shade."""
    cam = scenes.posed_camera(96, 64, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
    lit_mesh = scenes.Scene()
    lit_mesh.add_mesh(scenes.load_mesh("cube"), (0.2, 1.2, -0.4), (0, 30, 0), (0.8, 0.3, 0.8), scenes.make_bxdf(T.BXDF_EMITTER, emissive=(1, 0.9, 0.8), intensity=3.0))
    lit_mesh.add_mesh(scenes.load_mesh("wahoo"), (0, -2.4, 0), (0, 0, 0), (0.28, 0.28, 0.28), scenes.make_bxdf(T.BXDF_DIFFUSE, albedo=(0.8, 0.8, 0.2)))
    scenes._box(lit_mesh).finalize()
    cases = [(scenes.cornell_wahoo_scene(), True), (scenes.cornell_glass_scene(), True), (scenes.cornell_spheres_scene(), True), (lit_mesh, False)]
    for scene, cuts in cases:
        tracer.upload_scene(scene)
        for bounces, spp, shade in ((1, 3, T.SHADE_DIFFUSE_PATH), (2, 5, T.SHADE_DIFFUSE_PATH), (8, 6, T.SHADE_DIFFUSE_PATH), (4, 70, T.SHADE_DIFFUSE_PATH_SMOOTH)):
            params = lib.render_params(96, 64, bounces, spp, 77, T.TRACE_BVH, shade)
            on = tracer.render(cam, params)
            st = tracer.stats()
            # (one bounce and several samples: the only segment of every path is its primary one, answered from the pixel's stored hit -
            # the pre-pass that stores it needs the true closest hit and cuts nothing)
            assert (st.rays_cut_short > 0) == (cuts and not (bounces == 1 and spp > 1)), (bounces, spp, st.rays_cut_short)
            assert st.rays_cut_short <= 96 * 64 * spp  # at most one last segment per path
            monkeypatch.setenv("FF_NO_LAST_BOUNCE_CUT", "1")
            tracer.reload_switches()
            off = tracer.render(cam, params)
            st_off = tracer.stats()
            monkeypatch.delenv("FF_NO_LAST_BOUNCE_CUT")
            tracer.reload_switches()
            assert st_off.rays_cut_short == 0 and st_off.rays_traced == st.rays_traced
            assert np.array_equal(on[0], off[0]) and np.array_equal(on[1].view(np.uint32), off[1].view(np.uint32))
            params.trace_mode = T.TRACE_BRUTE_FORCE
            brute = tracer.render(cam, params)
            assert tracer.stats().rays_traced == st.rays_traced and tracer.stats().rays_cut_short == 0
            assert np.array_equal(on[1].view(np.uint32), brute[1].view(np.uint32))
    tracer.upload_scene(scenes.cornell_wahoo_scene())
    tracer.render(cam, lib.render_params(96, 64, 1, 1, 0, T.TRACE_BVH, T.SHADE_NORMAL_DEBUG))
    assert tracer.stats().rays_cut_short == 0


def test_wall_pairs_and_wall_table_do_not_change_a_bit(monkeypatch):
    """Floor and ceiling, left and right wall of the box share one entry of the wall table (csrc/ff_scene.cpp build_wall_table,
    csrc/ff_kernels.hip wall_test_pair): with pairs, without them (FF_NO_WALL_PAIRS=1) and without the table (FF_NO_WALL_TABLE=1;
    both read at ff_create) the frame is the same, from inside the box, from outside it (origins beyond the pair: the other wall
    goes to the exact per-lane screen) and from a camera ON the floor plane."""
    scene = scenes.cornell_wahoo_scene()
    cams = [scenes.posed_camera(80, 48, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0),
            scenes.posed_camera(80, 48, position=(0.3, 6.0, 9.0), yaw=-92.0, pitch=-30.0),   # above the ceiling's plane, outside
            scenes.posed_camera(80, 48, position=(-6.0, -1.0, 1.0), yaw=-10.0, pitch=5.0),   # left of the left wall
            scenes.posed_camera(80, 48, position=(0.5, -2.5, 2.0), yaw=-95.0, pitch=10.0)]    # on the floor's plane
    params = lib.render_params(80, 48, 5, 6, 9)
    frames = {}
    for knob in (None, "FF_NO_WALL_PAIRS", "FF_NO_WALL_TABLE"):
        if knob:
            monkeypatch.setenv(knob, "1")
        with lib.Tracer(0) as t:
            t.upload_scene(scene)
            frames[knob] = [t.render(cam, params)[1].copy() for cam in cams]
        if knob:
            monkeypatch.delenv(knob)
    for knob in ("FF_NO_WALL_PAIRS", "FF_NO_WALL_TABLE"):
        for a, b in zip(frames[None], frames[knob]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), knob
    assert frames[None][0].max() > 0


def test_job_pool_kernel_equals_the_lane_owned_kernel(monkeypatch):
    """FF_POOL=1 (csrc/ff_kernels.hip trace_pool_kernel, an opt-in experiment: profiles/r04_a_pool_kernel_negative.txt): the traversal
    of a (ray, mesh) pair is a job parked in LDS that ANY wave of the workgroup walks, the path stays with its lane.  Only who walks
    changes: same bits, same ray / answered / cut-short counts as the lane-owned kernel, with diffuse, specular and smooth-normal
    scenes, several blocks per pixel, tail items (short frames), the normal-debug shade, a tile and the strips of a three-part
    frame; the scheduling knobs do not change a bit either."""
    cam = scenes.posed_camera(112, 72, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
    outside = scenes.posed_camera(112, 72, position=(4.0, 1.0, 7.0), yaw=-118.0, pitch=-8.0)
    cases = [(scenes.cornell_wahoo_scene(), cam, lib.render_params(112, 72, 8, 70, 5)),
             (scenes.cornell_wahoo_scene(), cam, lib.render_params(112, 72, 3, 1100, 5)),
             (scenes.cornell_wahoo_scene(), cam, lib.render_params(112, 72, 1, 1, 0, T.TRACE_BVH, T.SHADE_NORMAL_DEBUG)),
             (scenes.cornell_glass_scene(), cam, lib.render_params(112, 72, 6, 9, 3)),
             (scenes.cornell_spheres_scene(), cam, lib.render_params(112, 72, 5, 6, 3, T.TRACE_BVH, T.SHADE_DIFFUSE_PATH_SMOOTH)),
             (scenes.blooper_scene(), outside, lib.render_params(112, 72, 8, 130, 3))]

    def frames(env):
        for k in ("FF_POOL", "FF_POOL_QUORUM", "FF_POOL_SLICE", "FF_POOL_REFILL", "FF_POOL_LEAVE", "FF_POOL_BATCH_MIN", "FF_POOL_STACK_LEVELS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        out = []
        with lib.Tracer(0) as t:
            for scene, c, p in cases:
                t.upload_scene(scene)
                rgb8, rad = t.render(c, p)
                st = t.stats()
                out.append((rgb8.copy(), rad.view(np.uint32).copy(), st.rays_traced, st.rays_answered, st.rays_cut_short, t.kernel_name()))
            t.upload_scene(cases[0][0])
            p = lib.render_params(112, 72, 4, 3, 8)
            tile = t.render_tile(cam, p, 28, 18, 56, 36)
            strips = t.render_strips(cam, p, 8, 1, 3)
            out.append((tile[0].copy(), tile[1].view(np.uint32).copy(), 0, 0, 0, ""))
            out.append((strips[0].copy(), strips[1].view(np.uint32).copy(), 0, 0, 0, ""))
        return out

    ref = frames({"FF_POOL": "0"})
    assert all("trace_pool" not in r[5] for r in ref)
    for env in ({"FF_POOL": "1"}, {"FF_POOL": "1", "FF_POOL_QUORUM": "9", "FF_POOL_SLICE": "1", "FF_POOL_REFILL": "3", "FF_POOL_LEAVE": "60", "FF_POOL_BATCH_MIN": "2"},
                {"FF_POOL": "1", "FF_POOL_QUORUM": "64", "FF_POOL_SLICE": "9", "FF_POOL_LEAVE": "0", "FF_POOL_STACK_LEVELS": "2"}):
        got = frames(env)
        assert all("trace_pool_kernel" in g[5] for g in got[:len(cases)]), [g[5] for g in got]
        for i, (a, b) in enumerate(zip(ref, got)):
            assert a[2:5] == b[2:5], (env, i, a[2:5], b[2:5])
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (env, i)


def test_stored_primary_hits_are_kept_while_camera_and_scene_stay(monkeypatch):
    """The pixels' stored primary hits (and the mask of pixels that see nothing) belong to a camera, a pixel mapping and a scene: the
    next frame from the same ones starts from them without a pre-pass (csrc/ff_api.cpp render_enqueue) - a viewer that accumulates
    1-spp frames with the camera at rest (kernel.cu:266,342).  Frames rendered that way equal the frames of a state that keeps nothing
    (FF_NO_PRIMARY_CACHE=1) and the brute-force kernel's, bits and ray / answered counts; a new camera, another frame size, a tile, a
    transform update and a mesh refit each make the next frame compute its own.  A 1-spp frame starts from stored hits when they are
    there or when the frame before it had the same camera (it has come to rest: this pre-pass is the last one); a one-off 1-spp frame
    traces its primary rays itself (FF_REUSE_MIN_SPP=1: every frame runs on stored hits)."""
    scene = scenes.cornell_wahoo_scene()
    inside = scenes.posed_camera(160, 96, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
    outside = scenes.posed_camera(160, 96, position=(6.0, 2.0, 9.0), yaw=-125.0, pitch=-10.0)
    steps = [("a", inside, lib.render_params(160, 96, 8, 1, 5)), ("a again", inside, lib.render_params(160, 96, 8, 1, 6)),
             ("a third", inside, lib.render_params(160, 96, 8, 1, 8)), ("a 70 spp", inside, lib.render_params(160, 96, 4, 70, 7)), ("a smooth", inside, lib.render_params(160, 96, 4, 3, 7, T.TRACE_BVH, T.SHADE_DIFFUSE_PATH_SMOOTH)),
             ("a flat again", inside, lib.render_params(160, 96, 4, 3, 7)), ("b", outside, lib.render_params(160, 96, 8, 1, 5)),
             ("b 130 spp", outside, lib.render_params(160, 96, 5, 130, 2)), ("b again", outside, lib.render_params(160, 96, 8, 3, 9)),
             ("b 200 spp", outside, lib.render_params(160, 96, 3, 200, 4)),  # (a fine-grained tail: the last block sample by sample, mask kept)
             ("a small", scenes.posed_camera(96, 64, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0), lib.render_params(96, 64, 6, 2, 1)),
             ("a back", inside, lib.render_params(160, 96, 8, 1, 5))]

    def run(env):
        for k in ("FF_NO_PRIMARY_CACHE", "FF_REUSE_MIN_SPP"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        out = []
        with lib.Tracer(0) as t:
            t.upload_scene(scene)
            for name, cam, p in steps:
                rgb8, rad = t.render(cam, p)
                st = t.stats()
                out.append((name, rad.view(np.uint32).copy(), st.rays_traced, st.rays_answered))
            tile = t.render_tile(inside, lib.render_params(160, 96, 4, 2, 3), 40, 24, 80, 48)
            out.append(("tile", tile[1].view(np.uint32).copy(), 0, 0))
            full = t.render(inside, lib.render_params(160, 96, 4, 2, 3))
            out.append(("full after tile", full[1].view(np.uint32).copy(), t.stats().rays_traced, t.stats().rays_answered))
            assert np.array_equal(out[-2][1], out[-1][1][24:72, 40:120])
            # the scene moves under a camera at rest: the stored hits go with the old scene
            shifted = scenes.Scene()
            shifted.add_mesh(scenes.load_mesh("wahoo"), (0.4, -2.4, 0.3), (0, 25, 0), (0.28, 0.28, 0.28), scenes.make_bxdf(T.BXDF_DIFFUSE, albedo=(1, 0, 0)))
            shifted.add_mesh(scenes.load_mesh("cube"), (1.2, -2.0, 0.6), (0, 10, 0), (1, 1, 1), scenes.make_bxdf(T.BXDF_DIFFUSE, albedo=(0.75, 0.75, 0.75)))
            scenes._box(shifted).finalize()
            t.update_transforms(shifted)
            upd = t.render(inside, lib.render_params(160, 96, 4, 2, 3))
            out.append(("after transform update", upd[1].view(np.uint32).copy(), t.stats().rays_traced, t.stats().rays_answered))
        with lib.Tracer(0) as t2:  # ... a fresh upload of the moved scene renders the same frame
            t2.upload_scene(shifted)
            fresh = t2.render(inside, lib.render_params(160, 96, 4, 2, 3))
            assert np.array_equal(out[-1][1], fresh[1].view(np.uint32)) and out[-1][2] == t2.stats().rays_traced
        return out

    kept = run({})
    none = run({"FF_NO_PRIMARY_CACHE": "1"})
    always = run({"FF_REUSE_MIN_SPP": "1"})
    by_name = lambda rows: {r[0]: r for r in rows}
    for a, b, c in zip(kept, none, always):
        assert np.array_equal(a[1], b[1]) and a[2] == b[2], a[0]
        assert np.array_equal(a[1], c[1]) and a[2] == c[2], a[0]
        if a[0] not in ("a", "a again", "a third", "b", "a back"):  # (frames of 2 spp or more run on stored hits in every state)
            assert a[3] == b[3] == c[3], a[0]
    # which 1-spp frames started from stored hits: rays_answered counts their primary segments (include/firefly/ff_types.h)
    k, n, al = by_name(kept), by_name(none), by_name(always)
    for name in ("a", "b", "a back"):          # one-off frames: primary rays traced by the frame itself
        assert k[name][3] == n[name][3] < al[name][3], name
    for name in ("a again", "a third"):        # the camera at rest: pre-pass once, then kept
        assert k[name][3] == al[name][3] > n[name][3], name
    with lib.Tracer(0) as t:
        t.upload_scene(scene)
        for (name, cam, p), ref in zip(steps, kept):
            p.trace_mode = T.TRACE_BRUTE_FORCE
            brute = t.render(cam, p)
            assert np.array_equal(brute[1].view(np.uint32), ref[1]) and t.stats().rays_traced == ref[2], name
            p.trace_mode = T.TRACE_BVH
