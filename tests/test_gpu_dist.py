"""GPU tests (`-m gpu`) of the multi-GPU entry points behind the C ABI (csrc/ff_dist.cpp) on a ONE-GPU box:

  * ff_multi_* with the same device listed several times (peer-copy transport: RCCL refuses duplicate devices): every
    part renders its strips on its own stream, device 0 gathers the packed strips and scatters them to image order;
  * ff_dist_* with a one-rank RCCL communicator whose rank 0 sends its own strips to itself (FF_DIST_SELF_LOOP=1): the
    grouped ncclSend / ncclRecv + scatter path that ranks 1..N-1 take on a real node;
  * two fresh processes sharing the GPU, each rendering its strips, gathered over gloo (the bench's rehearsal path).

The bar is the one of every partition test here: the gathered frame equals the single-launch frame bit for bit."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from gpupathtracer_amd import lib, scenes
from gpupathtracer_amd import types as T

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _same(a, b):
    return np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))


def _inside(w, h):
    return scenes.posed_camera(w, h, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)


@pytest.mark.parametrize("parts,size,strip_rows", [(2, (640, 368), 0), (3, (101, 67), 16), (8, (96, 20), 4), (4, (64, 64), 1)])
def test_one_process_several_parts(tracer, parts, size, strip_rows):
    """ff_multi_render: `parts` states on device 0 (8 parts of a 20-row image at 4-row strips: three parts own nothing)."""
    w, h = size
    scene = scenes.cornell_wahoo_scene()
    cam = _inside(w, h)
    tracer.upload_scene(scene)
    params = lib.render_params(w, h, 5, 3, seed=77)
    ref = tracer.render(cam, params)
    rays = tracer.stats().rays_traced
    with lib.MultiTracer([0] * parts) as m:
        assert len(m) == parts and not m.uses_rccl
        m.upload_scene(scene)
        got = m.render(cam, params, strip_rows)
        st = m.stats()
        assert _same(got, ref)
        assert st.rays_traced == rays and st.kernel_ms > 0
        # only one of the two framebuffers, and a second frame through the same buffers
        only8, none = m.render(cam, params, strip_rows, want_radiance=False)
        assert none is None and np.array_equal(only8, ref[0])
        dbg = lib.render_params(w, h, 1, 1, shade_mode=T.SHADE_NORMAL_DEBUG)
        assert _same(m.render(cam, dbg, strip_rows), tracer.render(cam, dbg))


def test_one_process_device_outputs_and_errors(tracer):
    import torch
    scene = scenes.cornell_wahoo_scene()
    cam = _inside(200, 120)
    params = lib.render_params(200, 120, 4, 2)
    tracer.upload_scene(scene)
    ref = tracer.render(cam, params)
    d8 = torch.zeros((120, 200, 3), dtype=torch.uint8, device="cuda")
    dr = torch.zeros((120, 200, 3), dtype=torch.float32, device="cuda")
    with lib.MultiTracer([0, 0]) as m:
        with pytest.raises(lib.FireflyError) as e:
            m.render(cam, params)
        assert e.value.status == T.FF_ERR_NO_SCENE
        m.upload_scene(scene)
        torch.cuda.synchronize()  # the states run on their own non-blocking streams: the zero fills above must have landed
        m.render_device(cam, params, 8, d8.data_ptr(), dr.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(d8.cpu().numpy(), ref[0]) and np.array_equal(dr.cpu().numpy().view(np.uint32), ref[1].view(np.uint32))
    with pytest.raises(lib.FireflyError) as e:
        lib.MultiTracer([0, 99])
    assert e.value.status == T.FF_ERR_INVALID_ARG


def test_rccl_transport_on_a_one_rank_communicator(monkeypatch):
    """ncclCommInitRank(1 rank) + FF_DIST_SELF_LOOP: rank 0 packs its strips, ncclSend / ncclRecv them to itself inside one
    group, and scatters the received buffer — the code every rank of a real job runs, on the transport it runs it on."""
    monkeypatch.setenv("FF_DIST_SELF_LOOP", "1")
    scene = scenes.cornell_wahoo_scene()
    w, h = 320, 188
    cam = _inside(w, h)
    params = lib.render_params(w, h, 6, 3, seed=5)
    with lib.Tracer(0) as t:
        t.upload_scene(scene)
        ref = t.render(cam, params)
        with pytest.raises(lib.FireflyError) as e:
            t.render_distributed(cam, params)
        assert e.value.status == T.FF_ERR_INVALID_ARG  # no communicator yet
        t.dist_init(0, 1, lib.dist_unique_id())
        for strip_rows in (0, 4, 16):
            got = t.render_distributed(cam, params, strip_rows)
            assert _same(got, ref), strip_rows
        assert t.stats().rays_traced > 0
        # A rank that fails before the gather (here: injected) is an error of the frame, not a stuck receive: the ranks agree on a
        # status first (a 4-byte all-reduce behind the strips), nobody posts the gather, and the communicator stays usable.
        t.debug_dist_fail_rank(0)
        with pytest.raises(lib.FireflyError) as e:
            t.render_distributed(cam, params)
        assert e.value.status == T.FF_ERR_OOM and "injected" in str(e.value)
        t.debug_dist_fail_rank(-1)
        assert _same(t.render_distributed(cam, params, 8), ref)
        # invalid arguments on a joined rank go through the same agreement
        bad = lib.render_params(w, h, 0, 3)
        with pytest.raises(lib.FireflyError) as e:
            t.render_distributed(cam, bad)
        assert e.value.status == T.FF_ERR_INVALID_ARG
        assert _same(t.render_distributed(cam, params), ref)
        t.dist_shutdown()
        with pytest.raises(lib.FireflyError):
            t.render_distributed(cam, params)


def test_two_processes_share_the_gpu(tmp_path):
    """One process per rank (the bench's launch shape), both on cuda:0, strips gathered over gloo: tools/multi_rank_check.py
    compares the gathered frame with the single-launch frame on rank 0."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "multi_rank_check.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), outs
    assert "multi-rank image identical to single launch: True world 2" in outs[0]
