"""The N > 1 path on CPU: strip partition + framebuffer gather + de-interleave with world_size 2 and 3 over gloo.
(The GPU box runs the same code over RCCL; rendering itself is replaced here by a synthetic per-pixel pattern.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gpupathtracer_amd import dist as ffdist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _pattern(height, width):
    y, x = np.mgrid[0:height, 0:width]
    return np.stack([(y * width + x) % 251, y % 256, x % 256], axis=2).astype(np.uint8)


def _worker(rank, world, port, height, width, strip_rows, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = _pattern(height, width)
        rows = ffdist.strip_row_indices(height, strip_rows, rank, world)
        local8 = torch.from_numpy(full[rows].copy())                       # this rank's compact strips (rgb8)
        localf = local8.to(torch.float32) * 0.5                            # and a float3 "radiance" twin
        g8 = ffdist.gather_strips(local8, height, strip_rows, rank, world, dist)
        gf = ffdist.gather_strips(localf, height, strip_rows, rank, world, dist)
        if rank == 0:
            img8 = ffdist.deinterleave_host(g8, height, strip_rows, world)
            imgf = ffdist.deinterleave_host(gf, height, strip_rows, world)
            ok = bool((img8.numpy() == full).all()) and bool((imgf.numpy() == full.astype(np.float32) * 0.5).all())
            with open(os.path.join(out_dir, "result"), "w") as f:
                f.write("ok" if ok else "mismatch")
        else:
            assert g8 is None and gf is None
    finally:
        dist.destroy_process_group()


# (8 ranks over 20 rows in 4-row strips: ranks 5, 6 and 7 own nothing and still take part in the collective)
@pytest.mark.parametrize("world,height,width,strip_rows", [(2, 1080, 64, 16), (3, 50, 17, 16), (2, 8, 8, 16), (8, 20, 12, 4)])
def test_gather_and_deinterleave(tmp_path, world, height, width, strip_rows):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, height, width, strip_rows, str(tmp_path)), nprocs=world, join=True)
    assert (tmp_path / "result").read_text() == "ok"


def test_strip_layout_matches_library(ff):
    lib = ff.load()
    for h, sr, n in [(1080, 16, 8), (2160, 16, 8), (1080, 16, 2), (17, 16, 4), (1, 16, 1)]:
        assert ffdist.strip_layout(h, sr, n) == [lib.ff_strips_local_rows(h, sr, p, n) for p in range(n)]
        allrows = np.concatenate([ffdist.strip_row_indices(h, sr, p, n) for p in range(n)])
        assert sorted(allrows.tolist()) == list(range(h))


def test_gather_wire_layout_and_parts_without_rows(ff):
    """ff_render_distributed's messages (ff_dist_part_bytes): one per part that owns rows, at back-to-back offsets in rank 0's
    gather buffer; a part without rows sends nothing and rank 0 posts no receive for it - both sides apply the same predicate
    (bytes > 0), so a rank that owns nothing cannot leave a receive unmatched.  World 8 on a 20-row image: three such ranks."""
    import ctypes as C
    lib = ff.load()
    for w, h, sr, n in [(1920, 1080, 4, 8), (96, 20, 4, 8), (17, 50, 16, 3), (64, 64, 1, 4), (5, 1, 16, 2)]:
        off = C.c_longlong(-1)
        expect = 0
        senders = []
        for p in range(n):
            rows = lib.ff_strips_local_rows(h, sr, p, n)
            b = lib.ff_dist_part_bytes(w, h, sr, p, n, C.byref(off))
            pad16 = lambda v: (v + 15) & ~15
            assert b == pad16(rows * w * 12) + pad16(rows * w * 3) and off.value == expect
            assert (b > 0) == (rows > 0)
            expect += b
            if b > 0:
                senders.append(p)
        assert sum(lib.ff_strips_local_rows(h, sr, p, n) for p in range(n)) == h
        if (w, h, sr, n) == (96, 20, 4, 8):
            assert senders == [0, 1, 2, 3, 4]
    assert lib.ff_dist_part_bytes(8, 8, 0, 0, 1, None) == -1 and lib.ff_dist_part_bytes(8, 8, 4, 2, 2, None) == -1
