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


@pytest.mark.parametrize("world,height,width,strip_rows", [(2, 1080, 64, 16), (3, 50, 17, 16), (2, 8, 8, 16)])
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
