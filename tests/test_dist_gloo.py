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


def test_default_strip_height_balances_the_parts(ff):
    """ff_dist_strip_rows_for (csrc/ff_dist.cpp default_strip_rows; mirrored by dist.strip_rows_for): the strip height - 1 .. 16 rows -
    whose largest part has the fewest rows, the thinnest such of at least two rows.  The slowest rank sets the frame time: 1080 rows
    over 2, 4 or 8 ranks come out at equal shares (2-row strips for 2 and 4 ranks, 3-row strips for 8: 135 rows each, not 136 and 132)."""
    lib = ff.load()
    for h in (1080, 2160, 720, 800, 1, 17, 1000):
        for n in (1, 2, 3, 4, 5, 8):
            s = lib.ff_dist_strip_rows_for(h, n)
            assert s == ffdist.strip_rows_for(n, h) and 1 <= s <= 16
            worst = max(ffdist.strip_layout(h, s, n))
            assert worst == min(max(ffdist.strip_layout(h, t, n)) for t in range(1, 17)), (h, n, s)
    assert [lib.ff_dist_strip_rows_for(1080, n) for n in (2, 4, 8)] == [2, 2, 3] and lib.ff_dist_strip_rows(8) == 3
    assert ffdist.strip_layout(1080, 3, 8) == [135] * 8 and ffdist.strip_layout(1080, 2, 4) == [270] * 4


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


# ---- bench.py's native / torch gather decision (gpupathtracer_amd.dist.negotiate_native_gather) with a stubbed library ----

def _negotiate_worker(rank, world, port, shape, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []
    try:
        def broadcast(obj):
            box = [obj]
            dist.broadcast_object_list(box, src=0)
            return box[0]

        def all_agree(flag):
            t = torch.tensor([1 if flag else 0])
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return int(t.item()) == 1

        def make_id():
            calls.append("make_id")
            if shape == "no_id":
                raise RuntimeError("rccl: no id for you")
            return b"\x07" * 128

        def join(uid):
            calls.append("join")
            assert uid == b"\x07" * 128
            if shape == "init_fails_on_1" and rank == 1:
                raise RuntimeError("ncclCommInitRank failed")

        def leave():
            calls.append("leave")

        def self_check():
            calls.append("self_check")
            if shape == "self_check_differs":
                return rank != 0  # rank 0 sees a frame that differs from its own
            if shape == "self_check_raises_on_1" and rank == 1:
                raise RuntimeError("FF_ERR_COMM")
            return True

        able = not (shape == "cannot_load_on_1" and rank == 1)
        gather, note = ffdist.negotiate_native_gather(rank, world, able=able, make_id=make_id, join=join, leave=leave, self_check=self_check,
                                                      broadcast=broadcast, all_agree=all_agree)
        # every rank must still be able to run a collective afterwards: nobody is stuck inside one
        t = torch.tensor([rank + 1])
        dist.all_reduce(t)
        with open(os.path.join(out_dir, f"r{rank}"), "w") as f:
            f.write(f"{gather}|{note}|{','.join(calls)}|{int(t.item())}")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("shape", ["all_well", "no_id", "cannot_load_on_1", "init_fails_on_1", "self_check_differs", "self_check_raises_on_1"])
def test_native_gather_negotiation_leaves_no_rank_behind(tmp_path, shape):
    """the decision tree bench.py runs before anything is timed, over gloo with a stubbed library, for the
    failure shapes a first multi-GPU run can meet: rank 0 has no communicator id; one rank cannot load RCCL; one rank fails
    ff_dist_init; the self-check frame differs (or raises) on one rank.  In every shape all ranks take the SAME decision, a rank
    that had joined leaves, nobody enters ff_dist_init unless all can, and every rank reaches the collective that follows."""
    world = 3
    port = _free_port()
    mp.spawn(_negotiate_worker, args=(world, port, shape, str(tmp_path)), nprocs=world, join=True)
    got = [(tmp_path / f"r{r}").read_text().split("|") for r in range(world)]
    decisions = {g[0] for g in got}
    assert len(decisions) == 1 and all(g[3] == str(sum(range(1, world + 1))) for g in got)
    decision = decisions.pop()
    calls = [g[2].split(",") if g[2] else [] for g in got]
    if shape == "all_well":
        assert decision == "native-rccl" and all(g[1] == "None" for g in got)
        assert all(c[-2:] == ["join", "self_check"] and "leave" not in c for c in calls)
    else:
        assert decision == "torch-rccl" and all(g[1] != "None" for g in got)
    if shape in ("no_id", "cannot_load_on_1"):
        assert all("join" not in c for c in calls)  # nobody entered ff_dist_init
    if shape == "init_fails_on_1":
        assert "leave" in calls[0] and "leave" in calls[2] and "leave" not in calls[1] and all("self_check" not in c for c in calls)
    if shape in ("self_check_differs", "self_check_raises_on_1"):
        assert all("leave" in c for c in calls)
