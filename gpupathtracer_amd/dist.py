"""Multi-GPU frame rendering: one process per GPU, image strips dealt round-robin to ranks, one framebuffer gather.

The path shards with no exchange during rendering (pixels are independent, the scene is replicated, and the RNG is
keyed on the global pixel index), so the only collective is the gather of the finished strips to rank 0 — the rank
that would own the GL pixel buffer in the viewer.  `torch.distributed` is plumbing here: backend "nccl" is RCCL over
xGMI on the GPU box, "gloo" on CPU for the tests.
"""
import numpy as np

STRIP_ROWS = 16  # the reference's block height (kernel.cu:306); strips of 16 rows are dealt round-robin to ranks


def strip_rows_for(world_size, height=1080):
    """Strip height used by the benchmark (mirrors ff_dist_strip_rows_for): of 1 .. 16 rows the one whose largest part has the
    fewest rows, the thinnest such of at least two rows (1080 rows: 2-row strips for 2 and 4 ranks, 3-row strips for 8: equal shares)."""
    if world_size <= 1 or height <= 0:
        return 16
    best, best_rows = 16, None
    for s in range(16, 0, -1):
        worst = max(strip_layout(height, s, world_size))
        if best_rows is None or worst < best_rows or (worst == best_rows and s >= 2):
            best, best_rows = s, worst
    return best


def strip_layout(height, strip_rows, num_parts):
    """rows owned by each part, in part order (mirrors ff_strips_local_rows)."""
    nstrips = (height + strip_rows - 1) // strip_rows
    rows = []
    for part in range(num_parts):
        r = 0
        for s in range(part, nstrips, num_parts):
            y0 = s * strip_rows
            r += strip_rows if y0 + strip_rows <= height else height - y0
        rows.append(r)
    return rows


def strip_row_indices(height, strip_rows, part, num_parts):
    """global row index of every local row of `part` (in local order)."""
    nstrips = (height + strip_rows - 1) // strip_rows
    idx = []
    for s in range(part, nstrips, num_parts):
        y0 = s * strip_rows
        idx.extend(range(y0, min(y0 + strip_rows, height)))
    return np.asarray(idx, dtype=np.int64)


def gather_strips(local, height, strip_rows, rank, world_size, dist=None, dst=0):
    """Gather every rank's compact strip block (torch tensor [local_rows, W, C]) to `dst`.

    Returns on `dst` a tensor [sum(rows), W, C] holding the parts' blocks back to back in part order (the layout
    ff_deinterleave_strips expects), on other ranks None.  Blocks are padded to the largest part for the collective.
    """
    import torch

    rows = strip_layout(height, strip_rows, world_size)
    if world_size == 1:
        return local
    max_rows = max(rows)
    padded = local
    if local.shape[0] != max_rows:
        padded = torch.zeros((max_rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        padded[: local.shape[0]] = local
    if rank == dst:
        bucket = [torch.empty_like(padded) for _ in range(world_size)]
        dist.gather(padded, gather_list=bucket, dst=dst)
        return torch.cat([bucket[p][: rows[p]] for p in range(world_size)], dim=0)
    dist.gather(padded, gather_list=None, dst=dst)
    return None


def deinterleave_host(blocks, height, strip_rows, num_parts):
    """CPU/torch twin of ff_deinterleave_strips for tests: [sum(rows), W, C] part-major -> [H, W, C] image order."""
    import torch

    out = torch.empty((height,) + tuple(blocks.shape[1:]), dtype=blocks.dtype, device=blocks.device)
    off = 0
    for part in range(num_parts):
        idx = torch.from_numpy(strip_row_indices(height, strip_rows, part, num_parts)).to(blocks.device)
        out[idx] = blocks[off: off + len(idx)]
        off += len(idx)
    return out


def negotiate_native_gather(rank, world_size, *, able, make_id, join, leave, self_check, broadcast, all_agree, log=None):
    """Which gather does a multi-rank job use: the library's own RCCL transport (ff_render_distributed) or torch.distributed's?

    Every rank calls this with the same `world_size` and runs EVERY collective below whatever failed locally - a rank that skipped
    one would leave the others inside it.  Order: (1) all ranks agree that RCCL can be loaded by the library (`able`) and that rank
    0 produced a communicator id (`make_id()`, broadcast to all; None when it could not); (2) only then does any rank enter
    `join(id)` (ff_dist_init blocks until every rank has), and all agree that it worked; (3) `self_check()` - one short frame
    through the native gather compared bitwise, on rank 0, with the same frame rendered by rank 0 alone - and all agree on its
    result.  Any failure on any rank sends ALL ranks to torch's gather; a rank that had joined leaves (`leave()`).

    Callables: make_id() -> bytes (rank 0 only; may raise), join(id) (may raise), leave() (errors ignored), self_check() -> bool
    (may raise), broadcast(obj) -> obj of rank 0, all_agree(flag) -> True iff flag holds on every rank, log(str).
    Returns (gather, note): ("native-rccl", None) or ("torch-rccl", why).
    """
    say = log if log is not None else (lambda text: None)
    uid = None
    if rank == 0 and able:
        try:
            uid = make_id()
        except Exception as e:  # noqa: BLE001
            say(f"rank 0: ff_dist_unique_id failed ({e})")
    uid = broadcast(uid)  # (None when rank 0 has no id to give)
    if not all_agree(bool(able) and uid is not None):
        return "torch-rccl", "RCCL could not be loaded by the library on some rank, or rank 0 could not make a communicator id"
    joined = False
    try:
        join(uid)
        joined = True
    except Exception as e:  # noqa: BLE001
        say(f"rank {rank}: ff_dist_init failed ({e})")
    if not all_agree(joined):
        if joined:
            try:
                leave()
            except Exception:  # noqa: BLE001
                pass
        return "torch-rccl", "ff_dist_init failed on some rank"
    good = False
    try:
        good = bool(self_check())
    except Exception as e:  # noqa: BLE001
        say(f"rank {rank}: native gather failed its self-check ({e})")
    if not all_agree(good):
        try:
            leave()
        except Exception:  # noqa: BLE001
            pass
        return "torch-rccl", "the native RCCL gather failed its bitwise self-check against a frame rendered by rank 0 alone"
    return "native-rccl", None
