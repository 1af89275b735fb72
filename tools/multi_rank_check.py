#!/usr/bin/env python3
"""Rehearsal of the N > 1 bench path on a box with one GPU: W ranks (gloo), each rendering its strips on cuda:0, gather to
rank 0, de-interleave on the device, and compare with the single-launch frame bit for bit.
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/multi_rank_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from gpupathtracer_amd import dist as ffdist, lib, scenes

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
torch.cuda.set_device(0)
W, H = 640, 360 + 8
scene = scenes.cornell_wahoo_scene()
cam = scenes.posed_camera(W, H, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
params = lib.render_params(W, H, 6, 4)
t = lib.Tracer(0)
t.upload_scene(scene)
rows = t.strips_local_rows(H, ffdist.STRIP_ROWS, rank, world)
rgb8 = torch.empty((rows, W, 3), dtype=torch.uint8, device="cuda")
rad = torch.empty((rows, W, 3), dtype=torch.float32, device="cuda")
t.render_strips_device(cam, params, ffdist.STRIP_ROWS, rank, world, rgb8.data_ptr(), rad.data_ptr())
g8 = ffdist.gather_strips(rgb8.cpu(), H, ffdist.STRIP_ROWS, rank, world, dist)
gr = ffdist.gather_strips(rad.cpu(), H, ffdist.STRIP_ROWS, rank, world, dist)
if rank == 0:
    full8 = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
    fullr = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    g8, gr = g8.cuda(), gr.cuda()
    t.deinterleave_strips(g8.data_ptr(), full8.data_ptr(), W, H, ffdist.STRIP_ROWS, world, 3)
    t.deinterleave_strips(gr.data_ptr(), fullr.data_ptr(), W, H, ffdist.STRIP_ROWS, world, 12)
    ref8, refr = t.render(cam, params)
    ok = np.array_equal(full8.cpu().numpy(), ref8) and np.array_equal(fullr.cpu().numpy().view(np.uint32), refr.view(np.uint32))
    print("multi-rank image identical to single launch:", ok, "world", world)
    assert ok
t.close()
dist.destroy_process_group()
