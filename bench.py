#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X path-tracing core.

Metric (BASELINE.json): Mrays/s and ms/frame at 1920x1080, 8 bounces, 1024 spp on the wahoo+cube Cornell box
(`configs[1]`), 1/2/4/8 GPUs.  One "step" = one full frame (every pixel, every sample) through the hot path;
inputs (scene, camera) are resident in HBM before the timed region; outputs stay on the device (with N > 1 the
finished strips are gathered to rank 0 over RCCL inside the timed region).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A ray = one closest-hit query (one path segment), counted on the device.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
NODE_BYTES, TRI_BYTES, RAY_BYTES = 112, 48, 24  # algorithmic bytes per 4-wide BVH node visit / triangle test / ray
# VALU issue roof (same guide, "Wave scheduling" + cycle constants): 256 CUs x 4 SIMDs, one wave64 VALU instruction per
# 2 cycles per SIMD, 2.4 GHz -> 1228.8 G wave-instructions/s
SIMDS, CLOCK_HZ, VALU_CYCLES_PER_INST = 1024, 2.4e9, 2.0
VALU_PEAK_GINST = SIMDS * CLOCK_HZ / VALU_CYCLES_PER_INST / 1e9


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--camera", choices=["inside", "default"], default="inside",
                    help="inside: (0,0,2.4) looking down -z, every primary ray hits the box (headline); "
                         "default: the reference's literals kernel.cu:312-321 (camera 12.5 units outside, box covers ~4%% of the frame)")
    ap.add_argument("--trace-mode", choices=["bvh", "brute"], default="bvh")
    ap.add_argument("--scene", choices=["c2", "c3", "c4"], default="c2",
                    help="c2: BASELINE configs[1], the headline (wahoo + cube Cornell box); c3: configs[2] (rocketman blooper scene, its own "
                         "camera); c4: configs[3] (983 040-triangle sphere in the box) - for profiles of the other configs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--spp-per-launch", type=int, default=0)
    ap.add_argument("--no-companion", action="store_true", help="skip the untimed default-camera frame")
    ap.add_argument("--keep-primary-hits", action="store_true",
                    help="let the library keep the pixels' stored primary hits from frame to frame (the viewer with its camera at rest); "
                         "by default every timed frame pays for its own pre-pass")
    return ap.parse_args()


def algorithmic_bytes(st, width, rows, spp):
    """sum over rays of (node bytes x nodes visited + triangle bytes x triangles tested + ray) plus
    the framebuffer: one float4 per pixel and sample block written by the trace kernel and read by the combine pass,
    float3 radiance + rgb8 out."""
    block_spp = 64 * ((spp + 1023) // 1024)
    blocks = (spp + block_spp - 1) // block_spp
    fb = (12 + 3 + 32 * blocks) * width * rows
    return st.rays_traced * RAY_BYTES + st.nodes_visited * NODE_BYTES + st.tris_tested * TRI_BYTES + fb


def measured_pmc(args, world, kernel):
    """(measurement or None, why not, scope): with N > 1 ranks there is no N-GPU counter file (the pool's boxes have one GPU); a rank
    runs the same kernel instantiation on a strip subset of the same frame, so the N = 1 file's PER-RAY figures apply when kernel
    name and source hash match: scope "n1" (the live rays and launch duration are this run's own)."""
    found, why = _measured_pmc(args, world, kernel)
    if found is not None or world == 1:
        return found, why, "exact"
    found, why1 = _measured_pmc(args, 1, kernel)
    return (found, None, "n1") if found is not None else (None, why, None)


def _measured_pmc(args, world, kernel):
    """The committed PMC measurement of the dominant kernel (profiles/*_pmc.json: rocprofv3 --pmc, separate passes,
    tools/pmc_passes.sh + tools/pmc_to_json.py) taken on exactly this workload and kernel instantiation AND on the kernel
    sources of this tree (kernel_source_hash: a counter file from other sources says nothing about this kernel).  Returns
    (measurement or None, why not)."""
    from gpupathtracer_amd.provenance import kernel_source_hash
    here = kernel_source_hash()
    best, stale = None, None
    pdir = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if not name.endswith("_pmc.json"):
            continue
        try:
            with open(os.path.join(pdir, name)) as f:
                d = json.load(f)
            w = d["workload"]
            if (w["width"], w["height"], w["bounces"], w["spp"], w["n_gpus"], w["camera"], w["trace"], w.get("scene", "c2")) == (
                    args.width, args.height, args.bounces, args.spp, world, args.camera, args.trace_mode, args.scene) and \
                    d["kernel"].replace(" ", "") == kernel.replace(" ", ""):
                if d.get("kernel_source_hash") == here:
                    best = dict(d, file="profiles/" + name)
                else:
                    stale = f"profiles/{name} was taken on other kernel sources (hash {d.get('kernel_source_hash', 'none')}, this tree {here})"
        except (OSError, KeyError, ValueError):
            continue
    if best is not None:
        return best, None
    return None, stale or "no committed PMC measurement (profiles/*_pmc.json) matches this workload and kernel instantiation"


def host_cores():
    """Cores this process may use: the affinity mask, capped by the cgroup's CPU quota where one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(scene, camera, args):
    """The CPU oracle (a from-scratch port of the reference's brute-force loop; the reference has no CPU path) timed on
    a bounded window of the same workload on this box's host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import oracle_render
    from gpupathtracer_amd import lib

    threads = host_cores()  # all host cores this process may use
    w, h, spp = 192, 108, 12  # (about 14 s of CPU work on 16 cores: three runs of ~3.3 s + the single-core sample)
    x0, y0 = (args.width - w) // 2, (args.height - h) // 2
    params = lib.render_params(args.width, args.height, args.bounces, spp, args.seed)
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        _, _, ctr = oracle_render(scene, camera, params, window=(x0, y0, w, h), threads=threads, want_counters=True)
        times.append(time.perf_counter() - t0)
    times.sort()
    dt = times[1]  # median of three
    # one core on a quarter of that window (SURVEY.md section 8d asks for both figures)
    w1, h1, spp1 = w // 2, h // 2, 4
    params1 = lib.render_params(args.width, args.height, args.bounces, spp1, args.seed)
    t1 = time.perf_counter()
    _, _, ctr1 = oracle_render(scene, camera, params1, window=((args.width - w1) // 2, (args.height - h1) // 2, w1, h1), threads=1, want_counters=True)
    dt1 = time.perf_counter() - t1
    return {
        "value": round(ctr.rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
        "sample": f"centre {w}x{h} window of the {args.width}x{args.height} frame, {args.bounces} bounces, {spp} spp, "
                  f"brute force over {scene.triangle_count} triangles: {ctr.rays} rays in {dt:.2f} s (median of 3; min {times[0]:.2f} s)",
        "best": round(ctr.rays / times[0] / 1e6, 4),
        "single_core": {"value": round(ctr1.rays / dt1 / 1e6, 4), "unit": "Mrays/s", "cores": 1,
                        "sample": f"centre {w1}x{h1} window of the same frame, {spp1} spp: {ctr1.rays} rays in {dt1:.2f} s"},
    }


def main():
    args = parse_args()
    # Every timed frame does ALL its work: the library would otherwise keep the pixels' stored primary hits from one frame to the next
    # (same camera, same scene: a viewer at rest) and spare the later frames their 0.17 ms pre-pass.
    if not args.keep_primary_hits:
        os.environ.setdefault("FF_NO_PRIMARY_CACHE", "1")
    import torch
    import torch.distributed as dist

    from gpupathtracer_amd import dist as ffdist
    from gpupathtracer_amd import lib, scenes
    from gpupathtracer_amd import types as T

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    # FF_DIST_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks (ranks share devices and the gather goes
    # through host memory); the real multi-GPU run uses nccl (= RCCL over xGMI), one rank per GPU.
    backend = os.environ.get("FF_DIST_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)

    def barrier_sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- inputs: resident before the timed region ----
    scene = {"c2": scenes.cornell_wahoo_scene, "c3": scenes.blooper_scene, "c4": scenes.sphere_stress_scene}[args.scene]()
    scene_name = {"c2": "C2 wahoo.obj+cube.obj Cornell box (5184 triangles, 6 planes)", "c3": "C3 rocketman.obj+cube.obj blooper scene (6048 triangles, 2 planes)",
                  "c4": "C4 983040-triangle sphere in the Cornell box (6 planes)"}[args.scene]
    if args.camera == "default":
        camera = scenes.default_camera(args.width, args.height)
    elif args.scene == "c3":
        camera = scenes.posed_camera(args.width, args.height, position=(4.0, 1.0, 7.0), yaw=-118.0, pitch=-8.0)  # the mesh is visible past the planes
    else:
        camera = scenes.posed_camera(args.width, args.height, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
    mode = T.TRACE_BVH if args.trace_mode == "bvh" else T.TRACE_BRUTE_FORCE
    params = lib.render_params(args.width, args.height, args.bounces, args.spp, args.seed, mode, T.SHADE_DIFFUSE_PATH,
                               T.GRID_FULL, args.spp_per_launch)
    tracer = lib.Tracer(dev_index)
    tracer.upload_scene(scene)
    tracer.set_stream(torch.cuda.current_stream().cuda_stream)

    strip_rows = lib.dist_strip_rows(world, args.height) if world > 1 else args.height
    local_rows = tracer.strips_local_rows(args.height, strip_rows, rank, world)
    # N > 1: every rank renders its strips, rank 0 gathers them behind the C ABI (ff_render_distributed: packed strips, one
    # grouped ncclSend / ncclRecv over RCCL, one scatter kernel).  torch.distributed only carries the 128-byte RCCL id here.
    gather = "none"
    gather_note = None

    def all_agree(flag):
        """True iff `flag` holds on every rank (torch's own communicator; every rank must call this the same number of times)."""
        t = torch.tensor([1 if flag else 0], device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item()) == 1

    full_rgb8 = torch.empty((args.height, args.width, 3), dtype=torch.uint8, device=device) if rank == 0 else None
    full_rad = torch.empty((args.height, args.width, 3), dtype=torch.float32, device=device) if rank == 0 else None

    def native_self_check():
        """One short frame through the native gather against the same frame rendered by rank 0 alone, bit for bit (rows in the
        wrong place, a wrong part offset or a lost message cannot hide in a rate)."""
        vparams = lib.render_params(args.width, args.height, args.bounces, min(args.spp, 8), args.seed, mode, T.SHADE_DIFFUSE_PATH, T.GRID_FULL, 0)
        tracer.render_distributed_device(camera, vparams, strip_rows, full_rgb8.data_ptr() if rank == 0 else None,
                                         full_rad.data_ptr() if rank == 0 else None)
        if rank != 0:
            return True
        torch.cuda.synchronize()
        got8, gotr = full_rgb8.clone(), full_rad.clone()
        tracer.render_device(camera, vparams, full_rgb8.data_ptr(), full_rad.data_ptr())
        torch.cuda.synchronize()
        same = bool(torch.equal(got8, full_rgb8)) and bool(torch.equal(gotr.view(torch.int32), full_rad.view(torch.int32)))
        if not same:
            print("[bench] rank 0: the natively gathered frame differs from the frame rendered alone", file=sys.stderr, flush=True)
        return same

    if world > 1 and backend == "nccl":
        os.environ.setdefault("FF_DIST_TIMEOUT_S", "120")  # a stuck native gather becomes an error after two minutes, not a hang

        def broadcast(obj):
            box = [obj]
            dist.broadcast_object_list(box, src=0)
            return box[0]

        # (the decision tree - every rank runs every collective whatever failed locally - lives in gpupathtracer_amd/dist.py, where
        # tests/test_dist_gloo.py drives it over gloo with a stubbed library for each failure shape)
        gather, gather_note = ffdist.negotiate_native_gather(
            rank, world, able=lib.dist_available(), make_id=lib.dist_unique_id, join=lambda uid: tracer.dist_init(rank, world, uid),
            leave=tracer.dist_shutdown, self_check=native_self_check, broadcast=broadcast, all_agree=all_agree,
            log=lambda text: print("[bench] " + text, file=sys.stderr, flush=True))
    elif world > 1:
        gather = "torch-" + backend
    rgb8 = rad = None
    if gather != "native-rccl":
        rgb8 = torch.empty((max(local_rows, 1), args.width, 3), dtype=torch.uint8, device=device)
        rad = torch.empty((max(local_rows, 1), args.width, 3), dtype=torch.float32, device=device)
    torch.cuda.synchronize()

    def step(cam=None):
        cam = camera if cam is None else cam
        if world == 1:
            tracer.render_device(cam, params, full_rgb8.data_ptr(), full_rad.data_ptr())
        elif gather == "native-rccl":
            tracer.render_distributed_device(cam, params, strip_rows, full_rgb8.data_ptr() if rank == 0 else None,
                                             full_rad.data_ptr() if rank == 0 else None)
        else:
            tracer.render_strips_device(cam, params, strip_rows, rank, world, rgb8.data_ptr(), rad.data_ptr())
            if backend == "nccl":
                g8 = ffdist.gather_strips(rgb8[:local_rows], args.height, strip_rows, rank, world, dist)
                gr = ffdist.gather_strips(rad[:local_rows], args.height, strip_rows, rank, world, dist)
            else:
                g8 = ffdist.gather_strips(rgb8[:local_rows].cpu(), args.height, strip_rows, rank, world, dist)
                gr = ffdist.gather_strips(rad[:local_rows].cpu(), args.height, strip_rows, rank, world, dist)
                if rank == 0:
                    g8, gr = g8.to(device), gr.to(device)
            if rank == 0:
                tracer.deinterleave_strips(g8.data_ptr(), full_rgb8.data_ptr(), args.width, args.height, strip_rows, world, 3)
                tracer.deinterleave_strips(gr.data_ptr(), full_rad.data_ptr(), args.width, args.height, strip_rows, world, 12)
        return tracer.stats()

    # ---- warmup (untimed); the first warmup frame also collects the node/triangle visit counters ----
    tracer.set_collect_stats(True)
    counted = step()
    tracer.set_collect_stats(False)
    for _ in range(max(0, args.warmup - 1)):
        step()

    # ---- timed region: exactly K steps, barrier + synchronize on both sides ----
    barrier_sync()
    t0 = time.perf_counter()
    rays = 0
    answered = 0
    cut_short = 0
    kernel_ms = 0.0
    launches = 0
    frame_ms = []
    for _ in range(args.steps):
        tf = time.perf_counter()
        st = step()
        frame_ms.append((time.perf_counter() - tf) * 1e3)
        rays += st.rays_traced
        answered += st.rays_answered
        cut_short += st.rays_cut_short
        kernel_ms += st.kernel_ms
        launches += st.kernel_launches
    barrier_sync()
    elapsed = time.perf_counter() - t0
    kernel = tracer.kernel_name()

    if world > 1:
        red_dev = device if backend == "nccl" else torch.device("cpu")
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        agg = torch.tensor([float(rays), float(answered), float(cut_short)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        total_rays, total_answered, total_cut = agg[0].item(), agg[1].item(), agg[2].item()
    else:
        total_rays, total_answered, total_cut = float(rays), float(answered), float(cut_short)

    # ---- companion figure (untimed region, N = 1 only): the same frame from the reference's default camera ----
    companion = None
    if world == 1 and args.camera == "inside" and args.scene == "c2" and not args.no_companion:
        dcam = scenes.default_camera(args.width, args.height)
        step(dcam)
        torch.cuda.synchronize()
        tc = time.perf_counter()
        st2 = step(dcam)
        torch.cuda.synchronize()
        dtc = time.perf_counter() - tc
        companion = {"camera": "default (kernel.cu:312-321: 12.5 units outside the box, ~96 % primary misses)",
                     "value": round(st2.rays_traced / dtc / 1e6, 2), "unit": "Mrays/s", "ms_per_frame": round(dtc * 1e3, 3),
                     "rays_per_frame": int(st2.rays_traced), "frames": 1}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_rays / elapsed / 1e6
        # The dominant kernel on this rank: mean launch duration from the HIP events the library records on its launch stream.
        mean_launch_s = kernel_ms / max(1, launches) / 1e3
        rays_per_launch = rays / max(1, launches)
        algo_bytes_launch = algorithmic_bytes(counted, args.width, local_rows, args.spp) / max(1, counted.kernel_launches)
        pmc, pmc_why_not, pmc_scope = measured_pmc(args, world, kernel)
        roof = {
            # What binds this kernel is VALU issue at partial lane occupancy, not HBM: the scene is
            # LDS/L2 resident.  achieved = wave-level VALU instructions per second, from the committed PMC count per ray of
            # this kernel instantiation on this workload x the rays and the launch duration measured live in this run.
            "bound": "valu_issue", "unit": "G wave-inst/s", "peak": round(VALU_PEAK_GINST, 1),
            "kernel": kernel, "kernel_ms_per_launch": round(mean_launch_s * 1e3, 3), "rays_per_launch": int(rays_per_launch),
            "algorithmic_bytes_per_launch": int(algo_bytes_launch),
            "algorithmic_GBps": round(algo_bytes_launch / mean_launch_s / 1e9, 1),
            "per_ray": {"nodes": round(counted.nodes_visited / max(1, counted.rays_traced), 3),
                        "tris": round(counted.tris_tested / max(1, counted.rays_traced), 3)},
            # the same per path segment that went through the closest-hit machinery (all segments minus those answered from a
            # block's parked primary hit or dropped with a culled pixel); `valu` joins it below when a PMC file matches
            "per_traversed_ray": {"nodes": round(counted.nodes_visited / max(1, counted.rays_traced - counted.rays_answered), 3),
                                  "tris": round(counted.tris_tested / max(1, counted.rays_traced - counted.rays_answered), 3)},
        }
        if pmc is not None:
            achieved = pmc["valu_insts_per_ray"] * rays_per_launch / mean_launch_s / 1e9
            traffic = pmc["hbm_bytes_per_ray"] * rays_per_launch
            roof.update({
                "achieved": round(achieved, 1), "frac": round(achieved / VALU_PEAK_GINST, 4),
                "lane_occupancy": pmc["lane_occupancy"], "fp32_lane_throughput_frac": round(achieved / VALU_PEAK_GINST * pmc["lane_occupancy"], 4),
                "traffic": int(traffic), "hbm_GBps": round(traffic / mean_launch_s / 1e9, 2),
                "hbm_frac": round(traffic / mean_launch_s / 1e9 / HBM_PEAK_GBS, 5),
                "algorithmic_vs_hbm": round(algo_bytes_launch / max(traffic, 1.0), 1),
                "pmc_source": pmc["file"], "pmc_kernel_source_hash": pmc["kernel_source_hash"], "pmc_scope": pmc_scope,
                "note": "algorithmic bytes (SURVEY.md section 8d: 24/ray + 112/node visit + 48/triangle test + framebuffer) are served by LDS and L2; "
                        "`traffic` is what reaches HBM (FETCH_SIZE x2 + WRITE_SIZE from the PMC passes, scaled by rays)",
            })
            roof["per_ray"]["valu"] = round(pmc["valu_insts_per_ray"], 2)
            roof["per_traversed_ray"]["valu"] = round(pmc["valu_insts_per_ray"] * total_rays / max(1.0, total_rays - total_answered), 2)
        else:
            roof.update({"achieved": None, "frac": None, "traffic": None, "note": "PMC-derived fields withheld: " + pmc_why_not})
        frame_ms.sort()
        out = {
            "metric": f"Mrays/s (path segments = closest-hit queries, device-counted) at {'1080p' if (args.width, args.height) == (1920, 1080) else f'{args.width}x{args.height}'}, "
                      f"{args.bounces} bounces, {args.spp} spp",
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "ms_per_frame": round(ms_per_step, 3), "frames_per_s": round(1e3 / ms_per_step, 4),
            "ms_per_frame_min": round(frame_ms[0], 3), "ms_per_frame_median": round(frame_ms[len(frame_ms) // 2], 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # `value` counts ALL path segments, as in every round.  Of those, `untraversed_frac` never reached the closest-hit machinery
            # (repeated primary rays of a sample block answered from its parked hit - kernel.cu:200-205 has no jitter - and the rays of
            # culled pixels); `traversed_Mrays_per_s` is the rate of the rest.  ms/frame is the number to compare across rounds.
            "traversed_Mrays_per_s": round((total_rays - total_answered) / elapsed / 1e6, 2),
            "untraversed_frac": round(total_answered / max(1.0, total_rays), 4),
            "config": {
                "workload": f"{scene_name}, {args.width}x{args.height}, "
                            f"{args.bounces} bounces, {args.spp} spp, camera={args.camera}, trace={args.trace_mode}, seed {args.seed}",
                "scene": args.scene,
                "rays_per_frame": int(total_rays / args.steps),
                # rays_untraversed: path segments per frame answered without a traversal (top level: untraversed_frac);
                # rays_cut_last: last segments of paths that ended after the analytic records were screened (only an emitter can still
                # add radiance there and the query held none: the meshes were not walked); rays_any_hit: last segments that held an
                # emitter and stopped at the first certain occluder in front of it
                "rays_untraversed": int(total_answered / args.steps),
                "rays_cut_last": int(total_cut / args.steps),
                "primary_hits_kept_between_frames": os.environ.get("FF_NO_PRIMARY_CACHE") is None,
                "partition": f"{strip_rows}-row strips round-robin over {world} rank(s)",
                "gather": gather if gather_note is None else f"{gather} ({gather_note})",
            },
            "roofline": roof,
        }
        if companion is not None:
            out["default_camera"] = companion
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, camera, args)
        print(json.dumps(out), flush=True)

    tracer.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
