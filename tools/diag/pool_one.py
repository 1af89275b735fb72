import faulthandler, os, sys
faulthandler.dump_traceback_later(40, exit=True)
os.environ["FF_POOL"] = "1"
sys.path.insert(0, "/root/repo")
import numpy as np
from gpupathtracer_amd import lib, scenes
print("start", flush=True)
t = lib.Tracer(0)
print("created", flush=True)
t.upload_scene(scenes.cornell_wahoo_scene())
print("uploaded", flush=True)
cam = scenes.posed_camera(64, 36, position=(0.0, 0.0, 2.4), yaw=-90.0, pitch=0.0)
try:
    rgb8, rad = t.render(cam, lib.render_params(64, 36, 4, 2, 7))
    print("rendered", t.stats().rays_traced, t.kernel_name(), flush=True)
except Exception as e:
    print("render failed:", e, flush=True)
