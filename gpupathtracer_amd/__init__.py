"""gpupathtracer_amd — MI355X-native (gfx950) path-tracing core behind the FireflyEngine host API.

The product is the C-ABI library ``libfirefly_hip.so`` (include/firefly/ff_api.h); this package is the thin
host-side mirror used by the tests, the benchmark and multi-GPU runs:

  types   ctypes mirrors of the reference's structs (include/firefly/ff_types.h)
  lib     ctypes binding + ``Tracer`` (upload once, render per frame)
  scenes  scene presets for the BASELINE configs
  dist    one-process-per-GPU strip partition + framebuffer gather over torch.distributed (RCCL)
"""
from . import types  # noqa: F401

__all__ = ["types", "lib", "scenes", "dist"]
