import sys, ctypes as C
sys.path.insert(0, '.')
from gpupathtracer_amd import lib, types as T
l = lib.load()
st = C.c_void_p()
print("create", l.ff_create(C.byref(st), 0))
rc = l.ff_register_gl_pbo(st, 1, 64, 64)
print("register rc", rc, l.ff_last_error())
print("render_to_pbo rc", l.ff_render_to_pbo(st, None, None))
print("unregister rc", l.ff_unregister_gl_pbo(st))
l.ff_destroy(st)
