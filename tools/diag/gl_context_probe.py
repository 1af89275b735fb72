#!/usr/bin/env python3
"""Can this box give us an OpenGL context to run the HIP-GL pixel-buffer path (utilities.h:605-618, kernel.cu:335-351) against?

Looks for every way a context could be made without a window system: EGL (device / surfaceless platform), GLX on an X
display, OSMesa, GBM; lists the DRM nodes; and calls hipGLGetDevices.  Prints one JSON object.  Read-only: nothing is
installed or configured."""
import ctypes as C
import ctypes.util
import glob
import json
import os


def try_dlopen(names):
    out = {}
    for n in names:
        try:
            C.CDLL(n)
            out[n] = "loaded"
        except OSError as e:
            out[n] = f"absent ({str(e).split(':')[-1].strip()})"
    return out


def main():
    rep = {"env": {k: os.environ.get(k) for k in ("DISPLAY", "WAYLAND_DISPLAY", "XDG_RUNTIME_DIR", "EGL_PLATFORM")}}
    rep["libraries"] = try_dlopen(["libEGL.so.1", "libEGL.so", "libEGL_mesa.so.0", "libgbm.so.1", "libOSMesa.so.8", "libOSMesa.so",
                                   "libGL.so.1", "libGLX.so.0", "libGLX_mesa.so.0", "libX11.so.6", "libglfw.so.3"])
    rep["find_library"] = {n: ctypes.util.find_library(n) for n in ("EGL", "gbm", "OSMesa", "GL", "X11", "glfw")}
    rep["dri_nodes"] = {p: {"readable": os.access(p, os.R_OK), "writable": os.access(p, os.W_OK)} for p in sorted(glob.glob("/dev/dri/*"))}
    rep["dri_drivers"] = sorted(os.path.basename(p) for p in glob.glob("/usr/lib/x86_64-linux-gnu/dri/*_dri.so"))
    rep["x_sockets"] = sorted(glob.glob("/tmp/.X11-unix/*"))
    rep["x_servers_on_path"] = [p for p in ("Xvfb", "Xorg", "Xwayland", "weston") if any(
        os.access(os.path.join(d, p), os.X_OK) for d in os.environ.get("PATH", "").split(":"))]
    # GLX needs a display connection: try the default one
    try:
        x11 = C.CDLL("libX11.so.6")
        x11.XOpenDisplay.restype = C.c_void_p
        x11.XOpenDisplay.argtypes = [C.c_char_p]
        dpy = x11.XOpenDisplay(None)
        rep["XOpenDisplay(NULL)"] = "connected" if dpy else "no display"
        if not dpy:
            dpy0 = x11.XOpenDisplay(b":0")
            rep["XOpenDisplay(:0)"] = "connected" if dpy0 else "no display"
    except OSError as e:
        rep["XOpenDisplay(NULL)"] = f"libX11 absent: {e}"
    # what the HIP runtime itself says without a current context
    try:
        hip = C.CDLL("libamdhip64.so")
        count = C.c_uint(0)
        devs = (C.c_int * 8)()
        rc = hip.hipGLGetDevices(C.byref(count), devs, 8, 1)  # hipGLDeviceListAll = 1
        hip.hipGetErrorString.restype = C.c_char_p
        rep["hipGLGetDevices"] = {"rc": rc, "error": hip.hipGetErrorString(rc).decode(), "count": count.value}
    except (OSError, AttributeError) as e:
        rep["hipGLGetDevices"] = f"unavailable: {e}"
    egl = any(v == "loaded" for k, v in rep["libraries"].items() if "EGL" in k)
    osmesa = any(v == "loaded" for k, v in rep["libraries"].items() if "OSMesa" in k)
    glx = rep.get("XOpenDisplay(NULL)") == "connected" or rep.get("XOpenDisplay(:0)") == "connected"
    rep["verdict"] = {
        "egl_context_possible": egl, "glx_context_possible": glx, "osmesa_possible": osmesa,
        "gl_context_possible": bool(egl or glx),
        "note": "HIP-GL interop (hipGraphicsGLRegisterBuffer) needs a CURRENT Mesa GLX or EGL context: the runtime resolves "
                "MesaGLInterop{GLX,EGL}ExportObject from libGL/libEGL.  OSMesa (software) has no interop export.",
    }
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
