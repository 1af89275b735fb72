#!/usr/bin/env python3
"""What RCCL allows on a one-GPU box: (1) a one-rank communicator with a send/recv to itself inside a group, (2) a
two-rank communicator with both ranks on device 0 (ncclCommInitAll with a duplicate device: expected to be refused)."""
import ctypes as C
import json

hip = C.CDLL("libamdhip64.so")
rccl = C.CDLL("librccl.so.1")
rccl.ncclGetErrorString.restype = C.c_char_p
rep = {}


def ok(rc):
    return "ok" if rc == 0 else f"error {rc}: {rccl.ncclGetErrorString(rc).decode()}"


ver = C.c_int(0)
rccl.ncclGetVersion(C.byref(ver))
rep["version"] = ver.value
comm = C.c_void_p()
devs = (C.c_int * 1)(0)
rep["commInitAll(1 rank)"] = ok(rccl.ncclCommInitAll(C.byref(comm), 1, devs))
if comm:
    a, b = C.c_void_p(), C.c_void_p()
    hip.hipMalloc(C.byref(a), 1 << 20)
    hip.hipMalloc(C.byref(b), 1 << 20)
    hip.hipMemset(a, 0x5A, 1 << 20)
    hip.hipMemset(b, 0, 1 << 20)
    rccl.ncclGroupStart()
    r1 = rccl.ncclSend(a, C.c_size_t(1 << 20), 0, 0, comm, None)  # ncclInt8 = 0 / ncclChar
    r2 = rccl.ncclRecv(b, C.c_size_t(1 << 20), 0, 0, comm, None)
    r3 = rccl.ncclGroupEnd()
    hip.hipDeviceSynchronize()
    host = (C.c_ubyte * 16)()
    hip.hipMemcpy(host, b, 16, 2)
    rep["self send/recv"] = {"send": ok(r1), "recv": ok(r2), "group_end": ok(r3), "first_bytes": list(host)[:4]}
    rccl.ncclCommDestroy(comm)
comms = (C.c_void_p * 2)()
devs2 = (C.c_int * 2)(0, 0)
rep["commInitAll(2 ranks on device 0)"] = ok(rccl.ncclCommInitAll(comms, 2, devs2))
print(json.dumps(rep, indent=1))
