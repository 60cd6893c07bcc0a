"""Python side of the host-emulation harness (tests/emu/emu.cpp): test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

from mi355fft import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("MI355_EMU_LIB")
        if not path:
            subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "emu"), "libmi355emu.so"])
            path = os.path.join(_HERE, "emu", "libmi355emu.so")
        L = ctypes.CDLL(path)
        L.emu_run_plan.restype = ctypes.c_int
        L.emu_run_plan.argtypes = [ctypes.POINTER(_abi.PlanDesc), ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64,
                                   ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_size_t,
                                   ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int)]
        L.emu_check_registry.restype = ctypes.c_int
        L.emu_check_registry.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        L.emu_fill_random.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint64]
        L.emu_diff_sumsq.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_uint64, ctypes.POINTER(ctypes.c_double)]
        L.emu_plan_only.restype = ctypes.c_int
        L.emu_plan_only.argtypes = [ctypes.POINTER(_abi.PlanDesc), ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t,
                                    ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint64)]
        _LIB = L
    return _LIB


class EmuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


def plan_only(desc, compute_units=256):
    """The product planner (MI355FFT_* environment as the library reads it) without running anything: (route, launches, workspace bytes)."""
    err = ctypes.create_string_buffer(1024)
    route = ctypes.create_string_buffer(2048)
    launches = ctypes.c_int(0)
    work = ctypes.c_uint64(0)
    rc = lib().emu_plan_only(ctypes.byref(desc), compute_units, err, 1024, route, 2048, ctypes.byref(launches), ctypes.byref(work))
    if rc != 0:
        raise EmuError(rc, err.value.decode())
    return route.value.decode(), launches.value, work.value


def run_plan(desc, x, out_floats, kernel=None, force_generic=False, chunk_bytes=0, out_init=None):
    """Runs the planned transform on host arrays; returns (out, route, launches)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    if desc.in_place:
        buf = x.copy()
        out = buf
        outp, outb = None, 0
    else:
        buf = x
        out = np.zeros(out_floats, dtype=np.float32) if out_init is None else np.array(out_init, dtype=np.float32, copy=True)
        outp, outb = out.ctypes.data, out.nbytes
    kp, kb = (None, 0)
    if kernel is not None:
        kernel = np.ascontiguousarray(kernel, dtype=np.float32)
        kp, kb = kernel.ctypes.data, kernel.nbytes
    err = ctypes.create_string_buffer(1024)
    route = ctypes.create_string_buffer(1024)
    launches = ctypes.c_int(0)
    rc = lib().emu_run_plan(ctypes.byref(desc), buf.ctypes.data, buf.nbytes, outp, outb, kp, kb, 1 if force_generic else 0, chunk_bytes,
                            err, 1024, route, 1024, ctypes.byref(launches))
    if rc != 0:
        raise EmuError(rc, err.value.decode())
    return out, route.value.decode(), launches.value
