"""Host-side driver of the Powell state machine (nlml_hpe_amd/csrc/powell.h) -- no GPU involved.

``minimize_powell(fun, x0)`` steps the same machine the device kernel runs, calling a Python
objective at each suspend point.  It exists so the restated control flow can be compared with
scipy.optimize.minimize(method='Powell') on the CPU (tests/test_powell_sm.py); the product path for
TD inference is the device kernel (ops.tucker_powell).
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace

import numpy as np

from . import _lib

STATUS = {0: "running", 1: "converged", 2: "maxfev", 3: "maxiter", 4: "nan"}


def minimize_powell(fun, x0, xtol: float = 1e-4, ftol: float = 1e-4, record: list | None = None):
    L = _lib.lib()
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    if x0.shape != (8,):
        raise ValueError("the state machine is built for the 8 parameters of TD_Tester.Test")
    state = C.create_string_buffer(L.nlml_powell_state_bytes())
    _lib.check(L.nlml_powell_init(state, x0.ctypes.data_as(C.c_void_p), xtol, ftol), "nlml_powell_init")
    xe = np.empty(8, dtype=np.float64)
    f = 0.0
    while L.nlml_powell_step(state, C.c_double(f), xe.ctypes.data_as(C.c_void_p)) == 1:
        if record is not None:
            record.append(xe.copy())
        f = float(fun(xe.copy()))
    x = np.empty(8)
    fval, nfev, nit, status = C.c_double(), C.c_int(), C.c_int(), C.c_int()
    _lib.check(L.nlml_powell_result(state, x.ctypes.data_as(C.c_void_p), C.byref(fval), C.byref(nfev), C.byref(nit),
                                    C.byref(status)), "nlml_powell_result")
    return SimpleNamespace(x=x, fun=fval.value, nfev=nfev.value, nit=nit.value, status=status.value,
                           message=STATUS.get(status.value, "?"))
