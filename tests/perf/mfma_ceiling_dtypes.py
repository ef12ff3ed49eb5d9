"""Bare-MFMA ceiling per operand type (scldm_mfma_sustained_tflops): the same register-only loop and values on the bf16 and the fp16 instruction."""
import ctypes as C, sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from scldm_amd import _lib
L = _lib.lib()
for rep in range(3):
    for fill, name in ((2, "bf16 normal"), (6, "fp16 normal"), (1, "bf16 uniform"), (5, "fp16 uniform"), (0, "bf16 zero"), (4, "fp16 zero")):
        v = C.c_double()
        _lib.check(L.scldm_mfma_sustained_tflops(fill, 12000, C.byref(v)), "x")
        print(f"{name:14s} {v.value:8.1f} TFLOP/s")
