#!/usr/bin/env python3
"""Measured peaks of this box: streaming-read HBM bandwidth and sustained dense bf16 MFMA rate."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cortex_amd import _lib
L = _lib.load()
out = {}
for gb in (1, 3, 8):
    v = C.c_double(0)
    assert L.cx_probe_read_bw(0, gb << 30, 5, C.byref(v)) == 0, L.cx_last_error()
    out[f"hbm_read_GBs_{gb}GiB"] = v.value
for ms in (5.0, 50.0, 500.0):
    v = C.c_double(0)
    assert L.cx_probe_mfma_tflops(0, ms, C.byref(v)) == 0, L.cx_last_error()
    out[f"mfma_bf16_TFLOPs_{int(ms)}ms"] = v.value
for dma in (0, 1, 2):
    w = C.c_double(0)
    assert L.cx_probe_mfma_lds_tflops(0, 50.0, dma, C.byref(w)) == 0, L.cx_last_error()
    out["mfma_bf16_lds_fed_TFLOPs" + ("", "_with_lds_dma", "_with_register_staged_loads")[dma]] = w.value
print(json.dumps(out))
