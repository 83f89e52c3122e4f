"""HBM copy-bandwidth calibration on this box (float4 grid-stride copy from libchap_hip.so)."""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
from chap_amd import _lib
L = _lib.lib()
L.chap_debug_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
for mb in (50, 512, 2048):
    n = mb * 1024 * 1024
    a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
    a.fill_(1)
    for blocks in (2048, 8192, 65536):
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            L.chap_debug_copy(a.data_ptr(), b.data_ptr(), n, blocks, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            L.chap_debug_copy(a.data_ptr(), b.data_ptr(), n, blocks, st)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print("copy %5d MB blocks=%6d: %8.1f us  %7.1f GB/s (read+write)" % (mb, blocks, us, 2 * n / us / 1e3))
