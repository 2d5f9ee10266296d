import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pmf_amd import _lib as L
lib = L.lib()
lib.pmf_bn_bwd_reduce.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
lib.pmf_bn_bwd_reduce.restype = C.c_int
def bench(C_, npix, tag):
    gy = torch.randn(npix, C_, device="cuda"); a = torch.randn(npix, C_, device="cuda")
    mean = torch.zeros(C_, device="cuda"); red = torch.zeros(4096*2*C_, dtype=torch.float64, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3): lib.pmf_bn_bwd_reduce(gy.data_ptr(), C_, a.data_ptr(), C_, npix, C_, mean.data_ptr(), red.data_ptr(), st)
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): lib.pmf_bn_bwd_reduce(gy.data_ptr(), C_, a.data_ptr(), C_, npix, C_, mean.data_ptr(), red.data_ptr(), st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1)/20*1e3
    print("%s C=%d npix=%d  %.1f us  %.2f TB/s" % (tag, C_, npix, us, 2*npix*C_*4/us/1e6))
    # torch reference streaming speed
    e0.record()
    for _ in range(20): (gy*a).sum(0)
    e1.record(); torch.cuda.synchronize(); print("   torch (gy*a).sum(0): %.1f us" % (e0.elapsed_time(e1)/20*1e3))
bench(32, 262144, os.environ.get("PMF_COL_GX","1024"))
bench(64, 262144, os.environ.get("PMF_COL_GX","1024"))
bench(256, 4096, os.environ.get("PMF_COL_GX","1024"))
