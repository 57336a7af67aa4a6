import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
print("torch avail", torch.cuda.is_available(), torch.cuda.device_count())
x = torch.zeros(4, device="cuda"); print(x.sum().item())
hip = ctypes.CDLL("libamdhip64.so.7")
c = ctypes.c_int(-1)
print("hipGetDeviceCount via soname:", hip.hipGetDeviceCount(ctypes.byref(c)), c.value)
import plonk_amd._lib as L
lib = L.load()
print("after load:", hip.hipGetDeviceCount(ctypes.byref(c)), c.value)
h = ctypes.c_void_p()
print("ctx_create rc", lib.plk_ctx_create(0, ctypes.byref(h)))
for l in open("/proc/self/maps"):
    if any(k in l for k in ("libamdhip64", "libhsa-runtime", "comgr", "rocprofiler-register")) and "r-xp" in l:
        print(l.split()[-1])
