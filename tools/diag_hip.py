import ctypes, sys, os
order = sys.argv[1] if len(sys.argv) > 1 else "torch_first"
def maps():
    return sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'hsa-runtime' in l))
if order == "torch_first":
    import torch
    print("torch avail", torch.cuda.is_available(), torch.cuda.device_count())
L = ctypes.CDLL(os.path.join(os.path.dirname(__file__), '..', 'loudgain_amd', 'csrc', 'libloudscan_hip.so'))
print(maps())
hip = ctypes.CDLL(None)
n = ctypes.c_int(-1)
try:
    f = hip.hipGetDeviceCount
    rc = f(ctypes.byref(n)); print("hipGetDeviceCount rc", rc, "n", n.value)
except Exception as e:
    print("no global sym", e)
L.lgd_create.restype = ctypes.c_void_p
L.lgd_last_error.restype = ctypes.c_char_p
c = L.lgd_create(0); print("ctx", c, L.lgd_last_error())
if order != "torch_first":
    import torch
    print(maps())
    print("torch avail", torch.cuda.is_available())
    x = torch.zeros(4, device="cuda"); print(x)
