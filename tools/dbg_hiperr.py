import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
x = torch.zeros(4).cuda()
maps = open("/proc/self/maps").read()
libs = sorted({l.split()[-1] for l in maps.splitlines() if "amdhip64" in l or "hsa-runtime" in l})
print("hip libs:", libs)
hip = ctypes.CDLL([l for l in libs if "amdhip64" in l][0])
hip.hipGetErrorString.restype = ctypes.c_char_p
e = hip.hipGetLastError()
print("after torch .cuda():", e, hip.hipGetErrorString(e))
e = hip.hipGetLastError()
print("again:", e)
from ibloc_amd import _lib
libs2 = sorted({l.split()[-1] for l in open("/proc/self/maps").read().splitlines() if "amdhip64" in l})
print("hip libs after ibloc:", libs2)
import numpy as np
from ibloc_amd import match
a = torch.randn(8, 64, device="cuda")
try:
    print(match.normalize_rows(a).shape)
except Exception as ex:
    print("ERR", ex)
e = hip.hipGetLastError()
print("after ibl:", e)
y = torch.zeros(4, device="cuda") + 1
torch.cuda.synchronize()
try:
    print(match.normalize_rows(a).shape)
except Exception as ex:
    print("ERR2", ex)
