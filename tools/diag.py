import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else "torch_first"
if mode == "torch_first":
    import torch
    print("torch cuda", torch.cuda.is_available(), torch.version.hip)
    x = torch.ones(4, device="cuda"); print(x.sum().item())
from rabe_amd import Engine
try:
    e = Engine(0)
    print(mode, "engine ok", e.device_info())
    ms, ops = e.calibrate(2, 100)
    print("kernel launch ok", ms)
except Exception as ex:
    print(mode, "FAILED:", ex)
with open("/proc/self/maps") as f:
    libs = sorted({l.split()[-1] for l in f if "libamdhip64" in l or "libhsa-runtime" in l})
print(libs)
if mode != "torch_first":
    import torch
    print("torch after: cuda", torch.cuda.is_available())
    x = torch.ones(4, device="cuda"); print(x.sum().item())
