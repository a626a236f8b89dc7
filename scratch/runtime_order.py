"""Which HIP / HSA runtime libraries end up in the process, and in which order torch and the engine can initialise the GPU."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def maps():
    out = set()
    for ln in open("/proc/self/maps"):
        p = ln.split()[-1]
        if any(k in p for k in ("libamdhip64", "libhsa-runtime", "libawsem_hip", "librccl", "libhiprtc")):
            out.add(p)
    return sorted(out)

def engine():
    from openawsem_amd import _capi
    api = _capi.load_hip_library()
    ctx = _capi.Context(api, 10, 1)
    return ctx

def torch_init():
    import torch
    torch.cuda.init()
    return torch.zeros(4, device="cuda").sum().item()

if __name__ == "__main__":
    if len(sys.argv) == 1:
        for order in ("torch-first", "engine-first", "import-torch-then-engine-then-cuda"):
            r = subprocess.run([sys.executable, __file__, order], capture_output=True, text=True)
            print("==", order, "rc", r.returncode); print(r.stdout[-1500:]); print(r.stderr[-600:])
        sys.exit(0)
    order = sys.argv[1]
    try:
        if order == "torch-first":
            print("torch", torch_init()); print(maps()); c = engine(); print("engine ok"); print(maps())
        elif order == "engine-first":
            c = engine(); print("engine ok"); print(maps()); print("torch", torch_init()); print(maps())
        else:
            import torch; print(maps()); c = engine(); print("engine ok"); print(maps()); print("torch", torch_init()); print(maps())
    except Exception as e:
        print("FAILED:", type(e).__name__, str(e)[:200]); print(maps())
