#!/usr/bin/env python3
"""The blur's time with its arenas mapped at a chosen VIRTUAL alignment (HIP virtual memory management,
tools/probes/vmm_alloc.hip): hipMalloc / torch hands out 2 MiB-aligned addresses; does a mapping aligned
to 64 MiB or 1 GiB (larger page-table fragments) make the fast case the rule?
    python tools/probe_arena_vmm.py [cfg3] [trials]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from pano360_amd import engine, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 3
vmm = C.CDLL(os.path.join(ROOT, "build", "probes", "libvmm_alloc.so"))
vmm.vmm_alloc.restype = C.c_void_p
vmm.vmm_alloc.argtypes = [C.c_size_t, C.c_size_t]
vmm.vmm_ptr.restype = C.c_void_p
vmm.vmm_ptr.argtypes = [C.c_void_p]
vmm.vmm_size.restype = C.c_size_t
vmm.vmm_size.argtypes = [C.c_void_p]
vmm.vmm_free.argtypes = [C.c_void_p]


class Mapped:
    """A VMM block as something torch.as_tensor can alias (__cuda_array_interface__)."""

    def __init__(self, nbytes, align):
        self.block = vmm.vmm_alloc(nbytes, align)
        if not self.block:
            raise RuntimeError("vmm_alloc failed")
        self.ptr, self.size = vmm.vmm_ptr(self.block), vmm.vmm_size(self.block)
        self.__cuda_array_interface__ = {"shape": (self.size // 4,), "typestr": "<f4",
                                         "data": (self.ptr, False), "version": 2}

    def tensor(self):
        return torch.as_tensor(self, device="cuda")

    def free(self):
        vmm.vmm_free(self.block)
        self.block = None


cfg = dict(synth.CONFIGS[name])
rots, intrs = synth.make_cameras(cfg["n"], cfg["width"], cfg["height"],
                                 sweep_deg=cfg.get("sweep_deg"), step_deg=cfg.get("step_deg"))
shapes = [(cfg["height"], cfg["width"])] * cfg["n"]
pool = engine.Engine().upload_frames([synth.make_frame(i, cfg["width"], cfg["height"], "A") for i in range(4)])
frames = [pool[i % 4] for i in range(cfg["n"])]
SIZE = {"planes": 1 << 30, "blurred": 5 << 29, "scratch": 1 << 29}          # bytes
BIG, KEEP = {}, []


def placed(self, nm, floats):
    assert floats * 4 <= SIZE[nm], (nm, floats)
    self._arenas[nm] = BIG[nm]
    return BIG[nm]


def measure(label):
    eng = engine.Engine()
    plan = eng.upload_plan(engine.Plan(shapes, rots, intrs, True, 10 ** 9))
    for _ in range(4):
        eng.stitch(frames, plan, "multiband", cfg["n_levels"])
    torch.cuda.synchronize()
    eng.timing(True)
    for _ in range(30):
        eng.stitch(frames, plan, "multiband", cfg["n_levels"])
    torch.cuda.synchronize()
    t = eng.kernel_times()
    eng.timing(False)
    pick = {k.replace("_kernel", ""): round(v[0] / v[1], 4) for k, v in t.items()
            if k in ("blur_lean_kernel", "multiband_compose_kernel", "warp_windows_kernel")}
    print(f"{label}: {pick} " + " ".join(f"{k} {hex(v.data_ptr())}" for k, v in BIG.items() if k != "scratch"),
          flush=True)
    del eng


engine.Engine.arena = placed
for trial in range(trials):
    for how in ("torch", 2 << 20, 64 << 20, 1 << 30):
        holders = []
        for k, nbytes in SIZE.items():
            if how == "torch":
                BIG[k] = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda")
            else:
                holders.append(Mapped(nbytes, how))
                BIG[k] = holders[-1].tensor()
        measure(f"trial {trial}, {'torch.empty' if how == 'torch' else 'mapped at %4d MiB alignment' % (how >> 20)}")
        KEEP.append((dict(BIG), holders))          # kept alive: the next round gets other pages
