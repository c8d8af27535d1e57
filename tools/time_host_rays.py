"""Closest-hit throughput when the caller hands over host buffers (transfers included) against device-resident rays:
python3 tools/time_host_rays.py [n_rays]   -- pageable numpy arrays, page-locked arrays, device tensors."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from pathtracer_amd import abi, capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
verts, idx = synth.soup(1_000_000)
scene = capi.Scene.from_mesh(verts, idx, device=0)
dev = torch.device("cuda", 0)
lib = capi.lib()
rays_h = synth.rays(n)
flat = rays_h.view(np.float32).reshape(-1, 8)

rays_d = torch.from_numpy(flat).to(dev)
ref = torch.empty((n, 2), dtype=torch.int32, device=dev)
scene.trace_closest_device(rays_d, ref, flags=abi.PTX_TRAVERSE_CULLED, n=n)
ref_h = ref.cpu().numpy()


def run(label, rays_ptr, hits_ptr, flags, hits_view, reps=5):
    capi.check(lib.ptx_trace_closest(scene.h, rays_ptr, n, hits_ptr, None, None, flags))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        capi.check(lib.ptx_trace_closest(scene.h, rays_ptr, n, hits_ptr, None, None, flags))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    same = bool(np.array_equal(hits_view(), ref_h))
    print(f"{label}: {dt * 1e3:.2f} ms per call, {n / dt / 1e6:.1f} Mrays/s, equal to the device-resident call: {same}", flush=True)


hits_d = torch.empty((n, 2), dtype=torch.int32, device=dev)
run("device-resident", rays_d.data_ptr(), hits_d.data_ptr(), abi.PTX_TRAVERSE_CULLED | abi.PTX_MEM_DEVICE, lambda: hits_d.cpu().numpy())

hits_p = np.empty((n, 2), dtype=np.int32)
run("pageable host arrays", flat.ctypes.data, hits_p.ctypes.data, abi.PTX_TRAVERSE_CULLED, lambda: hits_p)

rays_pin = torch.from_numpy(flat).pin_memory()
hits_pin = torch.empty((n, 2), dtype=torch.int32).pin_memory()
run("page-locked host arrays", rays_pin.data_ptr(), hits_pin.data_ptr(), abi.PTX_TRAVERSE_CULLED, lambda: hits_pin.numpy())
