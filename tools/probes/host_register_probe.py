"""Probe (round 4): what does hipHostRegister of the CALLER's pageable arrays cost, and what does it buy?
Host-array product of DESIGN section 8.4: B and C are 2^18 x 128 fp32 = 134 MB each.  Compared: the library's staged copy
(worker threads memcpy 4 MiB chunks into pinned slots and DMA them) vs  register -> one hipMemcpyAsync -> unregister,
and the same with the registration kept (what an LRU of registrations would give a caller that re-uses its arrays)."""
import ctypes as ct, time, os, sys
import numpy as np, torch
hip = ct.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [ct.c_void_p, ct.c_size_t, ct.c_uint]
hip.hipHostUnregister.argtypes = [ct.c_void_p]
hip.hipMemcpyAsync.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_size_t, ct.c_int, ct.c_void_p]
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
for mb in (8, 134, 537):
    n = mb << 20
    src = np.ones(n, dtype=np.uint8)
    dst = np.empty(n, dtype=np.uint8)
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    def sync(): torch.cuda.synchronize()
    def best(f, reps=4):
        f(); sync(); b = 1e9
        for _ in range(reps):
            t0 = time.perf_counter(); f(); sync(); b = min(b, time.perf_counter() - t0)
        return b * 1e3
    def pageable_up(): assert hip.hipMemcpyAsync(d.data_ptr(), src.ctypes.data, n, 1, None) == 0
    def pageable_down(): assert hip.hipMemcpyAsync(dst.ctypes.data, d.data_ptr(), n, 2, None) == 0
    def reg(a): assert hip.hipHostRegister(a.ctypes.data, n, 0) == 0
    def unreg(a): assert hip.hipHostUnregister(a.ctypes.data) == 0
    def reg_only(): reg(src); unreg(src)
    def reg_copy_unreg(): reg(src); pageable_up(); sync(); unreg(src)
    t_reg = best(reg_only)
    t_pg_up, t_pg_dn = best(pageable_up), best(pageable_down)
    t_rcu = best(reg_copy_unreg)
    reg(src); reg(dst)
    t_kept_up, t_kept_dn = best(pageable_up), best(pageable_down)
    unreg(src); unreg(dst)
    print("%4d MB: hipMemcpy pageable up %.2f / down %.2f ms | register+unregister alone %.2f ms | register, copy up, unregister %.2f ms | "
          "registration kept: up %.2f ms (%.1f GB/s), down %.2f ms (%.1f GB/s)"
          % (mb, t_pg_up, t_pg_dn, t_reg, t_rcu, t_kept_up, n / t_kept_up / 1e6, t_kept_dn, n / t_kept_dn / 1e6), flush=True)
print("cpus:", os.cpu_count())
