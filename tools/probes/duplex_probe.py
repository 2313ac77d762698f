"""Probe: can an upload and a download run at the same time on this box (pinned buffers, two streams), and what does the
host-side staging (pageable -> pinned memcpy by worker threads) add?  Sizes of the host-array product of DESIGN section 8.4."""
import time, threading
import numpy as np, torch
n = 134 << 20
dev = torch.device("cuda", 0)
hu = torch.empty(n, dtype=torch.uint8).pin_memory(); hd = torch.empty(n, dtype=torch.uint8).pin_memory()
du = torch.empty(n, dtype=torch.uint8, device=dev); dd = torch.empty(n, dtype=torch.uint8, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def t(f, reps=5):
    f(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
def up():
    with torch.cuda.stream(s1): du.copy_(hu, non_blocking=True)
def down():
    with torch.cuda.stream(s2): hd.copy_(dd, non_blocking=True)
def both(): up(); down()
a, b, c = t(up), t(down), t(both)
print("pinned 134 MB: up %.2f ms (%.1f GB/s), down %.2f ms (%.1f GB/s), both at once %.2f ms (sum %.2f)" % (a, n / a / 1e6, b, n / b / 1e6, c, a + b))
# chunked: 32 x 4 MiB each way, interleaved on two streams
def chunks():
    k = 4 << 20
    for i in range(0, n, k):
        with torch.cuda.stream(s1): du[i:i + k].copy_(hu[i:i + k], non_blocking=True)
        with torch.cuda.stream(s2): hd[i:i + k].copy_(dd[i:i + k], non_blocking=True)
print("pinned, 4 MiB chunks interleaved on two streams: %.2f ms" % t(chunks))
# host memcpy bandwidth, 1 and 8 threads (numpy releases the GIL in copyto)
src = np.ones(n, dtype=np.uint8); dst = np.empty(n, dtype=np.uint8)
def cp(nt):
    def f():
        th = [threading.Thread(target=lambda i=i: np.copyto(dst[i * n // nt:(i + 1) * n // nt], src[i * n // nt:(i + 1) * n // nt])) for i in range(nt)]
        [x.start() for x in th]; [x.join() for x in th]
    return f
for nt in (1, 4, 8, 16):
    x = t(cp(nt)); print("host memcpy 134 MB, %2d threads: %.2f ms (%.1f GB/s)" % (nt, x, n / x / 1e6))
import os; print("cpus:", os.cpu_count())
