"""Developer tool (round 5): the dense x dense fallback (mi_cblas_?gemm / mi_cblas_?syrk, the only
MFMA consumer) on device-resident operands -- TFLOP/s against the dense MFMA peak (fp32 157.3, fp64 78.6 TFLOP/s), with a
torch.matmul (hipBLASLt / rocBLAS) reading beside it as the practical ceiling on this box.

    python tools/gpu_gemm.py [sizes ...]      e.g. 1024 4096 8192 (square), or 4096x4096x512 as m x n x k"""
import sys, time, ctypes as ct, json
sys.path.insert(0, "/root/repo")
import torch
import sparse_dot_amd as sda
from sparse_dot_amd._mi_interface import MI, _check_return_value
dev = torch.device("cuda", 0)
sda.mi_set_stream(torch.cuda.current_stream().cuda_stream)
PEAK = {"s": 157.3, "d": 78.6}
shapes = []
for a in (sys.argv[1:] or ["1024", "4096", "8192"]):
    p = [int(x) for x in a.split("x")]
    shapes.append((p[0], p[0], p[0]) if len(p) == 1 else tuple(p))
for letter, tdt in (("s", torch.float32), ("d", torch.float64)):
    for (m, n, k) in shapes:
        A = torch.rand((m, k), device=dev, dtype=tdt); B = torch.rand((k, n), device=dev, dtype=tdt); C = torch.empty((m, n), device=dev, dtype=tdt)
        At = A.t().contiguous()
        for ta, tb, Aop, lda, Bop, ldb in ((111, 111, A, k, B, n), (111, 112, A, k, B.t().contiguous(), k),  # NN, NT (B^T stored)
                                           (112, 111, At, m, B, n), (112, 112, At, m, B.t().contiguous(), k)):  # TN (A^T stored), TT
            def step():
                r = MI.call("mi_cblas_%sgemm" % letter, 101, ta, tb, m, n, k, 1.0, Aop.data_ptr(), lda, Bop.data_ptr(), ldb, 0.0, C.data_ptr(), n)
                if r:
                    _check_return_value(r, "gemm")
            step(); torch.cuda.synchronize()
            err = float(((C - A @ B).abs() / (A @ B).abs().clamp(min=1e-30)).max())
            reps = 5 if m * n * k >= 2 ** 33 else 20
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                step()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            e0.record()
            for _ in range(reps):
                torch.matmul(A, B, out=C)
            e1.record(); torch.cuda.synchronize()
            ms_t = e0.elapsed_time(e1) / reps
            tf = 2.0 * m * n * k / ms / 1e9
            print(json.dumps({"gemm": letter, "m": m, "n": n, "k": k, "transa": ta == 112, "transb": tb == 112, "ms": round(ms, 4), "TFLOPs": round(tf, 2),
                              "frac_of_mfma_peak": round(tf / PEAK[letter], 4), "torch_matmul_TFLOPs": round(2.0 * m * n * k / ms_t / 1e9, 2),
                              "max_rel_err_vs_torch": err}), flush=True)
