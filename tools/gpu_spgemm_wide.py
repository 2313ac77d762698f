"""Developer tool (round 5): SpGEMM with a B wider than the big-row path's LDS bitmap (> ~1.1 M columns): product by column
panels (csrc/spgemm.hip, spgemm_panels) against the same product on a narrow B and against the global-memory hash.

    python tools/gpu_spgemm_wide.py [scale=18]

A, B0 = R-MAT 2^scale, 16 edges/row, fp64 (the configs[2] generator).  Cases: B0 as it is (the fast path);  B0 with its
columns spread over 16 x as many by a STRIDE (column 16 c: same pattern of collisions, same nnz(C) -- the ratio to the first
case is the panels' overhead);  spread at RANDOM (16 c + r: fewer collisions, a longer result);  each wide case with panels
and, at small scales, with option spgemm_col_panels = 0 (global-memory hash).  ms per call (3 calls) and per 1e9 products."""
import sys, time, ctypes as ct, json
sys.path.insert(0, "/root/repo")
import torch, bench
import sparse_dot_amd as sda
from sparse_dot_amd._mi_interface import MI, sparse_matrix_t, _check_return_value
dev = torch.device("cuda", 0)
sda.mi_set_stream(torch.cuda.current_stream().cuda_stream)
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 18
with_global = (len(sys.argv) > 2 and sys.argv[2] == "global")
if len(sys.argv) > 2 and sys.argv[2] == "trace":
    sda.mi_set_option("trace_phases", 1)
a = bench.rmat_csr(torch, scale, 16, 21, dev); b = bench.rmat_csr(torch, scale, 16, 23, dev)
n = a[3]; av, bv = a[2].double(), b[2].double()
ipa, ipb = a[0].to(torch.int64), b[0].to(torch.int64)
blen = (ipb[1:] - ipb[:-1])
products = int(blen[a[1].long()].sum())
g = torch.Generator(device=dev); g.manual_seed(3)
col_stride = (b[1].to(torch.int64) * 16).to(torch.int32)
col_rand = (b[1].to(torch.int64) * 16 + torch.randint(0, 16, (b[1].numel(),), device=dev, generator=g)).to(torch.int32)  # stays sorted, stays distinct
def mk(rows, cols, ipt, idx, v):
    h = sparse_matrix_t(); _check_return_value(MI.call("mi_sparse_d_create_csr", ct.byref(h), 0, rows, cols, ipt.data_ptr(), ipt.data_ptr() + 4, idx.data_ptr(), v.data_ptr()), "create"); return h
ha = mk(n, n, a[0], a[1], av)
cases = [("narrow B (2^%d columns)" % scale, mk(n, n, b[0], b[1], bv), 1),
         ("wide B, stride 16 (2^%d columns), panels" % (scale + 4), mk(n, n * 16, b[0], col_stride, bv), 1),
         ("wide B, random spread (2^%d columns), panels" % (scale + 4), mk(n, n * 16, b[0], col_rand, bv), 1)]
only = [a.split("=")[1] for a in sys.argv if a.startswith("only=")]
if only:
    cases = [c for c in cases if only[0] in c[0]]
if with_global:
    cases += [("wide B, stride 16, global hash", mk(n, n * 16, b[0], col_stride, bv), 0)]
for name, hb, opt in cases:
    sda.mi_set_option("spgemm_col_panels", opt)
    sda.mi_get_counter("reset")
    ts = []; nnzc = None
    for rep in range(3):
        hc = sparse_matrix_t(); torch.cuda.synchronize(); t0 = time.perf_counter()
        _check_return_value(MI.call("mi_sparse_spmm", 10, ha, hb, ct.byref(hc)), "spmm"); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        MI.call("mi_sparse_destroy", hc)
    print(json.dumps({"case": name, "scale": scale, "products": products, "ms": [round(t, 2) for t in ts],
                      "ms_per_1e9_products": round(min(ts) / products * 1e9, 2), "panels_per_call": sda.mi_get_counter("spgemm_panels") / 3}), flush=True)
sda.mi_set_option("spgemm_col_panels", 1)
