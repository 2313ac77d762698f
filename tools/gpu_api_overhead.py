"""Developer tool: where the time of host-array calls goes (per C-ABI symbol vs Python), cfg1-like and a larger case."""
import sys, time, collections
sys.path.insert(0, ".")
import numpy as np, scipy.sparse as sps
import sparse_dot_amd as sda
from sparse_dot_amd._mi_interface import MI

acc = collections.defaultdict(float); cnt = collections.Counter()
def _wrap(name, f):
    def g(*a):
        t0 = time.perf_counter(); r = f(*a); acc[name] += time.perf_counter() - t0; cnt[name] += 1; return r
    return g
for _n in list(MI.fn):
    MI.fn[_n] = _wrap(_n, MI.fn[_n])

def run(label, a, b, reps=20, **kw):
    for _ in range(3): sda.dot_product_mkl(a, b, **kw)
    acc.clear(); cnt.clear()
    t0 = time.perf_counter()
    for _ in range(reps): c = sda.dot_product_mkl(a, b, **kw)
    t = (time.perf_counter() - t0) / reps
    print("== %s: %.3f ms per call" % (label, t * 1e3))
    tot = 0
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
        print("   %-28s %7.3f ms (%d calls)" % (k, v / reps * 1e3, cnt[k] // reps)); tot += v / reps
    print("   %-28s %7.3f ms" % ("python / numpy / scipy", (t - tot) * 1e3))

rng = np.random.default_rng(0)
a = sps.random(10000, 10000, density=0.001, format="csr", random_state=1, dtype=np.float64)
b = rng.standard_normal((10000, 64))
n2 = 1 << 18   # built directly: scipy.sparse.random samples without replacement from m*n and needs far too much memory here
ind = np.sort(rng.integers(0, n2, (n2, 32)), axis=1).astype(np.int32).ravel()
a2 = sps.csr_matrix((rng.standard_normal(ind.size).astype(np.float32), ind, np.arange(0, ind.size + 1, 32)), shape=(n2, n2))
b2 = rng.standard_normal((1 << 18, 128)).astype(np.float32)
import os
for staged in ((1, 0) if not os.environ.get("ONLY_STAGED") else (1,)):
  sda.mi_set_option("staged_copies", staged)
  print("#### staged_copies =", staged)
  run("cfg1 SpMM 10k x 10k (1e5 nnz) fp64 x 64", a, b)
  run("SpMM 2^18 (8.4M nnz) fp32 x 128", a2, b2, reps=5)
  out2 = np.empty((n2, 128), dtype=np.float32)
  run("SpMM 2^18 (8.4M nnz) fp32 x 128, preallocated out (out_scalar=0)", a2, b2, reps=5, out=out2, out_scalar=0.0)
  run("SpGEMM 10k x 10k (1e5 nnz) fp64 squared", a, a)
  run("SpMV 2^18 fp32", a2, b2[:, 0].copy(), reps=10)
