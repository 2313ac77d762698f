#!/bin/bash
# Developer tool (survey-container only): point the UNMODIFIED reference at the MKL-named alias library through $MKL_RT
# and run the reference's OWN test-suite on it.  No GPU here, so the alias library is linked against the host-thread
# emulation build of the kernels (tools/hip_emu/build_emu.sh): what is exercised is the ABI (names, argument
# conventions, status codes, ownership) and the kernels' logic, not the hardware.  Everything is built under /tmp.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$HERE/../.."
bash "$HERE/build_emu.sh" /tmp/mi_alias/libmi_sparse.so > /dev/null 2>&1 || { mkdir -p /tmp/mi_alias; bash "$HERE/build_emu.sh" /tmp/mi_alias/libmi_sparse.so; }
g++ -O2 -std=c++17 -fPIC -shared -o /tmp/mi_alias/libmi_mkl_rt.so "$ROOT/sparse_dot_amd/csrc/mkl_alias.cpp" -L/tmp/mi_alias -lmi_sparse -Wl,-rpath,/tmp/mi_alias
cd /tmp
export PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference MKL_RT=/tmp/mi_alias/libmi_mkl_rt.so
python3 - <<'PY'
import sparse_dot_mkl as s, numpy as np, scipy.sparse as sps
print("imported the reference on:", s.get_version_string())
print("interface integer:", s.mkl_interface_integer_dtype())
a = sps.random(50, 40, density=0.2, format="csr", random_state=0); b = np.random.default_rng(0).random((40, 7))
print("spmm  max err", np.abs(s.dot_product_mkl(a, b) - a @ b).max())
c = s.dot_product_mkl(a, a.T.tocsr()); print("spgemm max err", np.abs(c.toarray() - (a @ a.T).toarray()).max())
g = s.gram_matrix_mkl(a, dense=True); print("gram  max err", np.abs(np.triu(g) - np.triu((a.T @ a).toarray())).max())
PY
# the reference's own tests (pytest collects them from the read-only tree; cache disabled)
timeout ${ALIAS_TEST_TIMEOUT:-3000} python3 -m pytest -p no:cacheprovider -q ${ALIAS_TESTS:-/root/reference/sparse_dot_mkl/tests/test_sparse_dense.py /root/reference/sparse_dot_mkl/tests/test_sparse_sparse.py /root/reference/sparse_dot_mkl/tests/test_gram_matrix.py /root/reference/sparse_dot_mkl/tests/test_dense_dense.py /root/reference/sparse_dot_mkl/tests/test_sparse_vector.py} 2>&1 | tail -15
