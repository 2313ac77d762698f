/*
 * sparse_oracle.c -- CPU oracle for the dot_product_mkl / gram_matrix_mkl hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load this library, and only as the checker.  The product
 * (sparse_dot_amd + libmi_sparse.so) never links, imports or falls back to anything in oracle/.
 *
 * What it restates: the arithmetic the reference (flatironinstitute/sparse_dot,
 * sparse_dot_mkl 0.9.6) delegates to Intel MKL -- mkl_sparse_?_mm, mkl_sparse_spmm (+ order,
 * export), mkl_sparse_?_spmmd, mkl_sparse_syrk, mkl_sparse_?_syrkd, mkl_sparse_convert_csr,
 * cblas_?gemm, cblas_?syrk.  MKL is closed source and absent from /root/reference, so the
 * routines follow MKL's published semantics; see oracle_kernels.inc for the per-function
 * reference call sites.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every routine here against
 * tests/golden/ (npz files), which hold inputs and outputs captured from the unmodified reference
 * running on oneMKL 2021.4 in the build container (generator: oracle/make_golden.py), and
 * against scipy.sparse for the same inputs.
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared -fPIC)  ->  oracle/liboracle.so
 */
#include <complex.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_ID(x) (x)

#define T float
#define FN(x) orc_s_##x
#define CONJ(x) ORC_ID(x)
#include "oracle_kernels.inc"
#undef T
#undef FN
#undef CONJ

#define T double
#define FN(x) orc_d_##x
#define CONJ(x) ORC_ID(x)
#include "oracle_kernels.inc"
#undef T
#undef FN
#undef CONJ

#define T float _Complex
#define FN(x) orc_c_##x
#define CONJ(x) conjf(x)
#include "oracle_kernels.inc"
#undef T
#undef FN
#undef CONJ

#define T double _Complex
#define FN(x) orc_z_##x
#define CONJ(x) conj(x)
#include "oracle_kernels.inc"
#undef T
#undef FN
#undef CONJ

int orc_version(void) { return 1; }
