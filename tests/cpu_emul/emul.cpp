// TEST INFRASTRUCTURE: executes the per-row / per-entry device functions of csrc/hpf_assembly.hpp serially on the
// host so that `-m "not gpu"` tests can check index maps and rounding order against the oracle without a GPU.
// It is NOT part of libhpf.so and never on the product path.
#include "hpf_assembly.hpp"
#include <stdlib.h>
#include <string.h>
using namespace hpf;

struct HostModel {
    Model M;
};

extern "C" {

// U,E scratch: caller provides [Hn*n] cplx each.
void emul_polar(int fund, int count, const double* Vm, const double* Va, double* U, double* E) {
    for (int k = 0; k < count; ++k) {
        cplx u, e;
        if (fund) polar<true>(Vm[k], Va[k], u, e); else polar<false>(Vm[k], Va[k], u, e);
        ((cplx*)U)[k] = u;
        ((cplx*)E)[k] = e;
    }
}

static Model mk(int n, int m, int c, int Hn, int nnz, int n_dev, int coupled, const int* rowptr, const int* col,
                const int* diag, const double* Y, const int* dev, const double* YN, const double* IN) {
    Model M;
    M.n = n; M.m = m; M.c = c; M.Hn = Hn; M.nnz = nnz; M.n_dev = n_dev; M.coupled = coupled;
    M.rowptr = rowptr; M.col = col; M.diag = diag; M.Y = (const cplx*)Y; M.dev = dev;
    M.YN = (const cplx*)YN; M.IN = (const cplx*)IN;
    return M;
}

void emul_mismatch(int fund, int n, int m, int c, int Hn, int nnz, int n_dev, int coupled, const int* rowptr,
                   const int* col, const int* diag, const double* Y, const int* dev, const double* YN,
                   const double* IN, const double* U, const double* P, const double* Q, double* f) {
    Model M = mk(n, m, c, Hn, nnz, n_dev, coupled, rowptr, col, diag, Y, dev, YN, IN);
    if (fund) {
        for (int k = 1; k < n; ++k) store_mismatch(f, n - 1, c, k, mismatch_row<true>(M, (const cplx*)U, P, Q, k));
    } else {
        const int Nc = n * Hn - 1;
        for (int k = 1; k < n * Hn; ++k) store_mismatch(f, Nc, c, k, mismatch_row<false>(M, (const cplx*)U, P, Q, k));
    }
}

void emul_jacobian(int fund, int n, int m, int c, int Hn, int nnz, int n_dev, int coupled, const int* rowptr,
                   const int* col, const int* diag, const double* Y, const int* dev, const double* YN,
                   const double* IN, const double* U, const double* E, double* J) {
    Model M = mk(n, m, c, Hn, nnz, n_dev, coupled, rowptr, col, diag, Y, dev, YN, IN);
    if (fund) {
        DenseEmit em{J, 2 * n - 1 - c, n - 1, c};
        for (int i = 0; i < n; ++i)
            for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) jac_entry_fund(M, (const cplx*)U, (const cplx*)E, i, e, em);
        return;
    }
    const int Nc = n * Hn - 1;
    DenseEmit em{J, 2 * Nc - (c - 1), Nc, c};
    for (int q = 0; q < Hn; ++q)
        for (int i = 0; i < n; ++i)
            for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) jac_entry(M, (const cplx*)U, (const cplx*)E, q, i, e, em);
    if (coupled)
        for (int i = m; i < n; ++i)
            for (int q = 0; q < Hn; ++q)
                for (int p = 0; p < Hn; ++p)
                    if (p != q) jac_cross(M, (const cplx*)U, (const cplx*)E, q, p, i, em);
}

// The CSR form (hpf_jacobian_csr): indptr by a serial count + prefix sum, then the same per-row fill as k_jcsr_fill.
// indptr [N+1]; indices / data may be NULL (size query): returns nnz.
long long emul_jacobian_csr(int n, int m, int c, int Hn, int nnz, int n_dev, int coupled, const int* rowptr,
                            const int* col, const int* diag, const double* Y, const int* dev, const double* YN,
                            const double* IN, const double* U, const double* E, int* indptr, int* indices, double* data) {
    Model M = mk(n, m, c, Hn, nnz, n_dev, coupled, rowptr, col, diag, Y, dev, YN, IN);
    const int Nc = n * Hn - 1, N = 2 * Nc - (c - 1);
    long long tot = 0;
    for (int r = 0; r < N; ++r) {
        indptr[r] = (int)tot;
        const JCount k = jcsr_count_row(M, Nc, r);
        tot += k.n_theta + k.n_v;
    }
    indptr[N] = (int)tot;
    if (data)
        for (int r = 0; r < N; ++r) jcsr_fill_row(M, (const cplx*)U, (const cplx*)E, Nc, r, indptr[r], indices, data);
    return tot;
}
}
