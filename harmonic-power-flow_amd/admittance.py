"""Per-harmonic bus admittance matrices as ONE shared CSR pattern with Hn value rows — the device layout.

Restates `build_admittance_matrices` (HG:132-171) in O(Hn*(n+L)) instead of the reference's O(Hn*n*L) pandas loops,
bug-compatibly:
  * a later line on the same bus pair overwrites an earlier one (assignment, HG:151-155);
  * diagonal = -(Python `sum` of the row) -> sequential left-to-right sum in column order (HG:159,161);
  * bus shunt 1/(j*X_sh*h) only for h != 1 (HG:158-159);
  * pi-model line shunts (G + j*h*B)/2 are added to the bus whose 0-BASED index equals the line's 1-BASED
    fromID/toID (HG:163-168), i.e. to the next bus; bus 0 never receives any.
The reference's dense `Y_all` DataFrame ((Hn*n) x n) holds the same numbers with exact zeros elsewhere;
`to_dense_frame` rebuilds it for callers that want the reference object.
"""
import numpy as np
import pandas as pd


def _seq_segment_sum(values, ptr):
    """Left-to-right sum of each segment values[ptr[i]:ptr[i+1]], vectorised across segments."""
    nseg = len(ptr) - 1
    deg = np.diff(ptr)
    acc = np.zeros(nseg, dtype=values.dtype)
    for k in range(int(deg.max()) if nseg else 0):
        sel = np.nonzero(deg > k)[0]
        acc[sel] = acc[sel] + values[ptr[sel] + k]
    return acc


class AdmittancePattern:
    """CSR pattern (off-diagonals from the lines + full diagonal) and the line that owns each off-diagonal."""

    def __init__(self, n, fromID, toID):
        f = np.asarray(fromID, dtype=np.int64) - 1
        t = np.asarray(toID, dtype=np.int64) - 1
        if len(f) and (f.min() < 0 or t.min() < 0 or f.max() >= n or t.max() >= n):
            raise ValueError("line endpoint outside 1..n")
        L = len(f)
        # writes happen in the order [f,t] (2k), [t,f] (2k+1); the last write to a cell wins
        r = np.empty(2 * L, dtype=np.int64)
        c = np.empty(2 * L, dtype=np.int64)
        r[0::2], c[0::2], r[1::2], c[1::2] = f, t, t, f
        owner = np.repeat(np.arange(L, dtype=np.int64), 2)
        key = r * n + c
        order = np.argsort(key, kind="stable")
        key_s, owner_s = key[order], owner[order]
        last = np.ones(len(key_s), dtype=bool)
        last[:-1] = key_s[1:] != key_s[:-1]
        key_u, owner_u = key_s[last], owner_s[last]
        offdiag = (key_u // n) != (key_u % n)          # a self-loop cell is overwritten by the diagonal rule
        key_u, owner_u = key_u[offdiag], owner_u[offdiag]
        dkey = np.arange(n, dtype=np.int64) * (n + 1)
        allkey = np.concatenate([key_u, dkey])
        allown = np.concatenate([owner_u, np.full(n, -1, dtype=np.int64)])
        o2 = np.argsort(allkey, kind="stable")
        allkey, allown = allkey[o2], allown[o2]
        rows = allkey // n
        self.n = n
        self.col = (allkey % n).astype(np.int32)
        self.rowptr = np.zeros(n + 1, dtype=np.int32)
        np.cumsum(np.bincount(rows, minlength=n), out=self.rowptr[1:])
        self.line = allown                              # -1 on the diagonal
        self.diag_pos = np.nonzero(allown < 0)[0]
        self.nnz = len(self.col)


def build_admittance_csr(buses, lines, harmonics):
    """-> (rowptr[n+1] int32, col[nnz] int32, Yval[Hn][nnz] complex128)."""
    n = len(buses)
    pat = AdmittancePattern(n, lines.fromID.to_numpy(), lines.toID.to_numpy())
    R = lines.R.to_numpy(dtype=float)
    X = lines.X.to_numpy(dtype=float)
    G = lines.G.to_numpy(dtype=float)
    B = lines.B.to_numpy(dtype=float)
    X_sh = buses["X_sh"].to_numpy(dtype=float)
    has_sh = X_sh != 0
    off = pat.line >= 0
    Yval = np.zeros((len(harmonics), pat.nnz), dtype=np.complex128)
    pi_pairs = None
    if np.any(G != 0) or np.any(B != 0):
        fid = lines.fromID.to_numpy(dtype=np.int64)
        tid = lines.toID.to_numpy(dtype=np.int64)
        tgt, lk = [], []
        for k in range(len(R)):                         # (bus index, line) pairs in the reference's visiting order
            for n0 in sorted({int(fid[k]), int(tid[k])}):
                if 0 <= n0 < n:
                    tgt.append(n0)
                    lk.append(k)
        tgt, lk = np.array(tgt, dtype=np.int64), np.array(lk, dtype=np.int64)
        o = np.lexsort((lk, tgt))
        tgt, lk = tgt[o], lk[o]
        ptr = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.bincount(tgt, minlength=n), out=ptr[1:])
        pi_pairs = (lk, ptr)
    for q, h in enumerate(harmonics):
        v = np.zeros(pat.nnz, dtype=np.complex128)
        if len(R):
            yl = -1 / (R + 1j * X * h)                                   # HG:151-152
            v[off] = yl[pat.line[off]]
        d = -_seq_segment_sum(v, pat.rowptr)                             # HG:159,161 (diagonal slot still 0)
        if h != 1 and np.any(has_sh):
            d[has_sh] = d[has_sh] + 1 / (1j * X_sh[has_sh] * h)          # HG:158-159
        if pi_pairs is not None:                                         # HG:163-168
            lk, ptr = pi_pairs
            sh = (G + 1j * h * B) / 2
            deg = np.diff(ptr)
            for k in range(int(deg.max())):
                sel = np.nonzero(deg > k)[0]
                d[sel] = d[sel] + sh[lk[ptr[sel] + k]]
        v[pat.diag_pos] = d
        Yval[q] = v
    return pat.rowptr, pat.col, Yval


def to_dense_frame(rowptr, col, Yval, harmonics, n):
    """The reference's `Y_all` object (HG:139-143,170): DataFrame (Hn*n) x n, MultiIndex (harmonic, bus)."""
    Hn = len(harmonics)
    dense = np.zeros((Hn * n, n), dtype=np.complex128)
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    for q in range(Hn):
        dense[q * n + rows, col] = Yval[q]
    idx = pd.MultiIndex.from_product([list(harmonics), list(range(n))], names=["harmonic", "bus"])
    return pd.DataFrame(dense, index=idx, columns=[np.arange(n)])


def from_dense_frame(Y, harmonics, n):
    """Inverse of `to_dense_frame` for callers that hand in a reference-style `Y_all`: union pattern of all
    harmonics plus the full diagonal."""
    A = np.asarray(Y.to_numpy() if hasattr(Y, "to_numpy") else Y, dtype=np.complex128).reshape(len(harmonics), n, n)
    mask = np.any(A != 0, axis=0) | np.eye(n, dtype=bool)
    rows, cols = np.nonzero(mask)
    rowptr = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(rows, minlength=n), out=rowptr[1:])
    Yval = np.ascontiguousarray(A[:, rows, cols])
    return rowptr, cols.astype(np.int32), Yval
