"""CPU model of the streamed column solve (pygpa_amd/csrc/gpa_unwrap_colstream.hip), formula for formula in NumPy, against
the thing it replaces: the column half of the reference's preconditioner  z = idct(dct(r) / (lambda_k + mu_j))  along the
columns (phase_unwrap.py:95-115; SciPy's unnormalised DCT-II and its inverse, the [0, 0] divisor replaced by 1).

Three "launches" on an (N, M) array of row spectra, column j having mu_j = 2 cos(pi j / N) - 2:
    agg    per chunk of C rows:  a = sum_k lam^k r_k,  b = sum_k lam^(len-1-k) r_k        (+ c, d for column 0)
    scan   per column over the chunks: A, B, p_(-1), the p entering every chunk, z_N, the z entering every chunk from
           below, with the chunk's zero-start z-sum in closed form  e_s = -g (a_s - lam^(len+1) b_s) - q2 (1 - lam^(2 len)) P_(s-1)
    apply  per chunk: p_n = r_n + lam p_(n-1),  z_n = lam (z_(n+1) - p_n)   from the true carries;
           rho = <r, z> from z alone: -sum (z_(n+1) - z_n)^2 + mu sum z_n^2
This pins the derivation on the CPU (no GPU needed): chunk heights that do and do not divide N, the singular column 0, and
the quadratic form of rho.  The GPU tests (tests/test_gpu_unwrap_long.py) hold the kernels to the oracle."""
import numpy as np
import pytest
import scipy.fft as sfft


def column_constants(N, M):
    j = np.arange(M)
    h = 2 * np.sin(np.pi * j / (2.0 * M)) ** 2            # 1 - cos(pi j / M): square images, M == N
    lam = (1 + h) - np.sqrt(h * (2 + h))
    lam[0] = 1.0
    return h, lam


def agg(R, lam, C):
    N, M = R.shape
    S = -(-N // C)
    a = np.zeros((S, M))
    b = np.zeros((S, M))
    c0 = np.zeros(S)
    d0 = np.zeros(S)
    for s in range(S):
        blk = R[s * C:(s + 1) * C]
        ln = blk.shape[0]
        k = np.arange(ln)[:, None]
        a[s] = (lam[None, :] ** k * blk).sum(axis=0)
        b[s] = (lam[None, :] ** (ln - 1 - k) * blk).sum(axis=0)
        p0 = np.cumsum(blk[:, 0])                           # column 0: zero-start running sum of the chunk
        c0[s] = p0.sum()
        d0[s] = ((np.arange(ln) + 1) * p0).sum()
    return a, b, c0, d0


def scan(a, b, c0, d0, lam, N, C):
    S, M = a.shape
    lens = np.array([min(C, N - s * C) for s in range(S)])
    carP = np.zeros((S, M))
    carZ = np.zeros((S, M))
    col = slice(1, M)
    l = lam[col]
    ml = l[None, :] ** lens[:, None]                        # lam^len(s)
    A = np.zeros(M - 1)
    for s in range(S - 1, -1, -1):
        A = a[s, col] + ml[s] * A
    B = np.zeros(M - 1)
    for s in range(S):
        B = b[s, col] + ml[s] * B
    P = (A + l ** N * B) / (1 - l ** (2 * N))               # p_(-1)
    g, q2 = l / (1 - l * l), l * l / (1 - l * l)
    e = np.zeros((S, M - 1))
    for s in range(S):
        carP[s, col] = P
        e[s] = -g * (a[s, col] - l * ml[s] * b[s, col]) - q2 * (1 - ml[s] ** 2) * P
        P = b[s, col] + ml[s] * P
    Z = -l / (1 - l) * P                                    # z_N from p_(N-1)
    for s in range(S - 1, -1, -1):
        carZ[s, col] = Z
        Z = e[s] + ml[s] * Z
    # column 0 (lam = 1): r - mean in, mean of z removed and mean of r added afterwards
    bs = b[:, 0]
    shift0 = bs.sum() / N
    Pin = np.concatenate([[0.0], np.cumsum(bs - lens * shift0)[:-1]])
    tri = 0.5 * lens * (lens + 1)
    ps = lens * Pin + c0 - shift0 * tri                     # sum of p over each chunk
    ws = np.arange(S) * C * ps + d0 - shift0 * tri * (2 * lens + 1) / 3 + Pin * tri   # sum (n + 1) p_n
    after = ps[::-1].cumsum()[::-1] - ps                    # sum of p over the chunks behind
    carP[:, 0] = Pin
    carZ[:, 0] = -after
    fix = shift0 + ws.sum() / N                             # shift0 - mean(z),  sum z = -sum (n + 1) p_n
    return carP, carZ, shift0, fix


def apply(R, lam, h, carP, carZ, shift0, fix, C):
    N, M = R.shape
    S = carP.shape[0]
    Zout = np.zeros_like(R)
    rho = 0.0
    for s in range(S):
        blk = R[s * C:(s + 1) * C].copy()
        blk[:, 0] -= shift0
        ln = blk.shape[0]
        p = np.zeros_like(blk)
        prev = carP[s].copy()
        for k in range(ln):
            prev = blk[k] + lam * prev
            p[k] = prev
        z = carZ[s].copy()
        dsq = np.zeros(M)
        zsq = np.zeros(M)
        for k in range(ln - 1, -1, -1):
            zn = lam * (z - p[k])
            if not (s == S - 1 and k == ln - 1):            # no difference across the reflecting end
                dsq += (z - zn) ** 2
            zsq += zn ** 2
            z = zn
            Zout[s * C + k] = zn
        rr = -dsq - 2 * h * zsq
        rr[0] *= 0.5
        rho += rr.sum()
    Zout[:, 0] += fix
    rho += 0.5 * N * shift0 ** 2
    return Zout, rho / (2.0 * M)


def reference_column_solve(R):
    """idct(dct(R) / (lambda_k + mu_j)) along axis 0, the DC divisor replaced by 1 (phase_unwrap.py:106-115)"""
    N, M = R.shape
    lam_k = 2 * np.cos(np.pi * np.arange(N) / N) - 2
    mu_j = 2 * np.cos(np.pi * np.arange(M) / M) - 2
    scale = lam_k[:, None] + mu_j[None, :]
    scale[0, 0] = 1.0
    return sfft.idct(sfft.dct(R, axis=0) / scale, axis=0)


@pytest.mark.parametrize('n,C', [(64, 32), (64, 64), (96, 32), (100, 32), (250, 64), (256, 128), (77, 32)])
def test_streamed_column_solve_model_equals_dct_solve(n, C):
    rng = np.random.default_rng(n + C)
    R = rng.standard_normal((n, n))
    R[:, 0] += 0.3                                          # a column 0 with a mean
    h, lam = column_constants(n, n)
    a, b, c0, d0 = agg(R, lam, C)
    carP, carZ, shift0, fix = scan(a, b, c0, d0, lam, n, C)
    Z, rho = apply(R, lam, h, carP, carZ, shift0, fix, C)
    ref = reference_column_solve(R)
    assert np.abs(Z - ref).max() < 5e-10 * max(1.0, np.abs(ref).max())
    # rho = <r, z> of the 2-D preconditioner in SciPy's DCT normalisation: sum_j c_j / (2 M) sum_n R[n, j] Z[n, j], c_0 = 1/2
    cj = np.ones(n)
    cj[0] = 0.5
    rho_ref = ((R * ref).sum(axis=0) * cj).sum() / (2.0 * n)
    assert abs(rho - rho_ref) < 1e-9 * abs(rho_ref)


def test_chunk_zsum_closed_form():
    """e_s = -sum_k lam^(k+1) p_k for the zero-start recursion equals -g (a - lam^(len+1) b)"""
    rng = np.random.default_rng(3)
    for lam in (0.3, 0.9, 0.9995):
        for ln in (1, 7, 64):
            r = rng.standard_normal(ln)
            p = np.zeros(ln)
            prev = 0.0
            for k in range(ln):
                prev = r[k] + lam * prev
                p[k] = prev
            e = -(lam ** (np.arange(ln) + 1) * p).sum()
            a = (lam ** np.arange(ln) * r).sum()
            b = (lam ** (ln - 1 - np.arange(ln)) * r).sum()
            assert abs(e - (-lam / (1 - lam * lam) * (a - lam ** (ln + 1) * b))) < 1e-10 * max(1.0, abs(e))
