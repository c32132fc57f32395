"""numpy prototype of the index arithmetic of formulation D (coarse partitions): a real FFT of N = 16384 points through two
complex FFTs of 4096 points (z_a[m] = x[4m] + i x[4m+2], z_b[m] = x[4m+1] + i x[4m+3]) plus one combine pass, its inverse,
the packed spectrum layout (bin 0 = (X[0], X[8192])) and the partitioned overlap-save convolution built on them.
Run: python tools/proto/coarse_math.py  (checks against numpy.fft / numpy.convolve)."""
import numpy as np

CB = 8192
N = 2 * CB
M = 4096


def fwd_packed(x):
    """x: real [16384] -> packed spectrum [8192] complex: P[0] = X[0] + i X[8192], P[k] = X[k]."""
    x = np.asarray(x, np.float64)
    za = np.fft.fft(x[0::4] + 1j * x[2::4])
    zb = np.fft.fft(x[1::4] + 1j * x[3::4])
    out = np.zeros(CB, complex)
    k = np.arange(0, 2049)
    a = np.exp(-2j * np.pi * k / 8192)      # W_8192^k
    b = np.exp(-2j * np.pi * k / 16384)     # W_16384^k
    kk = (M - k) % M
    def split(z):
        zk, zm = z[k % M], np.conj(z[kk])
        fe, fo = 0.5 * (zk + zm), -0.5j * (zk - zm)
        Ek = fe + a * fo                     # E[k]
        Emk = np.conj(fe - a * fo)           # E[4096 - k]
        return Ek, Emk
    Ek, Emk = split(za)
    Ok, Omk = split(zb)
    Xk = Ek + b * Ok                                   # X[k]
    Xmk = Emk + (-1j) * np.conj(b) * Omk               # X[4096 - k]      W^(4096-k) = -i conj(W^k)
    Xpk = np.conj(Emk) + (-1j) * b * np.conj(Omk)      # X[4096 + k]      E[4096+k] = conj(E[4096-k]),  W^(4096+k) = -i W^k
    Xnk = np.conj(Ek) - np.conj(b) * np.conj(Ok)       # X[8192 - k]      W^(8192-k) = -conj(W^k)
    out[k[1:]] = Xk[1:]
    out[4096 - k] = Xmk
    out[(4096 + k)[:-1]] = Xpk[:-1]     # k = 2048 -> 6144 written by Xmk... (4096-2048 = 2048, 4096+2048 = 6144)
    out[4096 + 2048] = Xpk[2048]
    out[(8192 - k)[1:]] = Xnk[1:]
    out[0] = Xk[0].real + 1j * Xnk[0].real
    return out


def inv_packed_second_half(P):
    """packed spectrum -> the last 8192 samples of the length-16384 real inverse transform (overlap-save keeps these)."""
    X = np.zeros(CB + 1, complex)
    X[1:CB] = P[1:]
    X[0] = P[0].real
    X[CB] = P[0].imag
    k = np.arange(0, 2049)
    a = np.exp(-2j * np.pi * k / 8192)
    b = np.exp(-2j * np.pi * k / 16384)
    # E[k] = (X[k] + conj(X[8192-k]))/2 ; O[k] = (X[k] - conj(X[8192-k]))/2 * conj(W^k)    k = 0..8191
    def EO(kv, w):
        xk, xn = X[kv], np.conj(X[8192 - kv])
        return 0.5 * (xk + xn), 0.5 * (xk - xn) * np.conj(w)
    Ek, Ok = EO(k, b)                              # bins k
    Emk, Omk = EO(4096 - k, -1j * np.conj(b))      # bins 4096 - k
    # Z[k] = Fe[k] + i Fo[k],  Fe[k] = (E[k] + conj(E[4096-k]))/2,  Fo[k] = (E[k] - conj(E[4096-k]))/2 * conj(W_8192^k)
    def Z(Ek_, Emk_):
        fe = 0.5 * (Ek_ + np.conj(Emk_))
        fo = 0.5 * (Ek_ - np.conj(Emk_)) * np.conj(a)
        zk = fe + 1j * fo
        zmk = np.conj(fe) + 1j * np.conj(fo)       # Z[4096-k] = conj(Fe[k]) + i conj(Fo[k])
        return zk, zmk
    zak, zamk = Z(Ek, Emk)
    zbk, zbmk = Z(Ok, Omk)
    za = np.zeros(M, complex)
    zb = np.zeros(M, complex)
    za[k % M] = zak
    zb[k % M] = zbk
    za[(4096 - k) % M] = zamk
    zb[(4096 - k) % M] = zbmk
    za[0] = zak[0]
    zb[0] = zbk[0]
    ya, yb = np.fft.ifft(za), np.fft.ifft(zb)
    x = np.zeros(N)
    x[0::4], x[2::4] = ya.real, ya.imag
    x[1::4], x[3::4] = yb.real, yb.imag
    return x[CB:]


def packed_mul(A, B):
    """bin-wise product of two packed spectra (bin 0 holds two real bins)."""
    C = A * B
    C[0] = A[0].real * B[0].real + 1j * A[0].imag * B[0].imag
    return C


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    x = rng.standard_normal(N)
    P = fwd_packed(x)
    ref = np.fft.rfft(x)
    assert np.allclose(P[1:], ref[1:CB], atol=1e-9), np.abs(P[1:] - ref[1:CB]).max()
    assert abs(P[0].real - ref[0].real) < 1e-9 and abs(P[0].imag - ref[CB].real) < 1e-9
    assert np.allclose(inv_packed_second_half(P), x[CB:], atol=1e-10)
    # partitioned overlap-save: taps 3*CB - 100, signal of 5.3 coarse blocks with history
    taps = 3 * CB - 100
    h = rng.standard_normal(taps) * np.exp(-np.arange(taps) / 4000.0)
    Pn = (taps + CB - 1) // CB
    hp = np.zeros(Pn * CB)
    hp[:taps] = h
    H = [fwd_packed(np.concatenate([hp[p * CB:(p + 1) * CB], np.zeros(CB)])) for p in range(Pn)]
    n = int(5.3 * CB) // 128 * 128
    hist = rng.standard_normal(Pn * CB)      # samples before the chunk
    sig = rng.standard_normal(n)
    full = np.concatenate([hist, sig, np.zeros(2 * CB)])
    def sample(i0, i1):                       # chunk-relative sample range, zero beyond the chunk
        return full[Pn * CB + i0: Pn * CB + i1] * 1.0
    nT = (n + CB - 1) // CB
    X = {u: fwd_packed(np.concatenate([sample((u - 1) * CB, u * CB), np.where(np.arange(u * CB, (u + 1) * CB) < n, sample(u * CB, (u + 1) * CB), 0.0)]))
         for u in range(-(Pn - 1), nT)}
    out = np.zeros(nT * CB)
    for t in range(nT):
        Y = np.zeros(CB, complex)
        for p in range(Pn):
            Y += packed_mul(X[t - p], H[p])
        out[t * CB:(t + 1) * CB] = inv_packed_second_half(Y)
    want = np.convolve(np.concatenate([hist, sig]), h)[Pn * CB: Pn * CB + n]
    err = np.abs(out[:n] - want).max()
    print("overlap-save max err", err)
    assert err < 1e-8
    print("ok")


def fwd_packed_v2(x):
    """Same spectrum through z_a[m] = x[4m] + i x[4m+1], z_b[m] = x[4m+2] + i x[4m+3] (each complex point is one aligned pair of
    input samples: a thread loads x[4m..4m+3] as one 16-byte word and owns point m of BOTH transforms).  Returns 2 X (the factor
    the kernels fold into the impulse-response scale)."""
    x = np.asarray(x, np.float64)
    A = np.fft.fft(x[0::4] + 1j * x[1::4])
    B = np.fft.fft(x[2::4] + 1j * x[3::4])
    out = np.zeros(CB, complex)
    for k in range(0, 2049):
        b = np.exp(-2j * np.pi * k / 16384)
        a = b * b
        Ak, Am, Bk, Bm = A[k % M], A[(M - k) % M], B[k % M], B[(M - k) % M]
        aB, caBm = a * Bk, np.conj(a) * Bm
        Wk, Wp = Ak + aB, Ak - aB
        Wm, Wn = Am - caBm, Am + caBm
        fe, fo = Wk + np.conj(Wn), -1j * (Wk - np.conj(Wn))
        Xk, Xn = fe + b * fo, np.conj(fe - b * fo)
        fe2, fo2 = Wm + np.conj(Wp), -1j * (Wm - np.conj(Wp))
        w2 = -1j * np.conj(b)
        Xm, Xp = fe2 + w2 * fo2, np.conj(fe2 - w2 * fo2)
        if k == 0:
            out[0] = Xk.real + 1j * Xn.real
            out[M] = Xm
        else:
            out[k], out[2 * M - k] = Xk, Xn
            if k != M // 2:
                out[M - k], out[M + k] = Xm, Xp
    return out


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    x = rng.standard_normal(N)
    assert np.allclose(fwd_packed_v2(x), 2 * fwd_packed(x), atol=1e-8)
    print("v2 forward ok")
