"""numpy emulation of the fused body kernel's *schedule* (TEST INFRASTRUCTURE ONLY).

Design-validation tool for stofnet_amd/csrc/body_sweep.hip: it executes the
same left-to-right sweep over a stream of waveforms with two LDS rings (X:
residual stream, Y: intermediates), skewed per-layer frontiers and zero masks,
in float64, so that index arithmetic (ring wrap, frontier lag, gap rows,
in-place residual update) can be checked against oracle.stofnet_forward
without a GPU.  See DESIGN.md "body sweep".
"""
import numpy as np

GAP = 4          # zero rows between consecutive waveforms in the stream (conv1 pad)
NLAYER = 13      # j = 0 (conv1+SGB) .. 11 (conv12), 12 (conv_last)
LAG = [0] + [3 * j for j in range(1, 12)] + [34]


def lrelu(v):
    return np.where(v > 0, v, 0.01 * v)


def sweep_forward(params, x, sgb, r, S=192, RING=256, n_begin=0, n_end=None):
    """x [N,1,L] float; sgb [N,64,P] (post-activation expand map) or None.
    Returns y [n_end-n_begin, 1, L*r] computed by the sweep schedule."""
    N, _, L = x.shape
    n_end = N if n_end is None else n_end
    Lp = L + GAP
    P = L // 80
    rem = L - 80 * P
    g_begin, g_end = n_begin * Lp, n_end * Lp
    f64 = np.float64
    W = {k: np.asarray(v, f64) for k, v in params.items()}
    X = np.zeros((RING, 64), f64)
    Y = np.zeros((RING, 64), f64)
    out = np.zeros((n_end - n_begin, L * r), f64)

    def decode(g):
        """stream row -> (waveform, t, valid)"""
        n = g // Lp
        t = g - n * Lp
        valid = (g >= g_begin) & (g < g_end) & (t < L)
        return n, t, valid

    def raw(g):
        n, t, v = decode(g)
        return np.where(v, x[np.clip(n, 0, N - 1), 0, np.clip(t, 0, L - 1)], 0.0)

    def x0_rows(g):
        """relu(conv1) + SGB add for stream rows g (recomputed for the long skip too)"""
        n, t, v = decode(g)
        acc = np.zeros((len(g), 64), f64) + W['conv1.bias'][None, :]
        for d in range(9):
            acc += raw(g + d - 4)[:, None] * W['conv1.weight'][None, :, 0, d]
        acc = np.maximum(acc, 0)
        if sgb is not None:
            pos = t - rem // 2
            ok = (pos >= 0) & (pos < 80 * P) & v
            w = np.clip(pos // 80, 0, max(P - 1, 0))
            acc += np.where(ok[:, None], sgb[np.clip(n, 0, N - 1), :, w], 0.0)
        return np.where(v[:, None], acc, 0.0)

    def conv_rows(ring, g, wname, k):
        half = k // 2
        w = W[wname + '.weight']
        acc = np.zeros((len(g), w.shape[0]), f64) + W[wname + '.bias'][None, :]
        for d in range(k):
            acc += ring[(g + d - half) % RING] @ w[:, :, d].T
        return acc

    nsteps = -(-(g_end - g_begin - GAP + LAG[-1]) // S)
    for i in range(1, nsteps + 1):
        F = g_begin + i * S
        for j in range(NLAYER):
            g = np.arange(F - S - LAG[j], F - LAG[j])
            n, t, v = decode(g)
            slot = g % RING
            if j == 0:
                X[slot] = x0_rows(g)
            elif j <= 10:
                name = f'conv{j + 1}'
                if j % 2 == 1:      # conv2,4,..,10: Y = lrelu(conv(X))
                    Y[slot] = np.where(v[:, None], lrelu(conv_rows(X, g, name, 7)), 0.0)
                else:               # conv3,5,..,11: X += conv(Y)   (in place)
                    X[slot] = np.where(v[:, None], X[slot] + conv_rows(Y, g, name, 7), 0.0)
            elif j == 11:           # conv12: Y = x0 + conv12(X)
                Y[slot] = np.where(v[:, None], x0_rows(g) + conv_rows(X, g, 'conv12', 7), 0.0)
            else:                   # conv_last + shuffle store
                o = conv_rows(Y, g, 'conv_last', 3)          # [rows, r]
                for idx in np.nonzero(v)[0]:
                    out[n[idx] - n_begin, t[idx] * r:(t[idx] + 1) * r] = o[idx]
    return out[:, None, :]
