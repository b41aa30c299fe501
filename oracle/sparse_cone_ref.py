"""Dependency cone of the cost-regularisation net (TEST INFRASTRUCTURE ONLY — never imported by the product path).

The network reads the probability volume only at the chosen pixels (network_v5.py:449-455), so every tensor of CostRegNet
(network_v5.py:260-291) is needed only inside the cone of those pixels.  Per axis and per chosen coordinate y this file restates the
index intervals [lo, hi] that the device computes in csrc/prob_sparse.hip (cone_of): walking the graph backwards from the prob conv,
    Conv3d k3 pad 1 stride 1      (network_v5.py:17-28):  output i reads inputs i-1 .. i+1
    Conv3d k3 pad 1 stride 2:                              output i reads inputs 2i-1 .. 2i+1
    ConvTranspose3d k3 s2 p1 op1  (network_v5.py:246-252): output o = 2i - 1 + k, so o reads inputs floor(o/2) .. ceil(o/2)
    skip adds (network_v5.py:287-289) read their own index.
tests/test_sparse_cone.py checks the intervals against the oracle network itself: values outside the cone never reach the chosen
pixel's probabilities, values on its boundary do."""


def _clamp(a, b, n):
    return max(a, 0), min(b, n - 1)


def _tr(o, n):
    return _clamp(o[0] >> 1, (o[1] + 1) >> 1, n)


def _s1(o, n):
    return _clamp(o[0] - 1, o[1] + 1, n)


def _s2(o, n):
    return _clamp(2 * o[0] - 1, 2 * o[1] + 1, n)


def _or(p, q):
    return min(p[0], q[0]), max(p[1], q[1])


def cone(y, S):
    """Needed index interval of every tensor along one spatial axis for a pixel at coordinate y of an S-wide crop."""
    u11 = _clamp(y - 1, y + 1, S)                 # the prob conv (k3 p1) at the pixel
    u9 = _tr(u11, S // 2)
    u7 = _tr(u9, S // 4)
    c6 = _tr(u7, S // 8)
    c5 = _s1(c6, S // 8)
    c4 = _or(_s2(c5, S // 4), u7)
    c3 = _s1(c4, S // 4)
    c2 = _or(_s2(c3, S // 2), u9)
    c1 = _s1(c2, S // 2)
    c0 = _or(_s2(c1, S), u11)
    vol = _s1(c0, S)                              # conv0's input (the plane-sweep volume)
    return dict(u11=u11, u9=u9, u7=u7, c6=c6, c5=c5, c4=c4, c3=c3, c2=c2, c1=c1, c0=c0, vol=vol)


def sweep_tiles_needed(choose, S, th=12, tw=16):
    """Fraction of the depth-sweeping conv0's th x tw-pixel tiles that intersect the c0 box of any chosen pixel: choose [V, P]."""
    import numpy as np
    nth, ntw = -(-S // th), -(-S // tw)
    lo = np.array([cone(v, S)["c0"][0] for v in range(S)])
    hi = np.array([cone(v, S)["c0"][1] for v in range(S)])
    total = 0
    for ch in np.asarray(choose).reshape(-1, np.asarray(choose).shape[-1]):
        y, x = ch // S, ch % S
        m = np.zeros((nth, ntw), bool)
        for ra, rb, ca, cb in set(zip(lo[y] // th, hi[y] // th, lo[x] // tw, hi[x] // tw)):
            m[ra:rb + 1, ca:cb + 1] = True
        total += int(m.sum())
    return total / (len(np.asarray(choose).reshape(-1, np.asarray(choose).shape[-1])) * nth * ntw)
