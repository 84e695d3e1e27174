"""OpenCV-style "bit exact" Gaussian taps, host side
(reference src/GaussianKernelBitExact.py: getGaussianKernelBitExact).

Evaluated in 28-digit decimal arithmetic like the reference.  Quirk kept on purpose: a POSITIVE sigma
is ignored and the width n*0.15+0.35 is used; only a negative sigma is taken as given
(reference :102-106)."""
from decimal import Decimal

import numpy as np

softOne = Decimal('1.0')
softZero = Decimal('0.0')

_TABLES = {
    1: (1.0,),
    3: (0.25, 0.5, 0.25),
    5: (0.0625, 0.25, 0.375, 0.25, 0.0625),
    7: (0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125),
    9: (4 / 256, 13 / 256, 30 / 256, 51 / 256, 60 / 256, 51 / 256, 30 / 256, 13 / 256, 4 / 256),
}


def getGaussianKernelBitExact(n, sigma):
    """Returns (sum of the taps, taps) like the reference; taps as float64."""
    assert (n > 0)
    if sigma <= 0 and n in _TABLES:
        if n == 1:
            return softOne, np.array([softOne])
        return softOne, np.array(_TABLES[n], dtype=np.float64)
    width = Decimal(sigma) if sigma < 0 else Decimal(n) * Decimal('0.15') + Decimal('0.35')
    coef = Decimal('-0.125') / (width * width)
    half = int((n - 1) / 2)
    tail = []
    total = softZero
    for i in range(half):
        x = 1 - n + 2 * i                     # twice the tap offset
        t = (Decimal(x * x) * coef).exp()
        tail.append(t)
        total += t
    total *= Decimal(2.0)
    total += softOne
    if (n & 1) == 0:
        total += softOne
    norm = softOne / total
    taps = [softZero] * n
    check = softZero
    for i, t in enumerate(tail):
        taps[i] = taps[n - 1 - i] = t * norm
        check += t * norm
    check *= Decimal(2.0)
    taps[half] = softOne * norm
    check += taps[half]
    if (n & 1) == 0:
        taps[half + 1] = taps[half]
        check += taps[half]
    return np.float64(check), np.array([float(t) for t in taps], dtype=np.float64)
