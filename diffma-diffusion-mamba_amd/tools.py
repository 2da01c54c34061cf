"""Spiral scan orders (reference tools.py:2-43), built with numpy.

`spiral(n)` returns the reference's two lists of 16 permutations of the n*n raster-ordered tokens:
    matrix_list[2*i]     [p] = spiral rank of raster cell p for direction set i (0 = centre cell)
    matrix_list[2*i + 1] [p] = n*n - 1 - that rank                       (the reversed walk)
    original_order_indexes_list[k] = inverse permutation of matrix_list[k]
The walk starts at (n//2, n//2), takes runs of length 1,1,2,2,3,3,... turning through the 4 unit steps
of the direction set, and numbers only the cells that fall inside the n x n grid (tools.py:20-29).
NB: CrossScan uses these as `x[..., list]`, i.e. scan position p reads token list[p].
"""
from __future__ import annotations

import numpy as np

# the 8 (first step, turn sense) combinations, in the reference's order (tools.py:4-11)
_DIRECTION_SETS = (
    ((0, 1), (1, 0), (0, -1), (-1, 0)),
    ((1, 0), (0, -1), (-1, 0), (0, 1)),
    ((0, -1), (-1, 0), (0, 1), (1, 0)),
    ((-1, 0), (0, 1), (1, 0), (0, -1)),
    ((0, 1), (-1, 0), (0, -1), (1, 0)),
    ((0, -1), (1, 0), (0, 1), (-1, 0)),
    ((1, 0), (0, 1), (-1, 0), (0, -1)),
    ((-1, 0), (0, -1), (1, 0), (0, 1)),
)


def _spiral_rank(n: int, steps) -> np.ndarray:
    """rank[r, c] = order in which the spiral walk visits cell (r, c)."""
    rank = np.full((n, n), -1, dtype=np.int64)
    r = c = n // 2
    placed, run, turn = 0, 1, 0
    total = n * n
    while placed < total:
        for _ in range(2):
            dr, dc = steps[turn % 4]
            for _ in range(run):
                if 0 <= r < n and 0 <= c < n:
                    rank[r, c] = placed
                    placed += 1
                r += dr
                c += dc
            turn += 1
        run += 1
    return rank.reshape(-1)


def spiral_arrays(n: int):
    """(orders [16, n*n] int64, inverses [16, n*n] int64)."""
    orders = []
    for steps in _DIRECTION_SETS:
        rk = _spiral_rank(n, steps)
        orders.append(rk)
        orders.append(n * n - 1 - rk)
    orders = np.stack(orders)
    inverses = np.argsort(orders, axis=1, kind="stable")
    return orders, inverses


def spiral(n: int):
    """Same return convention as the reference: two lists of 16 Python lists."""
    orders, inverses = spiral_arrays(n)
    return [row.tolist() for row in orders], [row.tolist() for row in inverses]


# ------------------------------------------------------------------------------------------------
# Scan orders of the baseline blocks (reference tools.py:46-152)
# ------------------------------------------------------------------------------------------------
def _zig_rank(n: int, variant: int) -> np.ndarray:
    """rank[r*n + c] of the reference's zig<variant>(n) matrix (0-based), variant in 1..8.

    zig1 numbers the cells row by row, every other row right-to-left (a boustrophedon); zig2 does the same column by
    column.  The other six are mirror images: 3/4 mirror the columns, 5/6 the rows, 7/8 both (tools.py:46-100)."""
    r, c = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    if variant in (3, 4, 7, 8):
        c = n - 1 - c
    if variant in (5, 6, 7, 8):
        r = n - 1 - r
    if variant % 2 == 1:                                   # rows first
        rank = r * n + np.where(r % 2 == 0, c, n - 1 - c)
    else:                                                  # columns first
        rank = c * n + np.where(c % 2 == 0, r, n - 1 - r)
    return rank.reshape(-1).astype(np.int64)


def zig(n: int, i: int):
    """(rearrange_list, original_order_indexes) of ZigMa block i (reference tools.py:102-128): variant i % 8, with 0 -> 8."""
    order = _zig_rank(n, i % 8 if i % 8 else 8)
    return order.tolist(), np.argsort(order, kind="stable").tolist()


def vmamba_(n: int):
    """(order_list, original_list): the four VMamba scan orders zig1, zig2, zig7, zig8 and their inverses (tools.py:130-152)."""
    orders = [_zig_rank(n, v) for v in (1, 2, 7, 8)]
    return [o.tolist() for o in orders], [np.argsort(o, kind="stable").tolist() for o in orders]


def efficient_scan_tokens(n: int) -> np.ndarray:
    """[4, (n/2)^2] raster token ids visited by the four atrous scans of EfficientVMamba (block/mamba.py:169-180):
    (even rows, even cols) row-major; (odd rows, even cols) column-major; (even rows, odd cols) row-major;
    (odd rows, odd cols) column-major.  Each token belongs to exactly one scan; n must be even."""
    if n % 2:
        raise ValueError("EfficientVMamba's 2x2 atrous split needs an even token grid (the reference fails for odd n too)")
    tok = np.arange(n * n).reshape(n, n)
    return np.stack([tok[::2, ::2].reshape(-1), tok.T[::2, 1::2].reshape(-1), tok[::2, 1::2].reshape(-1),
                     tok.T[1::2, 1::2].reshape(-1)]).astype(np.int64)
