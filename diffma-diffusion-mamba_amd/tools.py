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
