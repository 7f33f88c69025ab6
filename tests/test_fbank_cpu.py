"""Host-side check of the tree shape csrc/fbank.hip's k_sumsq assumes: numpy's pairwise sum over a chunk of n <= 8192
float32 elements splits while a range is longer than 128 (n2 = n/2 rounded down to a multiple of 8).  The kernel gives
the tree 256 heap slots (depth <= 7); this walks every n and asserts that bound, and that the recursion -- restated here
exactly as the kernel walks it -- reproduces np.sum bit for bit (so the kernel's order IS numpy's order)."""
import numpy as np


def _tree(x):
    """-> (sum in the kernel's order, depth of the deepest leaf)"""
    n = len(x)
    if n <= 128:
        if n < 8:
            r = np.float32(0)
            for v in x:
                r = np.float32(r + v)
            return r, 0
        full = n - n % 8
        acc = x[:8].copy()
        for i in range(8, full, 8):
            acc = (acc + x[i:i + 8]).astype(np.float32)
        r = np.float32(np.float32(np.float32(acc[0] + acc[1]) + np.float32(acc[2] + acc[3])) +
                       np.float32(np.float32(acc[4] + acc[5]) + np.float32(acc[6] + acc[7])))
        for v in x[full:]:
            r = np.float32(r + v)
        return r, 0
    n2 = n // 2
    n2 -= n2 % 8
    a, da = _tree(x[:n2])
    b, db = _tree(x[n2:])
    return np.float32(a + b), 1 + max(da, db)


def test_pairwise_tree_depth_and_order():
    rng = np.random.Generator(np.random.PCG64(5))
    x = (rng.standard_normal(8192) ** 2).astype(np.float32)
    deepest = 0
    for n in list(range(1, 8193, 61)) + list(range(7600, 8193)) + [129, 135, 136, 263, 519, 1031, 2055, 4103]:
        s, d = _tree(x[:n])
        deepest = max(deepest, d)
        assert s.tobytes() == np.sum(x[:n]).tobytes(), n
    assert deepest == 7  # kPwDepth in csrc/fbank.hip


def test_depth_bound_for_all_lengths():
    def depth(n):
        if n <= 128:
            return 0
        n2 = n // 2
        n2 -= n2 % 8
        return 1 + max(depth(n2), depth(n - n2))
    assert max(depth(n) for n in range(1, 8193)) == 7
