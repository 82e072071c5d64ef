"""The claim behind the FORM_TABLE kernel builds (csrc/trace_device.hpp, x_thresholds), restated in numpy and checked on the CPU:
treeLookup's x index  int(round_even(((v + f) * fl(1 / cc)) * (2 cc) - 0.5))  (raytracer.comp:376-378 as compiled) is,
for a fixed cell index v, the step function  2v - 1 + (f >= F0) + (f >= F1) + (f >= F2)  of the level's coordinate f in [0, 1).
The GPU suite checks this exhaustively (tdt_selftest_index: every f x every cell); here the thresholds are found by the same
bisection in float32 numpy and the claim is checked on every float within 48 ulps of each threshold, on the ends of the interval
and on 4096 random coordinates per cell — for the reference's own cell_count = 100000 (main.rs:459) and a few others."""
import numpy as np
import pytest

ONE = np.uint32(0x3F800000)


def x_index(v, f, cc):
    """the literal formula, float32 step by step (numpy float32 arithmetic is IEEE, one rounding per operation)"""
    ic = np.float32(1.0) / np.float32(cc)                       # octree.rs:49
    two_cc = np.float32(np.int32(cc << 1))
    s = (v.astype(np.float32) + f).astype(np.float32)
    x = (s * ic).astype(np.float32)
    x = (x * two_cc).astype(np.float32)
    x = (x + np.float32(-0.5)).astype(np.float32)
    return np.rint(x).astype(np.int64)                          # round half to even


def first_at_least(v, target, cc):
    """per cell: bit pattern of the smallest f in [0, 1) with x_index(v, f) >= target (ONE: none), by bisection"""
    lo = np.zeros(v.shape, np.uint32)
    hi = np.full(v.shape, ONE, np.uint32)
    for _ in range(31):
        mid = ((lo.astype(np.uint64) + hi.astype(np.uint64)) >> 1).astype(np.uint32)
        ge = x_index(v, mid.view(np.float32), cc) >= target
        hi = np.where(ge & (lo < hi), mid, hi)
        lo = np.where(~ge & (lo < hi), mid + 1, lo)
    return lo


@pytest.mark.parametrize("cc", [100000, 99999, 12345, 65537, 3000, 1000003, 1 << 16])
def test_x_index_is_a_three_step_function_of_the_coordinate(cc):
    rng = np.random.default_rng(cc)
    v = np.concatenate([np.arange(0, 600), rng.integers(600, 5120, 400)]).astype(np.int64)
    base = 2 * v
    top = np.array([0x3F7FFFFF], np.uint32).view(np.float32)[0]      # 1 - 2^-24
    g0 = x_index(v, np.zeros(v.shape, np.float32), cc)
    g1 = x_index(v, np.full(v.shape, top, np.float32), cc)
    assert ((g0 == base) | (g0 == base - 1)).all() and (g0[v == 0] == 0).all()
    assert ((g1 == base + 1) | (g1 == base + 2)).all()
    F = [first_at_least(v, base + k, cc) for k in (0, 1, 2)]      # bit patterns of F0, F1, F2 (ONE = never)

    def table(fbits):
        return base[:, None] - 1 + sum((fbits >= Fk[:, None]).astype(np.int64) for Fk in F)

    # around every threshold, the ends of [0, 1), and random coordinates
    probes = [np.zeros((v.size, 1), np.uint32), np.full((v.size, 1), 0x3F7FFFFF, np.uint32),
              rng.integers(0, int(ONE), (v.size, 4096), dtype=np.uint32)]
    for Fk in F:
        around = Fk[:, None].astype(np.int64) + np.arange(-48, 49)[None, :]
        probes.append(np.clip(around, 0, int(ONE) - 1).astype(np.uint32))
    fbits = np.concatenate(probes, axis=1)
    lit = x_index(np.broadcast_to(v[:, None], fbits.shape), fbits.view(np.float32), cc)
    assert (lit == table(fbits)).all()
    if cc & (cc - 1) == 0:
        # a power-of-two count: the products are exact, so never 2v - 1, and the steps are q > 1/2 and q == 1 for q = fl(v + f) - v
        assert (F[0] == 0).all()
        f = fbits.view(np.float32)
        fv = np.broadcast_to(v[:, None].astype(np.float32), f.shape)
        q = ((fv + f).astype(np.float32) - fv).astype(np.float32)
        assert (lit == 2 * v[:, None] + (q > 0.5) + (q == 1.0)).all()


def test_the_references_own_count_reads_the_previous_cell_at_the_bottom_of_some_cells():
    """cell_count = 100000: for cell 7 (and 11, 14, 15, ...) a coordinate below half an ulp of the cell index gives 2v - 1 — the
    upper half of the PREVIOUS cell.  The reference really reads that node there, so the kernels must (and do: F0)."""
    v = np.array([7, 11, 14, 15], np.int64)
    assert (x_index(v, np.zeros(4, np.float32), 100000) == 2 * v - 1).all()
    assert (x_index(np.array([1, 2, 3, 4, 5, 6, 8]), np.zeros(7, np.float32), 100000) == 2 * np.array([1, 2, 3, 4, 5, 6, 8])).all()
    F0 = first_at_least(v, 2 * v, 100000).view(np.float32)
    assert (F0 > 0).all() and (F0 < 1e-6).all()                   # about half an ulp of the cell index


def test_unrelated_uniforms_are_not_of_the_shape():
    """an inv_cell_count that is not 1 / cell_count: the index is not 2v + steps — such scenes must run the literal kernel"""
    v = np.arange(1, 50)
    ic, two_cc = np.float32(2e-5), np.float32(200000.0)
    g0 = np.rint(((v.astype(np.float32) * ic).astype(np.float32) * two_cc).astype(np.float32) + np.float32(-0.5)).astype(np.int64)
    assert (g0 != 2 * v).all() and (g0 != 2 * v - 1).all()
