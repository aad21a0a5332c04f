"""On-GPU check that the tuned kernel's exact-arithmetic shortcuts (shared-reciprocal division in
f32 and f64, shared-reciprocal normalisation) return the same bits as the plain IEEE operations they
replace, on ~10^9 random operand sets each (plus near-power-of-two, correlated, zero, denormal, huge
and epsilon-threshold operands)."""
import pytest

import ray_tracing_amd as rt

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("which,name", [(0, "f32 divide"), (1, "f64 divide"), (2, "normalize"), (3, "f64 sqrt of float"), (4, "iszerof threshold"), (7, "normalize a random direction"),
                                        (8, "side of a surface a random vector points to, before it is normalised")])
def test_shortcuts_are_bit_exact(which, name):
    g = rt.Renderer(0)
    total = 0
    for seed in (1, 2, 3, 4):
        out = g.selftest(which, seed=seed, blocks=4096, iters=256)      # 2.7e8 cases per call
        total += 4096 * 256 * 256
        assert out[0] == 0, f"{name}: {out[0]} mismatches, e.g. operands {[hex(x) for x in out[1:6]]}"
    g.close()
    print(f"{name}: {total:.2e} cases, 0 mismatches")


def test_exact_division_every_significand():
    """rcp_refined == RN(1/d) for all 2^23 significands, and the 3-instruction quotient == `/` for every
    denominator significand x 4 x 2048 numerator significands (6.9e10 pairs here; the sweep over ALL 2^46
    pairs -- iters=1<<23, about a minute, 0 mismatches -- runs with RT_FULL_SWEEP=1)."""
    import os
    g = rt.Renderer(0)
    if os.environ.get("RT_FULL_SWEEP"):
        out = g.selftest(5, seed=0, blocks=8192, iters=1 << 23)
        assert out[0] == 0, f"{out[0]} mismatches, e.g. {[hex(x) for x in out[1:6]]}"
    else:
        for seed in (0, 1234567, 4242424, 8000001):
            out = g.selftest(5, seed=seed, blocks=8192, iters=2048)
            assert out[0] == 0, f"{out[0]} mismatches, e.g. {[hex(x) for x in out[1:6]]}"
    g.close()


def test_tuned_sqrt_every_float_in_window():
    g = rt.Renderer(0)
    out = g.selftest(6, blocks=8192, iters=1)
    assert out[0] == 0, f"{out[0]} mismatches, e.g. x={hex(out[1])} want={hex(out[3])} got={hex(out[4])}"
    g.close()
