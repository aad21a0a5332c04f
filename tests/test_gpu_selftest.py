"""On-GPU check that the tuned kernel's exact-arithmetic shortcuts (shared-reciprocal division in
f32 and f64, shared-reciprocal normalisation) return the same bits as the plain IEEE operations they
replace, on ~10^9 random operand sets each (plus near-power-of-two, correlated, zero, denormal, huge
and epsilon-threshold operands)."""
import pytest

import ray_tracing_amd as rt

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("which,name", [(0, "f32 divide"), (1, "f64 divide"), (2, "normalize"), (3, "f64 sqrt of float"), (4, "iszerof threshold")])
def test_shortcuts_are_bit_exact(which, name):
    g = rt.Renderer(0)
    total = 0
    for seed in (1, 2, 3, 4):
        out = g.selftest(which, seed=seed, blocks=4096, iters=256)      # 2.7e8 cases per call
        total += 4096 * 256 * 256
        assert out[0] == 0, f"{name}: {out[0]} mismatches, e.g. operands {[hex(x) for x in out[1:6]]}"
    g.close()
    print(f"{name}: {total:.2e} cases, 0 mismatches")
