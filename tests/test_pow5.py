"""a16: fresnel_schlick() raises 1 - u to the fifth power with the fp64 pow() (main.c:126-129).
The kernels (rt_kernels.hip, both trace kernels) and the oracle use x2*x2*x in fp64 instead.  This test pins
the equivalence exhaustively: every float u in [0, 1] -- the range clamp() leaves (main.c:214-216) -- gives
the same float.  ~1e9 evaluations of libm's pow, a few seconds on 8 threads."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_pow5_equals_x2_x2_x_for_every_float_in_unit_interval(tmp_path):
    exe = tmp_path / "pow5_check"
    # -ffp-contract=off: x2 * x2 * x must be two separately rounded products, as in the kernels
    subprocess.run(["gcc", "-std=c11", "-O2", "-ffp-contract=off", "-o", str(exe),
                    os.path.join(HERE, "csrc", "pow5_check.c"), "-lm", "-lpthread"], check=True)
    out = subprocess.run([str(exe), str(min(os.cpu_count() or 1, 16))], capture_output=True, text=True, timeout=900)
    total, bad, example = out.stdout.split()
    assert int(total) == 0x3f800001
    assert out.returncode == 0 and int(bad) == 0, f"{bad} floats differ, e.g. bits 0x{example}"
