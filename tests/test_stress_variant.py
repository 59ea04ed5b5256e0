"""The parity tests once more against the build of the library whose waves take EVERY leaf through the loop with four triangle
records in flight (Makefile: librt_amd_stress.so).  The production build takes that loop only for large leaves of waves with a
long life - few leaves of few waves at the test sizes - so this is where every scene, kernel variant and re-entry walks it."""
import os
import subprocess
import sys

import pytest

import raytracer_amd as rt


@pytest.mark.gpu
def test_parity_with_the_long_wave_leaf_loop_on_every_leaf():
    lib = os.path.join(rt.ROOT, "raytracer_amd", "librt_amd_stress.so")
    assert os.path.exists(lib), "build it with `make` (it is one of the default targets)"
    env = dict(os.environ, RT_AMD_LIB=lib)
    out = subprocess.run([sys.executable, "-u", "-m", "pytest", os.path.join(rt.ROOT, "tests", "test_gpu_parity.py"), "-x", "-q", "-m", "gpu",
                          "--timeout", "120", "--timeout-method", "thread", "-p", "no:cacheprovider"],
                         cwd=rt.ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:]
    assert " passed" in out.stdout and "failed" not in out.stdout, out.stdout[-1500:]


@pytest.mark.gpu
def test_decision_routine_of_the_block_against_the_compiled_form():
    # three correctly rounded divisions and seven compares written in assembly (tools/gen_trace_asm.py, finish()), run on four
    # million cases - edges, zero numerators, zero determinants, NaN - beside the compiled rt_triangle_finish: same bits
    exe = os.path.join(rt.ROOT, "raytracer_amd", "finish_check")
    assert os.path.exists(exe), "build it with `make` (it is one of the default targets)"
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert out.returncode == 0 and " 0 differ" in out.stdout, out.stdout[-2000:]
