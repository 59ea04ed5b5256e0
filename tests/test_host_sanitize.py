"""The host half of the library under AddressSanitizer + UndefinedBehaviorSanitizer (tests/host_sanitize_check.cpp): every
scene of the assets loaded, described, created again from its description and compared; the image decoders and the PNG
writer; and more than a thousand damaged scene / resources / OBJ / PNG / JPEG files, which must end in a status code."""
import os
import subprocess

import raytracer_amd as rt

ROOT = rt.ROOT


def test_host_code_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "host_sanitize_check")
    c = os.path.join(ROOT, "raytracer_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-static-libasan", "-static-libubsan",   # (the runtime inside the program: it need not be first among the loaded libraries)
                           "-fno-omit-frame-pointer", "-pthread", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "host_sanitize_check.cpp"), os.path.join(c, "rt_host.cpp"), os.path.join(c, "rt_image.cpp"),
                           "-lz", "-o", exe])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", RT_AMD_BUILD_THREADS="4")
    out = subprocess.run([exe, rt.ASSETS, str(tmp_path), "1200", "7"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-4000:]
    last = out.stdout.strip().splitlines()[-1]
    assert last.startswith("damaged inputs:") and last.endswith("failures 0"), out.stdout[-2000:]
    accepted, refused = int(last.split()[2]), int(last.split()[4])
    assert refused > 300 and accepted > 100, last   # (both happen: the test damages enough, and not everything)
