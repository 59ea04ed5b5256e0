#!/usr/bin/env python3
"""Writes the per-scene golden fixtures of tests/golden/: for every runnable scene at 128x128 (max depth 3) the primary-hit
primitive ids, the bits of the primary-hit distances, the RGB8 image and the work counters, as rendered by the CPU oracle
(oracle/rt_oracle.cpp, the restatement of the reference's path) in THIS container, plus a manifest with the SHA-256 of every
array and the toolchain the oracle was built with.

The fixtures are data (inputs named by scene, expected outputs); the tests compare the HIP path with them as well as with the
oracle built on the GPU box, and a CPU test asserts that the oracle as built there still reproduces them, which catches a
drift of the oracle or of the host's libm between machines.  Undefined behaviour of the reference has one defined rule in the
oracle (refraction level index clamped to [0, 2], pushes onto a full ray stack dropped, negative colours stored as 0); the
counters of those events are part of the fixture.

   python3 tools/make_goldens.py        (from the repository root, after `make`)
"""
import hashlib
import json
import os
import platform
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_api import OracleScene          # noqa: E402
from scenes_common import ALL_SCENES        # noqa: E402

W = H = 128
DEPTH = 3


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    out_dir = os.path.join(ROOT, "tests", "golden", "scenes_128")
    os.makedirs(out_dir, exist_ok=True)
    manifest = {
        "what": "CPU oracle renders, %dx%d, max depth %d, resources.json defaults; arrays in <scene>.npz: ids int32[H][W] "
                "(triangle >= 0, sphere s = -2 - s, miss = -1), t_bits uint32[H][W] (IEEE bits of the primary-hit distance, 64.0 on a miss), "
                "rgb uint8[H][W][3] (row 0 = top)" % (W, H, DEPTH),
        "undefined_behaviour_rule": "refraction level index clamped to [0, 2] (material.cu:40,61,87); a push onto a full 10-ray stack is dropped; "
                                    "negative colours are stored as 0, NaN as 255",
        "made_by": "tools/make_goldens.py",
        "compiler": subprocess.check_output(["g++", "--version"], text=True).splitlines()[0],
        "libc": " ".join(platform.libc_ver()),
        "oracle_flags": "-O2 -ffp-contract=off (oracle/Makefile)",
        "scenes": {},
    }
    for name in ALL_SCENES:
        rgb, ids, ts, ct = OracleScene("scenes/%s.json" % name, W, H, DEPTH).render(threads=8)
        t_bits = ts.view(np.uint32)
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), ids=ids, t_bits=t_bits, rgb=rgb)
        manifest["scenes"][name] = {"width": W, "height": H, "depth": DEPTH, "counters": ct,
                                    "sha256": {"ids": sha(ids), "t_bits": sha(t_bits), "rgb": sha(rgb)}}
        print("%-24s rays %8d  ids %s" % (name, ct["rays"], sha(ids)[:12]))
    with open(os.path.join(out_dir, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
