"""Scene lists shared by the CPU and GPU tests."""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))

with open(os.path.join(HERE, "golden", "reference_counters.json")) as f:
    REFERENCE = json.load(f)

# Scenes with reference counters (256x256, depth 3).
COUNTED_SCENES = list(REFERENCE["scenes"].keys())
# Runnable scenes without reference counters (same code paths).
OTHER_SCENES = ["meshTeapot", "meshSphere", "meshPlane", "meshTwoBunnies"]
ALL_SCENES = COUNTED_SCENES + OTHER_SCENES
COUNTER_KEYS = ("rays", "shadow_rays", "node_tests", "leaf_visits", "tri_tests", "sphere_tests")
