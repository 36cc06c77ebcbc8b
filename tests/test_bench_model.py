"""bench.py's host side without a GPU: every BASELINE configuration builds from the reference's scenefiles, and the
algorithmic-work model is the arithmetic DESIGN.md §6 states."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from raymarcher_amd import abi  # noqa: E402


@pytest.mark.parametrize("name,size,objects,lights", [("c1", (256, 256), 2, 3), ("c2", (1920, 1080), 5, 3), ("c3", (3840, 2160), 1, 3),
                                                      ("c4", (3840, 2160), 1, 1), ("c5", (7680, 4320), 1, 3)])
def test_configs_build_from_the_reference_scenefiles(name, size, objects, lights):
    t, s, W, H, d = bench.build_config(name)
    assert (W, H) == size and t.num_objects == objects and t.num_lights == lights
    assert d["baseline_config"].startswith("configs[") and "Mpixels/s" in d["metric"]
    if name == "c1":
        assert s.maxSteps == 64 and len(t.textures) == 1  # the floor's blackmarble.png through the product's PNG reader
    if name == "c2":
        assert s.enableSoftShadow and s.enableAmbientOcclusion
    if name == "c3":
        assert s.fractalIters == 12 and t.objects[0].type == abi.RM_MANDELBULB and d["metric"] == "Mpixels/s at 3840x2160 Mandelbulb, 256 march steps"
    if name == "c4":
        assert s.features & abi.RM_FEAT_TERRAIN and s.features & abi.RM_FEAT_CLOUD and t.camera.initialFar == 2000.0
    if name == "c5":
        assert s.mengerLevels == 5 and s.numReflection == 2 and s.enableReflection and t.objects[0].type == abi.RM_MENGERSPONGE


def test_flop_model_arithmetic():
    # the headline frame's counters (BENCH_r02.json): SURVEY §8(d)'s 79 / 41 / 1800 give 68.98 GFLOP
    t, s, W, H, _ = bench.build_config("c3")
    c = abi.RmCounters(330035979, 640027016, 2717394, 2717394, 0, 0)
    flop, parts, model = bench.flop_model(t, s, c)
    assert model["flop_per_evaluation"] == 41 and model["flop_per_iteration"] == 79 and model["flop_per_shaded_point"] == 1800
    assert flop == 640027016 * 79 + 330035979 * 41 + 2717394 * 1800 and abs(flop / 1e9 - 68.98) < 0.01
    # table-walk classes: per evaluation 10 + per object 21 + its SDF.  Menger at iTime = 0: box + 42 per level (the launch-uniform
    # prologue and the identity rotation mix are priced only in the shader-as-written side figure: 40 + 27 per level more)
    t, s, W, H, _ = bench.build_config("c5")
    assert bench.flop_model(t, s, abi.RmCounters(1, 0, 0, 0, 0, 0))[0] == 10 + 21 + 19 + 42 * 5
    assert bench.flop_model(t, s, abi.RmCounters(1, 0, 0, 0, 0, 0), as_written=True)[0] == 10 + 21 + (19 + 40) + 69 * 5
    t.globals_.iTime = 2.5  # an animated sponge does execute the mix
    assert bench.flop_model(t, s, abi.RmCounters(1, 0, 0, 0, 0, 0))[0] == 10 + 21 + 19 + 69 * 5
    t, s, W, H, _ = bench.build_config("c2")  # cylinder, cone, sphere, two cubes (any order)
    assert bench.flop_model(t, s, abi.RmCounters(1, 0, 0, 0, 0, 0))[0] == 10 + 5 * 21 + 17 + 28 + 7 + 19 + 19
    assert bench.flop_model(t, s, abi.RmCounters(0, 0, 0, 1, 0, 0))[0] == 50 + 3 * 50 + 4 * 400
    t, s, W, H, _ = bench.build_config("c4")
    assert bench.flop_model(t, s, abi.RmCounters(0, 0, 0, 0, 1, 1))[0] == bench.FLOP_PER_FBM9 + bench.FLOP_PER_FBMD8
