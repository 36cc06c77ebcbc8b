"""The host side of the library — the scenefile loader (src/utils/scenefilereader.cpp + sceneparser.cpp of the reference are what
rm_scene.cpp replaces) and the image readers (QImage in realtimerender.cpp:267-303: PNG in rm_host.cpp, rm_jpeg.cpp, rm_gif.cpp) —
under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU (the GPU build cannot run under a sanitizer on this pool): every
scenefile and image of tests/golden/scenes, then seeded random mutations of them.  A mutated input may be rejected; it must not
touch memory it does not own, overflow, or leak."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "raymarcher_amd", "csrc")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    out = tmp_path_factory.mktemp("host_fuzz") / "host_fuzz"
    cmd = ["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-o", str(out), os.path.join(ROOT, "tests", "host_fuzz", "host_fuzz.cpp")]
    cmd += [os.path.join(CSRC, f) for f in ("rm_host.cpp", "rm_scene.cpp", "rm_jpeg.cpp", "rm_gif.cpp")] + ["-lz"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and ("cannot find -lasan" in r.stderr or "cannot find -lubsan" in r.stderr):
        pytest.skip("sanitizer runtimes not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    return str(out)


@pytest.mark.parametrize("seed", [1, 2])
def test_loader_and_image_readers_under_asan_ubsan(harness, tmp_path, seed):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([harness, os.path.join(ROOT, "tests", "golden", "scenes"), str(tmp_path), "1500", str(seed)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "host_fuzz ok: 51 of 52 scenefiles loaded" in r.stdout, r.stdout  # unit_terrain.json is invalid in the reference itself
