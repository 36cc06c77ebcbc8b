"""bench.py's own N > 1 branch (RCCL process group, FramePipeline with three frames in flight, rm_render_tiles, the pipelined
gather, rm_deinterleave / rm_deinterleave_rgba8, wall-time roofline, the MAX all-reduce of the times) executed on the one-GPU
box: `--force-dist` runs it with a one-rank group, as a FRESH child process exactly as the driver would start a rank.  The
first execution of that code on an 8-GPU node must not also be its first execution anywhere (missing #3)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_bench(*flags):
    for attempt in range(3):  # the port found free can be taken before the child binds it (seen once: EADDRINUSE): another port then
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--no-cpu-baseline", *flags],
                           cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        if p.returncode == 0 or "EADDRINUSE" not in p.stderr:
            break
    assert p.returncode == 0, f"bench.py failed ({p.returncode}):\n{p.stdout[-2000:]}\n{p.stderr[-4000:]}"
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"rank 0 must print ONE JSON line, got {len(lines)}"
    return json.loads(lines[0])


@pytest.mark.parametrize("flags,px", [
    (("--config", "c3", "--steps", "3", "--warmup", "1"), 3840 * 2160),
    (("--config", "c3", "--steps", "3", "--warmup", "1", "--gather", "rgba8"), 3840 * 2160),
    (("--config", "c5", "--steps", "2", "--warmup", "1"), 7680 * 4320),
    (("--config", "c3", "--steps", "3", "--warmup", "1", "--gather-root", "rotate"), 3840 * 2160),
], ids=["c3_float4", "c3_rgba8", "c5_float4", "c3_rotating_root"])
def test_bench_distributed_branch_in_a_child_process(flags, px):
    line = run_bench(*flags)
    assert line["n_gpus"] == 1 and line["steps"] == int(flags[3]) and line["unit"] == "Mpixels/s" and line["scaling"] == "strong"
    assert line["value"] > 0 and abs(line["value"] - px / (line["ms_per_step"] * 1e3)) <= 0.01 * line["value"]
    roof = line["roofline"]
    assert roof["time_base"].startswith("wall time per frame")  # overlapping launches: event spans are not kernel time
    assert 0.0 < roof["frac"] < 1.0
    # the frame the pipeline delivered on rank 0 (gathered + de-interleaved; as RGBA8 when that is what travelled) is, bit for
    # bit, the frame a direct whole-frame render produces
    assert roof["algorithmic"]["counted_frame_identical_to_timed_frame"] is True
    assert "three frames in flight" in line["config"]["rows"]
    assert ("RGBA8" in line["config"]["rows"]) == ("rgba8" in flags)
    assert "cpu_baseline" not in line
    if "rotate" in flags:
        assert "on its root" in line["config"]["rows"] and "rotating_root" not in line["variants"]
    else:  # the rotating-root pipeline is timed beside the headline and delivers the same frame
        v = line["variants"]["rotating_root"]
        assert v["value"] > 0 and v["frame_identical_to_headline"] is True
