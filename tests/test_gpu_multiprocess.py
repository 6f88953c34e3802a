"""The multi-process tools (one process per GPU under torch.distributed.run) rehearsed with two ranks sharing the one
GPU of the test box and gloo between them (RH_REHEARSE_ON_ONE_GPU=1): rendezvous, strip assignment, the all-gather of
the intermediate target, the gather of results and of encoded frames are the real code; only RCCL itself is replaced."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def launch(world, script, args, port):
    # (one stream per process: two processes on one GPU take turns on it, see bench.py)
    env = dict(os.environ, RH_REHEARSE_ON_ONE_GPU="1", RH_EARLY_BINNING="0")
    if world == 1:
        cmd = [sys.executable, os.path.join(ROOT, "tools", script)] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "tools", script)] + args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout


def test_two_rank_strips_equal_one_rank(tmp_path):
    from PIL import Image
    one, two = str(tmp_path / "one.png"), str(tmp_path / "two.png")
    args = ["--size", "512", "--quads", "3000", "--side", "24"]
    launch(1, "render_strips.py", args + ["--out", one], 0)
    launch(2, "render_strips.py", args + ["--out", two], 29611)
    a, b = np.array(Image.open(one)), np.array(Image.open(two))
    assert a.shape == (512, 512, 4) and np.array_equal(a, b)
    assert len(np.unique(a)) > 16                         # a real image, not a cleared target


def test_two_rank_frame_parallel_recording_equals_one_rank(tmp_path):
    one, two = str(tmp_path / "one.pngstream"), str(tmp_path / "two.pngstream")
    args = ["--frames", "12", "--width", "320", "--height", "180"]
    launch(1, "record_frame_parallel.py", args + ["--out", one], 0)
    launch(2, "record_frame_parallel.py", args + ["--out", two], 29612)
    a, b = open(one, "rb").read(), open(two, "rb").read()
    assert a.count(b"IEND") == 12 and a == b              # the same 12 PNGs in frame order


def test_bench_with_two_ranks_prints_one_json_line(tmp_path):
    """bench.py under torch.distributed.run with two ranks (rehearsal mode): rank 0 prints exactly one JSON line on
    stdout with the whole-job value; nothing else reaches stdout."""
    import json
    env = dict(os.environ, RH_REHEARSE_ON_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29613",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--quads", "100000", "--no-cpu-baseline",
           "--capture-frames", "8", "--capture-streams", "4"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["roofline"]["frac"] > 0 and d["capture"]["value"] > 0 and d["capture"]["frames_per_gpu"] == 8


def test_bench_started_plainly_with_gpus_2_starts_two_ranks_itself(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it: the process starts the two ranks itself (before any GPU
    call of its own) and relays rank 0's line, which names the ranks that really ran."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["RH_REHEARSE_ON_ONE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--quads", "50000",
           "--no-cpu-baseline", "--capture-frames", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and d["devices"] == [0, 0] and d["backend"].startswith("gloo")


@pytest.mark.parametrize("world", [1, 2])
def test_bench_strips_leg_prints_a_line(world):
    """bench.py --workload c3 (the two-pass render-to-texture configuration split into strips of tile rows), at a small
    size: one rank, and two ranks in rehearsal mode started by bench.py itself."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["RH_REHEARSE_ON_ONE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c3", "--gpus", str(world), "--width", "1024", "--quads", "20000",
           "--side", "32", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["ranks"] == world and d["scaling"] == "strong" and d["value"] > 0 and d["roofline"]["frac"] > 0
    assert d["exchange"]["bytes_received_per_rank_per_step"] == (0 if world == 1 else 1024 * 1024 * 4)


@pytest.mark.parametrize("before", ["torch", "c1"])
def test_recording_run_in_a_process_that_has_used_the_gpu_before(before):
    """A recording run (renderer + compressor: five streams) after PyTorch has run a kernel through the HIP runtime the
    two then share, or after a render-only run of the same process.  With passes binned early on a third stream beside
    the compressor's streams this never got past its third frame; the harness ends a run that stalls and says where."""
    env = dict(os.environ, RH_WATCHDOG="120")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "exp_capture_env.py"), before, "256"],
                         capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["mode"] == before and line["fps"] > 0
