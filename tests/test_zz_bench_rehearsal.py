"""bench.py's N > 1 code walked through on one GPU: two ranks share GPU 0 and exchange over gloo
(MIRT_BENCH_REHEARSE=1); the real run uses one GPU per rank over RCCL.  Checks the contract of the JSON line."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.gpu
@pytest.mark.parametrize("width,height,spp", [(640, 360, 16), (1920, 1080, 32)])   # the second: the frame the image fixture is of
def test_two_rank_bench_prints_one_whole_job_line(width, height, spp):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MIRT_BENCH_REHEARSE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--spp", str(spp),
           "--width", str(width), "--height", str(height)]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "strong"
    assert d["config"]["workload"] == f"final_scene {width}x{height}, {spp} spp, depth 5"      # half the samples on each of two ranks
    assert d["unit"] == "Msamples/s" and d["value"] > 0 and d["higher_is_better"] is True
    assert abs(d["value"] - width * height * spp / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * d["value"]
    # the scene is served from LDS: the vector ALUs are the roof, and no fraction may exceed it
    assert d["roofline"]["bound"] == "valu" and 0 < d["roofline"]["frac"] <= 1
    # rank 0 times the CPU oracle after the timed region on N > 1 too (the other ranks wait at the barrier)
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    assert d["metric"] == ("Msamples/s on final_scene 1080p" if width == 1920 else f"Msamples/s on final_scene {width}x{height}")
    if width == 1920:
        # rank 0 checks the reduced frame of the last timed step against the oracle's (no rank renders again for it)
        assert d["golden"] is not None and d["golden"]["psnr_db"] > 30.0


@pytest.mark.gpu
def test_in_library_bench_shares_the_frame_out_inside_one_context():
    """`bench.py --gpus 2 --in-library`: one process, mirt_context_create_multi (device 0 listed twice here)."""
    env = dict(os.environ, MIRT_BENCH_REHEARSE="1")
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--in-library", "--steps", "2", "--warmup", "1", "--spp", "32",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["workload"] == "final_scene 1920x1080, 32 spp, depth 5"
    assert "inside the library" in d["config"]["sharding"]
    assert abs(d["value"] - 1920 * 1080 * 32 / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * d["value"]
    assert d["roofline"]["bound"] == "valu" and 0 < d["roofline"]["frac"] <= 1
    assert d["golden"] is not None and d["golden"]["psnr_db"] > 30.0      # the summed frame is the frame
