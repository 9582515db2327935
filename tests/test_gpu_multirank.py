"""Ranks + HIP kernels + collectives together on one GPU: two and three rank processes started by
the library's own launcher render their shards of the same image on GPU 0 and combine them into
rank 0's sum image.  RCCL refuses several ranks on one device, so the exchange itself goes through
the host-file stand-in (`Comm(rehearse=True)`); launcher, rendezvous, sharding rules, kernels and
the finalise step are the real ones.  The RCCL entry points are exercised with a one-rank
communicator in tests/test_gpu_c5.py."""

import subprocess
import sys
import textwrap
from pathlib import Path

import numpy as np
import pytest

from util import GOLDEN, load_golden_scene

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
W, H, BOUNCES, SPP = 320, 180, 4, 12

CHILD = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests")
    import numpy as np
    from raytracer_amd import dist as D
    from raytracer_amd.renderer import Renderer
    from util import load_golden_scene
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    r = Renderer(device=0, singleton=False)
    load_golden_scene(r, "gally")
    r.update_size({w}, {h}); r.bounces = {b}
    comm = D.Comm(r, rank, world, rehearse=True)
    r.clear_sum()
    if {mode!r} == "spp":
        r.render_sum(0, {spp}, frame_stride=world, frame_phase=rank)
    else:
        r.render_sum(0, {spp}, shard=rank, nshards=world)
    comm.reduce_sum_image(0)
    total = comm.all_reduce([float(r.stats()["rays"])], "sum")[0]
    if rank == 0:
        r.finalize_sum({spp})
        np.savez({out!r}, sum=r.read_sum(), mean=r.read_accumulation(), rays=total)
    comm.barrier()
    r.close()
""")


@pytest.mark.parametrize("world,mode", [(2, "spp"), (3, "tiles")])
def test_rank_processes_render_shards_on_the_gpu_and_combine(tmp_path, world, mode):
    from raytracer_amd.renderer import Renderer

    out = str(tmp_path / "combined.npz")
    code = CHILD.format(root=str(ROOT), w=W, h=H, b=BOUNCES, spp=SPP, mode=mode, out=out)
    launcher = ("import sys; sys.path.insert(0, %r); from raytracer_amd import dist as D; "
                "sys.exit(D.spawn_ranks([sys.executable, '-c', %r], %d, timeout=240))" % (str(ROOT), code, world))
    p = subprocess.run([sys.executable, "-c", launcher], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    got = np.load(out)
    # the same image in one process
    r = Renderer(device=0, singleton=False)
    try:
        load_golden_scene(r, "gally")
        r.update_size(W, H)
        r.bounces = BOUNCES
        r.reset_stats()
        r.clear_sum()
        r.render_sum(0, SPP)
        want_sum = r.read_sum()
        rays = r.stats()["rays"]
        r.finalize_sum(SPP)
        want_mean = r.read_accumulation()
    finally:
        r.close()
    assert np.all(got["sum"][..., 3] == SPP)
    assert float(got["rays"]) == float(rays)
    if mode == "tiles":   # every pixel is rendered by one rank, all samples in the same order: exact
        assert np.array_equal(got["sum"].view(np.uint32), want_sum.view(np.uint32))
        assert np.array_equal(got["mean"].view(np.uint32), want_mean.view(np.uint32))
    else:                 # samples are added in another order: fp32 reassociation only
        np.testing.assert_allclose(got["sum"], want_sum, rtol=2e-6, atol=1e-6)
        assert np.abs(got["mean"] - want_mean).max() <= 1e-5


def test_bench_rehearsal_with_eight_ranks_carries_the_diagnostics():
    """bench.py --gpus 8 end to end, its own launcher included, with the eight ranks sharing this one
    GPU (RT_BENCH_REHEARSE=1: collectives through host files, RCCL refuses several ranks on one
    device; small image and scene): the line carries what a first run on eight GPUs must show --
    the communicator's rank count, every rank's device and PCI bus id, the reduce time and its
    warm-up, stack overflows, the sustained leg with the measured shader clock."""
    import json
    import os

    env = dict(os.environ, RT_BENCH_REHEARSE="1")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1",
                        "--width", "320", "--height", "184", "--detail", "0.05", "--sustain-seconds", "0.2",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["rccl_ranks"] == 8 and out["scaling"] == "weak"
    assert "REHEARSAL" in out["data"]
    assert len(out["ranks"]) == 8 and [r["rank"] for r in out["ranks"]] == list(range(8))
    assert all(r["device"] == 0 and r["pci_bus_id"] != "unknown" for r in out["ranks"])
    assert all(1.0 < r["shader_clock_GHz"] < 3.0 for r in out["ranks"]), out["ranks"]
    assert out["stack_overflows"] == 0 and out["reduce_ms"] >= 0.0 and len(out["reduce_warmup_ms"]) == 2
    assert out["sustained"]["frames"] >= 8 * 3 and out["sustained"]["value"] > 0
    assert out["value"] > 0 and out["ms_per_step"] > 0
    # weak scaling: every rank rendered its own frames of the same image
    assert abs(out["config"]["rays_per_frame"] * 8 * 3 / (out["value"] * 1e6 * out["ms_per_step"] * 3e-3) - 1) < 0.05


def test_bench_c5_rehearsal_with_four_ranks_tile_and_sample_shards():
    """bench.py --config c5 --gpus 4 end to end (two tile shards x two sample shards, the four ranks
    sharing this GPU, collectives through host files): the strong-scaling line with its per-rank
    render times, reduce time and imbalance."""
    import json
    import os

    env = dict(os.environ, RT_BENCH_REHEARSE="1")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4", "--config", "c5", "--steps", "1", "--warmup", "1",
                        "--width", "640", "--height", "360", "--spp", "8", "--detail", "0.05"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 4 and out["scaling"] == "strong" and out["rccl_ranks"] == 4
    assert "2 tile shard(s)" in out["config"]["parallelism"] and "x 2 sample shard(s)" in out["config"]["parallelism"]
    assert len(out["per_rank_render_ms"]) == 4 and all(t > 0 for t in out["per_rank_render_ms"])
    assert out["imbalance_max_over_mean"] >= 1.0 and out["stack_overflows"] == 0 and out["value"] > 0
    # every pixel-sample of the image was rendered exactly once across the ranks: rays of a 640x360 8-spp image
    assert 640 * 360 * 8 < out["value"] * 1e6 * out["ms_per_step"] * 1e-3 < 640 * 360 * 8 * 4.0


def test_bench_under_torch_distributed_run_with_two_ranks():
    """The external launcher: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2
    --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...` (the ranks read RANK / LOCAL_RANK /
    WORLD_SIZE from its environment and find each other through a directory derived from the
    launcher's pid, start time, port and run id), the two ranks sharing this GPU with collectives
    through host files."""
    import json
    import os
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RT_BENCH_REHEARSE="1")
    env.pop("RT_RDZV_DIR", None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"),
                        "--gpus", "2", "--steps", "3", "--warmup", "1", "--width", "320", "--height", "184",
                        "--detail", "0.05", "--sustain-seconds", "0", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert [r["rank"] for r in out["ranks"]] == [0, 1] and out["value"] > 0
