"""The N>1 path on CPU: two gloo ranks shard frames (and tiles), sum-reduce their linear radiance
images to rank 0, and the result equals the single-process sum.  The per-rank renderer here is
the CPU oracle, so the test exercises the sharding rules and the collective, not the kernels."""

import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
W, H, BOUNCES, FRAMES = 96, 64, 3, 6


def _linear(img):
    c = img[..., :3].astype(np.float64)
    return np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)


def _worker(rank, world, port, mode, out_path):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    import oracle as O
    from raytracer_amd import dist as D

    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = O.OracleScene.from_dir(ROOT / "tests" / "golden" / "debug_box")
    acc = np.zeros((H, W, 4), np.float64)
    if mode == "spp":
        frames, mask = D.frames_of_rank(0, FRAMES, rank, world), np.ones((H, W), bool)
    else:
        frames, mask = list(range(FRAMES)), D.owner_mask(W, H, rank, world)
    for f in frames:
        img = sc.render(W, H, f, BOUNCES, threads=1)
        acc[..., :3] += _linear(img) * mask[..., None]
        acc[..., 3] += mask
    t = torch.from_numpy(acc)
    D.reduce_sum_to_root(t, dst=0)
    if rank == 0:
        np.save(out_path, t.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("mode", ["spp", "tiles"])
def test_two_rank_sum_equals_single_process(tmp_path, mode):
    import torch.multiprocessing as mp

    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle as O

    out = tmp_path / f"sum_{mode}.npy"
    mp.spawn(_worker, args=(2, _free_port(), mode, str(out)), nprocs=2, join=True)
    got = np.load(out)
    sc = O.OracleScene.from_dir(ROOT / "tests" / "golden" / "debug_box")
    ref = np.zeros((H, W, 4), np.float64)
    for f in range(FRAMES):
        ref[..., :3] += _linear(sc.render(W, H, f, BOUNCES, threads=1))
        ref[..., 3] += 1
    if mode == "tiles":
        ref[..., 3] = FRAMES  # every pixel is owned by exactly one rank
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)


def test_sharding_rules():
    from raytracer_amd import dist as D

    assert D.frames_of_rank(10, 7, 1, 3) == [11, 14]
    assert sorted(sum((D.frames_of_rank(0, 9, r, 4) for r in range(4)), [])) == list(range(9))
    m = [D.owner_mask(200, 130, s, 3) for s in range(3)]
    total = m[0].astype(int) + m[1] + m[2]
    assert total[:128, :200].min() == 1 and total[:128, :200].max() == 1  # a partition of the covered area
    assert total[128:, :].max() == 0  # rows beyond 4*(H/4) are never rendered
    assert D.tile_owner(0, 0, 8) == 0 and D.tile_owner(64, 0, 8) == 1 and D.tile_owner(0, 64, 8) == 2
