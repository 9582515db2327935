"""Host side of the multi-GPU launch path, without a GPU: bench.py's own rank launcher
(raytracer_amd.dist.spawn_ranks), the file rendezvous the ranks exchange RCCL's id through, the
host-file stand-in collectives used when rehearsing, and bench.py's refusal to run with the wrong
number of ranks."""

import os
import subprocess
import sys
import textwrap
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]

CHILD = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, %r)
    import numpy as np
    from raytracer_amd import dist as D
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert int(os.environ["LOCAL_RANK"]) == rank
    rd = D.Rendezvous()
    if rank == 0:
        rd.publish("id", bytes(range(128)))
    ident = rd.wait("id", timeout=30)
    assert ident == bytes(range(128))
    class FakeRenderer:            # the two calls Comm's rehearsal path makes
        def __init__(self): self.img = np.full((4, 6, 4), float(rank + 1), np.float32)
        def read_sum(self): return self.img
        def write_sum(self, a): self.img = np.array(a)
    r = FakeRenderer()
    comm = D.Comm(r, rank, world, rdzv=rd, rehearse=True)
    s = comm.all_reduce([rank + 1.0, 10.0], "sum")
    m = comm.all_reduce([rank + 1.0, -rank], "max")
    comm.barrier()
    comm.reduce_sum_image(0)
    if %r and rank == world - 1:
        sys.exit(3)
    if rank == 0:
        print(json.dumps({"world": world, "sum": s.tolist(), "max": m.tolist(), "img": float(r.img[0, 0, 0])}))
    else:
        print("only rank 0's stdout is relayed")
""")


def _run(world, fail):
    code = CHILD % (str(ROOT), fail)
    launcher = ("import sys; sys.path.insert(0, %r); from raytracer_amd import dist as D; "
                "sys.exit(D.spawn_ranks([sys.executable, '-c', %r], %d, timeout=60))" % (str(ROOT), code, world))
    return subprocess.run([sys.executable, "-c", launcher], capture_output=True, text=True, timeout=120)


def test_spawn_ranks_rendezvous_and_rehearsal_collectives():
    import json

    p = _run(3, False)
    assert p.returncode == 0, p.stderr
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out == {"world": 3, "sum": [6.0, 30.0], "max": [3.0, 0.0], "img": 6.0}


def test_spawn_ranks_reports_a_failed_rank():
    p = _run(2, True)
    assert p.returncode == 1
    assert "ranks failed" in p.stderr


def test_bench_refuses_a_rank_count_that_does_not_match():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True,
                       timeout=120)
    assert p.returncode != 0
    assert "--gpus 4" in p.stderr and "2 rank(s)" in p.stderr


def test_c5_split_and_shards_cover_every_pixel_sample_once():
    sys.path.insert(0, str(ROOT))
    import bench
    from raytracer_amd import dist as D

    w, h, spp = 200, 136, 8
    for world in (1, 2, 3, 4, 8):
        tiles, phases = bench.c5_split(world)
        assert tiles * phases == world
        count = np.zeros((h, w), np.int64)
        for rank in range(world):
            shard, phase = rank % tiles, rank // tiles
            frames = [f for f in range(spp) if f % phases == phase]
            count += D.owner_mask(w, h, shard, tiles) * len(frames)
        assert np.all(count[: (h // 4) * 4, : (w // 8) * 8] == spp), world
