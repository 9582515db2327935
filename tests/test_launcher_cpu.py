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
    g = comm.gather(np.arange(30) + 100.0 * rank)      # 3 x 30 values: more than one 64-value all-reduce
    assert g.shape == (world, 30) and all(np.array_equal(g[k], np.arange(30) + 100.0 * k) for k in range(world))
    assert comm.info() == (rank, world)
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


def test_spawn_ranks_relays_the_stderr_of_the_rank_that_failed():
    code = ("import os, sys, time\n"
            "rank = int(os.environ['RANK'])\n"
            "print('diagnostic from rank', rank, file=sys.stderr, flush=True)\n"
            "if rank == 2: raise SystemExit('rank two gives up: no such device')\n"
            "time.sleep(30)\n")
    launcher = ("import sys; sys.path.insert(0, %r); from raytracer_amd import dist as D; "
                "sys.exit(D.spawn_ranks([sys.executable, '-c', %r], 4, timeout=60))" % (str(ROOT), code))
    p = subprocess.run([sys.executable, "-c", launcher], capture_output=True, text=True, timeout=120)
    assert p.returncode == 1
    assert "rank 2 failed" in p.stderr and "rank two gives up: no such device" in p.stderr
    assert "diagnostic from rank 0" in p.stderr            # rank 0's stderr is always relayed
    assert "rank 1 was still running" in p.stderr          # the ranks left waiting are named too


def test_rendezvous_dir_of_an_external_launcher_is_private_and_per_run(tmp_path, monkeypatch):
    sys.path.insert(0, str(ROOT))
    from raytracer_amd import dist as D

    monkeypatch.delenv("RT_RDZV_DIR", raising=False)
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    import tempfile

    tempfile.tempdir = None   # re-read TMPDIR
    try:
        monkeypatch.setenv("MASTER_PORT", "29999")
        monkeypatch.setenv("TORCHELASTIC_RUN_ID", "none")
        monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "0")
        a = D.rendezvous_dir()
        assert a.parent == tmp_path and (os.stat(a).st_mode & 0o777) == 0o700
        assert D.rendezvous_dir() == a                      # every rank of the run derives the same one
        monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "1")
        b = D.rendezvous_dir()
        assert b != a                                       # a restarted run never reads the earlier run's blobs
        os.chmod(b, 0o777)                                  # a directory others could have written is refused
        try:
            D.rendezvous_dir()
            raise AssertionError("a world-writable rendezvous directory was accepted")
        except RuntimeError as e:
            assert "private" in str(e)
    finally:
        tempfile.tempdir = None


def test_ranks_of_an_external_launcher_find_each_other_and_leave_nothing_behind(tmp_path):
    """Three ranks started the way an external launcher starts them (plain children of one process,
    RANK / WORLD_SIZE / MASTER_PORT in the environment, no RT_RDZV_DIR): they meet in the derived
    directory, rank 1 dawdles before its last barrier, and Comm.finish() removes the directory only
    after every rank has read its last blob."""
    code = textwrap.dedent("""
        import os, sys, time
        sys.path.insert(0, %r)
        from raytracer_amd import dist as D
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        comm = D.Comm(None, rank, world, rehearse=True)
        print(str(comm.rdzv.dir), flush=True)
        assert comm.all_reduce([float(rank)], "sum")[0] == 3.0
        assert comm.gather([rank * 2.0])[:, 0].tolist() == [0.0, 2.0, 4.0]
        if rank == 1:
            time.sleep(0.5)
        comm.finish()
    """ % str(ROOT))
    env = dict(os.environ, WORLD_SIZE="3", MASTER_PORT="29533", TMPDIR=str(tmp_path), TORCHELASTIC_RUN_ID="t1",
               TORCHELASTIC_RESTART_COUNT="0")
    env.pop("RT_RDZV_DIR", None)
    procs = [subprocess.Popen([sys.executable, "-c", code], env=dict(env, RANK=str(k), LOCAL_RANK=str(k)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for k in range(3)]
    outs = [p.communicate(timeout=120) for p in procs]
    assert [p.returncode for p in procs] == [0, 0, 0], [o[1][-800:] for o in outs]
    dirs = {o[0].strip() for o in outs}
    assert len(dirs) == 1 and Path(dirs.pop()).parent == tmp_path
    assert list(tmp_path.iterdir()) == []


def test_c4_line_carries_the_multi_gpu_diagnostics():
    """The JSON line of an 8-rank run, assembled from stand-in measurements (the GPU rehearsal of the
    same path is tests/test_gpu_multirank.py)."""
    import argparse
    import json

    sys.path.insert(0, str(ROOT))
    import bench

    world = 8
    args = argparse.Namespace(width=1920, height=1080, bounces=4, steps=20, warmup=3)
    per = [[k, bench.pci_to_number("0000:%02x:00.0" % (5 + 16 * k)), 36.0 + k, 0.9, 2.1, 0, k] for k in range(world)]
    m = {"elapsed": 0.040, "rays": 8.2e6 * 20 * world, "reduce_ms_max": 0.9, "reduce_warmup_ms": [55.0, 0.9],
         "per_rank": per, "rccl_ranks": world, "ref_frame": None,
         "sustained": {"value": 1.0, "ms_per_step": 1.8, "frames": 8 * 1100, "shader_clock_GHz": {}},
         "fast": None, "kernel_ms": 1.94, "rays_per_launch": 8.2e6, "per_ray": (22.0, 3.7, 0.13, 0.99, 3.96),
         "variant": "unified", "max_stack": 17, "build_s": 0.3}
    out = json.loads(json.dumps(bench.assemble_c4(args, world, "sponza_synth(detail=1.0)", 265034, 132402, False, m)))
    assert out["n_gpus"] == 8 and out["rccl_ranks"] == 8 and out["scaling"] == "weak"
    assert abs(out["value"] - 8.2e6 * 20 * 8 / 0.040 / 1e6) < 1e-6
    assert out["reduce_ms"] == 0.9 and out["reduce_warmup_ms"] == [55.0, 0.9] and out["stack_overflows"] == 0
    assert [r["device"] for r in out["ranks"]] == list(range(8))
    assert out["ranks"][3]["pci_bus_id"] == "0000:35:00.0" and out["ranks"][3]["comm_rank"] == 3
    assert "sustained" in out and "fast_mode" not in out and out["reference_frame"] is None
    assert "same step at every n_gpus" in out["config"]["step"]
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    for key in ("hbm_counter_frac", "td_busy", "vmem_frac", "l2_hit", "binding_resource"):
        assert key in rf


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
