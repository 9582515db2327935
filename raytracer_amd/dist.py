"""Multi-GPU sharding of the render path: one process per GPU, scene replicated.

The reference renders every pixel of every frame on one device; there is nothing to mirror
here.  Work is split two ways, both without any exchange during rendering because every
pixel-sample is independent (its RNG is keyed by pixel and frame index only):

* samples per pixel: rank r renders the frames f with (f - first_frame) % world == r;
* image tiles: 64x64-pixel macro tiles are dealt round-robin in Morton order
  (``tile_owner``), the rule the kernels apply when ``rt_render_sum`` is given nshards > 1.

Each rank adds linear radiance into its own fp32 RGBA sum image; the images are combined once
per run with a sum-reduce to rank 0, after which rank 0 divides by the sample count and
sRGB-encodes.  On GPUs the reduce is RCCL over xGMI called through the C ABI (``rt_comm_*`` in
include/rt_mi355x.h, wrapped by ``Comm`` below); no framework is involved.  The only thing ranks
must exchange beforehand is RCCL's 128-byte id, which goes through a directory all ranks of the
node can see (``Rendezvous``).  ``spawn_ranks`` starts one process per GPU when no launcher did.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np


def frames_of_rank(first_frame: int, nframes: int, rank: int, world: int) -> list[int]:
    """Frame indices rank `rank` renders under samples-per-pixel sharding."""
    return [first_frame + i for i in range(nframes) if i % world == rank]


def _part1by1(v: np.ndarray) -> np.ndarray:
    v = v.astype(np.uint32) & 0xFFFF
    v = (v | (v << 8)) & 0x00FF00FF
    v = (v | (v << 4)) & 0x0F0F0F0F
    v = (v | (v << 2)) & 0x33333333
    v = (v | (v << 1)) & 0x55555555
    return v


def tile_owner(x, y, nshards: int) -> np.ndarray:
    """Shard that owns pixel (x, y): Morton index of its 64x64 macro tile modulo nshards."""
    x = np.asarray(x, dtype=np.uint32) >> 6
    y = np.asarray(y, dtype=np.uint32) >> 6
    return ((_part1by1(x) | (_part1by1(y) << 1)) % np.uint32(nshards)).astype(np.uint32)


def owner_mask(width: int, height: int, shard: int, nshards: int) -> np.ndarray:
    """(H, W) bool mask of the pixels shard `shard` renders (only the area the dispatch covers)."""
    ww, hh = (width // 8) * 8, (height // 4) * 4
    yy, xx = np.mgrid[0:height, 0:width]
    return (tile_owner(xx, yy, nshards) == shard) & (xx < ww) & (yy < hh)


def reduce_sum_to_root(sum_tensor, dst: int = 0):
    """One collective per run: element-wise sum of every rank's image into rank `dst`."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(sum_tensor, dst=dst, op=dist.ReduceOp.SUM)
    return sum_tensor


def finalize_mean_srgb(sum_rgba: np.ndarray, spp: int) -> np.ndarray:
    """Host mirror of rt_finalize_sum for tests: srgb(sum / spp) with plain libm pow.

    The device version uses the reference's own pow polynomial; this one is only used to compare
    sharded against unsharded sums, where both sides go through the same function."""
    lin = np.asarray(sum_rgba, dtype=np.float32)[..., :3] / np.float32(spp)
    out = np.where(lin <= 0.0031308, 12.92 * lin, 1.055 * np.power(np.maximum(lin, 0), 1 / 2.4) - 0.055)
    return np.concatenate([out, np.ones(out.shape[:-1] + (1,), out.dtype)], axis=-1).astype(np.float32)


# ------------------------------------------------------------------------------------------------
# launching and rendezvous (host side only: nothing here touches a GPU)
# ------------------------------------------------------------------------------------------------
def _tail(path: Path, nbytes: int = 4000) -> str:
    try:
        with open(path, "rb") as f:
            f.seek(0, os.SEEK_END)
            size = f.tell()
            f.seek(max(0, size - nbytes))
            return f.read().decode(errors="replace")
    except OSError:
        return ""


def spawn_ranks(argv: list[str], world: int, extra_env: dict | None = None, timeout: float | None = None) -> int:
    """Starts `world` fresh processes running `argv`, one per GPU, with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT / RT_RDZV_DIR set, relays rank 0's stdout, and returns 0
    only if every rank exited with 0.  Every rank's stderr goes to a file of its own in the
    rendezvous directory; rank 0's is relayed in full, and when ranks fail (or the run times out)
    the tail of each failing rank's -- and of the ranks still waiting for it -- follows, so that a
    failure on a machine nobody can log into leaves its reason in the launcher's output.  The caller
    must not have initialised the GPU."""
    rdzv = tempfile.mkdtemp(prefix="rt_mi355x_rdzv_")  # mode 0700, owned by this user
    procs, errs = [], []
    out_path = Path(rdzv) / "rank0.stdout"
    err_paths = [Path(rdzv) / f"rank{r}.stderr" for r in range(world)]
    try:
        with open(out_path, "wb") as out0:
            for rank in range(world):
                env = dict(os.environ)
                env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world),
                            "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": env.get("MASTER_PORT", "29511"),
                            "RT_RDZV_DIR": rdzv,
                            "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
                env.update(extra_env or {})
                errs.append(open(err_paths[rank], "wb"))
                procs.append(subprocess.Popen(argv, env=env, stdout=out0 if rank == 0 else subprocess.DEVNULL,
                                              stderr=errs[-1]))
            # a rank that dies would leave the others waiting in a collective for ever: watch them all
            t0, failed = time.time(), []
            while True:
                codes = [p.poll() for p in procs]
                failed = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
                if failed or all(c == 0 for c in codes):
                    break
                if timeout is not None and time.time() - t0 > timeout:
                    failed = [("timeout", timeout)]
                    break
                time.sleep(0.02)
        still_running = [r for r, p in enumerate(procs) if p.poll() is None]
        for p in procs:
            if p.poll() is None:
                p.kill()  # exactly the processes started here
                p.wait()
        for e in errs:
            e.close()
        sys.stdout.write(out_path.read_text(errors="replace"))
        sys.stdout.flush()
        sys.stderr.write(err_paths[0].read_text(errors="replace"))
        if failed:
            print(f"spawn_ranks: ranks failed out of {world} (rank, exit code): {failed}", file=sys.stderr)
            bad = [r for r, _ in failed if isinstance(r, int)]
            for r in sorted(set(bad + still_running)):
                if r == 0:
                    continue  # relayed in full above
                what = "failed" if r in bad else "was still running (killed)"
                print(f"---- rank {r} {what}; tail of its stderr ----\n{_tail(err_paths[r])}", file=sys.stderr)
            return 1
        return 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
        for e in errs:
            if not e.closed:
                e.close()
        for f in list(Path(rdzv).glob("*")) + list(Path(rdzv).glob(".*")):
            try:
                f.unlink()
            except OSError:
                pass
        try:
            os.rmdir(rdzv)
        except OSError:
            pass


def rendezvous_dir() -> Path:
    """Directory shared by the ranks of this run: RT_RDZV_DIR when spawn_ranks launched them (a
    fresh mkdtemp directory), otherwise derived from what all ranks of one external launcher have in
    common: the launcher's pid and start time, MASTER_PORT, and the launcher's run id / restart
    count when it exports them (torch.distributed.run does), so that a restarted or later run never
    reads an earlier run's blobs.  The directory is created with mode 0700 and must belong to this
    user: another local user cannot plant an id in it."""
    d = os.environ.get("RT_RDZV_DIR")
    if d:
        return Path(d)
    ppid = os.getppid()
    try:
        start = Path(f"/proc/{ppid}/stat").read_text().rsplit(")", 1)[1].split()[19]
    except OSError:
        start = "0"
    run = "_".join(os.environ.get(k, "") for k in ("TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT"))
    run = "".join(ch if ch.isalnum() or ch in "_-" else "-" for ch in run)[:64]
    path = Path(tempfile.gettempdir()) / f"rt_mi355x_rdzv_{os.getuid()}_{ppid}_{start}_{os.environ.get('MASTER_PORT', '0')}_{run}"
    try:
        path.mkdir(mode=0o700)
    except FileExistsError:
        pass
    st = os.lstat(path)
    import stat as _stat

    if not _stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise RuntimeError(f"rendezvous: {path} exists but is not a private directory of this user; "
                           "set RT_RDZV_DIR to a directory only this run's ranks can write")
    return path


class Rendezvous:
    """Write-once named blobs in a directory every rank can see."""

    def __init__(self, directory: Path | None = None) -> None:
        self.dir = Path(directory) if directory is not None else rendezvous_dir()

    def publish(self, name: str, data: bytes) -> None:
        tmp = self.dir / f".{name}.{os.getpid()}.tmp"
        tmp.write_bytes(data)
        os.replace(tmp, self.dir / name)  # atomic: a reader sees all of it or none

    def wait(self, name: str, timeout: float = 120.0) -> bytes:
        path, t0 = self.dir / name, time.time()
        while not path.exists():
            if time.time() - t0 > timeout:
                raise TimeoutError(f"rendezvous: {path} did not appear within {timeout:.0f} s "
                                   "(did every rank start?)")
            time.sleep(0.005)
        return path.read_bytes()


class Comm:
    """The run's collectives for one rank.  `rehearse=True` (several ranks sharing one GPU, where
    RCCL refuses to form a communicator) moves the same data through host files instead; it exists
    to rehearse the control flow and is never a measurement."""

    def __init__(self, renderer, rank: int, world: int, rdzv: Rendezvous | None = None, rehearse: bool = False,
                 tag: str = "run") -> None:
        from . import _lib

        self.r, self.rank, self.world, self.rehearse, self.tag = renderer, rank, world, rehearse, tag
        self.rdzv = rdzv if rdzv is not None else Rendezvous()
        self._seq = 0
        self._L = _lib.load()
        if world == 1 or rehearse:
            return
        if rank == 0:
            buf = C.create_string_buffer(_lib.RT_COMM_ID_BYTES)
            _lib.check(self._L.rt_comm_unique_id(buf))
            self.rdzv.publish(f"{tag}.rccl_id", buf.raw)
            ident = buf.raw
        else:
            ident = self.rdzv.wait(f"{tag}.rccl_id")
        _lib.check(self._L.rt_comm_init(renderer._ctx, rank, world, ident))

    # -- file-based stand-ins used when rehearsing ------------------------------------------------
    def _exchange(self, payload: np.ndarray) -> list[np.ndarray]:
        self._seq += 1
        self.rdzv.publish(f"{self.tag}.x{self._seq}.r{self.rank}", payload.tobytes())
        return [np.frombuffer(self.rdzv.wait(f"{self.tag}.x{self._seq}.r{k}"), dtype=payload.dtype).reshape(payload.shape)
                for k in range(self.world)]

    def all_reduce(self, values, op: str = "sum") -> np.ndarray:
        """Element-wise sum or max of a few float64 values over all ranks."""
        from . import _lib

        v = np.ascontiguousarray(values, dtype=np.float64).copy()
        if self.world == 1:
            return v
        if self.rehearse:
            parts = self._exchange(v)
            return np.sum(parts, axis=0) if op == "sum" else np.max(parts, axis=0)
        _lib.check(self._L.rt_comm_all_reduce_f64(self.r._ctx, v.ctypes.data_as(C.POINTER(C.c_double)), v.size,
                                                  _lib.RT_COMM_SUM if op == "sum" else _lib.RT_COMM_MAX))
        return v

    def barrier(self) -> None:
        self.all_reduce([1.0])

    def finish(self) -> None:
        """Last call of a run, on every rank: a barrier, and under an external launcher (no
        RT_RDZV_DIR, so the directory is one this run derived and created) the rendezvous directory
        is removed -- by rank 0, once every rank has said that it has read its last blob."""
        self.barrier()
        if self.world == 1 or os.environ.get("RT_RDZV_DIR"):
            return   # spawn_ranks removes the directory it made
        self.rdzv.publish(f"{self.tag}.done.r{self.rank}", b"1")
        if self.rank != 0:
            return
        for k in range(self.world):
            self.rdzv.wait(f"{self.tag}.done.r{k}")
        for f in list(self.rdzv.dir.glob("*")) + list(self.rdzv.dir.glob(".*")):
            try:
                f.unlink()
            except OSError:
                pass
        try:
            self.rdzv.dir.rmdir()
        except OSError:
            pass

    def gather(self, values) -> np.ndarray:
        """(world, k) array of every rank's k float64 values, on every rank."""
        v = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)
        slots = np.zeros((self.world, v.size))
        slots[self.rank] = v
        flat, out = slots.reshape(-1), []
        for i in range(0, flat.size, 64):   # one scalar all-reduce carries at most 64 values
            out.append(self.all_reduce(flat[i:i + 64], "sum"))
        return np.concatenate(out).reshape(self.world, v.size)

    def info(self) -> tuple[int, int]:
        """(rank, number of ranks) as the communicator itself reports them: RCCL's own view when it
        is in use, the constructor's arguments otherwise (one rank, or a rehearsal)."""
        from . import _lib

        if self.world == 1 or self.rehearse:
            return self.rank, self.world
        a, b = C.c_int(), C.c_int()
        _lib.check(self._L.rt_comm_info(self.r._ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    def reduce_sum_image(self, root: int = 0) -> None:
        """Adds every rank's own linear sum image into rank `root`'s (one collective per run)."""
        from . import _lib

        if self.world == 1:
            return
        if self.rehearse:
            img = self.r.read_sum()
            parts = self._exchange(img)
            if self.rank == root:
                total = parts[0].copy()
                for p_ in parts[1:]:
                    total = total + p_
                self.r.write_sum(total)
            return
        _lib.check(self._L.rt_comm_reduce_sum(self.r._ctx, None, root))
