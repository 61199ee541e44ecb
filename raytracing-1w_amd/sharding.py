"""Multi-GPU partitioning of the pixel/sample loop (one process per GPU).

The reference parallelises src/main.rs:957-963 with rayon over rows x columns in one
process; pixels are independent (own RNG stream, main.rs:964) and, with the
counter-based generator, so are samples.  So a node's GPUs split the job with NO
data-path collective: every rank renders its share into its own framebuffer and the
host gathers (north star: "host gather only; no RCCL").

Two partitions, both bit-exact against the single-GPU render:
  - rows:    16-row strips dealt round-robin (Cornell cost varies by region, so
             interleaving balances load); stitched by copying strips.
  - samples: rank r renders samples [r*spp_per_rank, (r+1)*spp_per_rank) of every
             pixel as raw sums (RT1W_OUT_SUM); the host adds the per-rank sums in
             rank order and then applies into_sampled.  Equals a single render
             whose chunk size is spp_per_rank (same summation order).
`render_fn(tile, sample_offset, spp, out_sum) -> ndarray[tile_h, tile_w, 3]` abstracts the
renderer (Context.render on the GPU; the CPU build of the core in the gloo tests).
"""
import numpy as np

STRIP_ROWS = 16


def row_strips(height, world_size, rank, strip_rows=STRIP_ROWS):
    """[(y0, rows), ...] owned by `rank`: strip k goes to rank k % world_size."""
    out = []
    k = 0
    y = 0
    while y < height:
        rows = min(strip_rows, height - y)
        if k % world_size == rank:
            out.append((y, rows))
        y += rows
        k += 1
    return out


def interleaved_tile(height, world_size, rank, strip_rows=STRIP_ROWS):
    """The same partition as row_strips, as ONE row-interleaved tile for rt1w_render_params:
    (y0, tile_h, strip_rows, strip_period) -- tile row r is image row y0 + r//strip_rows*strip_period + r%strip_rows.
    tile_h == 0: this rank owns nothing (more ranks than strips)."""
    rows = sum(n for _, n in row_strips(height, world_size, rank, strip_rows))
    return rank * strip_rows, rows, strip_rows, world_size * strip_rows


class SharedFrame:
    """One whole-image f64 frame [height, width, 3] in POSIX shared memory (/dev/shm), mapped by every rank of a node and
    pinned for the GPU (rt1w_host_register): each rank's device->host copy writes its strips straight to their places
    (RT1W_OUT_FRAME) -- that copy IS the host gather of the image-tiled job (north star: "host gather only; no RCCL");
    nothing is stitched afterwards.  Rank 0 creates, the others attach after a barrier; rank 0 unlinks."""

    def __init__(self, name, width, height, create, rt=None):
        import atexit
        import mmap
        import os
        self.path = os.path.join("/dev/shm", name)
        self.nbytes = width * height * 3 * 8
        self.created = create
        self.pin_error = None
        nofollow = getattr(os, "O_NOFOLLOW", 0)
        if create:
            # never follow or truncate something planted under this name: a stale regular file of OUR OWN (a rank that died) is
            # removed first, then the segment is created exclusively
            try:
                st = os.lstat(self.path)
                import stat
                if stat.S_ISREG(st.st_mode) and st.st_uid == os.getuid():
                    os.unlink(self.path)
            except FileNotFoundError:
                pass
            fd = os.open(self.path, os.O_RDWR | os.O_CREAT | os.O_EXCL | nofollow, 0o600)
            atexit.register(self._unlink)          # the creator removes the segment even when it dies on an assert
        else:
            fd = os.open(self.path, os.O_RDWR | nofollow)
        try:
            if create:
                os.ftruncate(fd, self.nbytes)
            elif os.fstat(fd).st_size != self.nbytes:
                raise ValueError(f"{self.path}: {os.fstat(fd).st_size} bytes, expected {self.nbytes}")
            self._mm = mmap.mmap(fd, self.nbytes)
        finally:
            os.close(fd)
        self.array = np.frombuffer(self._mm, dtype=np.float64).reshape(height, width, 3)
        self._rt = rt
        self._pinned = False
        if rt is not None and hasattr(rt, "host_register"):
            try:   # pinning is an optimisation (full-rate, asynchronous device->host copies); a pageable frame works too
                rt.host_register(self.array)
                self._pinned = True
            except Exception as e:
                self._pinned = False
                self.pin_error = str(e)            # reported by bench.py (config.host_frame_pin_error)

    def _unlink(self):
        import os
        if self.created:
            try:
                os.unlink(self.path)
            except OSError:
                pass
            self.created = False

    def close(self):
        if self._pinned:
            try:
                self._rt.host_unregister(self.array)
            except Exception:
                pass
            self._pinned = False
        self.array = None
        try:
            self._mm.close()
        except BufferError:
            pass
        self._unlink()


def sample_range(spp_total, world_size, rank):
    """(offset, count) of the contiguous sample range of `rank`; requires divisibility so that
    the summation order equals a single render with chunk = spp_total / world_size."""
    if spp_total % world_size:
        raise ValueError("spp_total must be divisible by the number of ranks")
    n = spp_total // world_size
    return rank * n, n


def render_rows(render_fn, width, height, spp, world_size, rank):
    """This rank's strips, as (list_of_strips, packed ndarray[rows_owned, width, 3])."""
    strips = row_strips(height, world_size, rank)
    parts = [render_fn((0, y0, width, rows), 0, spp, False) for (y0, rows) in strips]
    packed = np.concatenate(parts, axis=0) if parts else np.zeros((0, width, 3))
    return strips, packed


def stitch_rows(width, height, world_size, packed_by_rank):
    """Inverse of render_rows over all ranks -> full image [height, width, 3]."""
    img = np.empty((height, width, 3), dtype=np.float64)
    for r in range(world_size):
        pos = 0
        for (y0, rows) in row_strips(height, world_size, r):
            img[y0:y0 + rows] = packed_by_rank[r][pos:pos + rows]
            pos += rows
    return img


def combine_sample_sums(sums_by_rank):
    """Pixel sums of the whole job from per-rank sums, added in rank order starting from 0.0
    (what the resolve kernel does with chunk partials)."""
    total = np.zeros_like(sums_by_rank[0])
    for s in sums_by_rank:
        total = total + s
    return total


def gather_to_rank0(local, group=None):
    """Host gather of equally- or unequally-sized float64 arrays with torch.distributed
    (gloo on CPU tensors).  Returns list of ndarrays on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    objs = [None] * world if rank == 0 else None
    dist.gather_object(np.ascontiguousarray(local), objs, dst=0, group=group)
    return objs
