"""
Sequences of frames across the GPUs of one node.

Whole frames are independent units (the reference iterates them with a plain ``map``,
auromat/mapping/spacecraft.py:326-332, cli/convert.py:178-185), so the path shards by frame with
no data-path collective: one process per GPU (``torch.distributed``; backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in CPU tests), each rank runs :class:`auromat_amd.pipeline.FramePipeline`
over its contiguous block of frames and the per-frame output grids — a few hundred KB each — are
gathered to rank 0 in one padded ``gather`` (point-to-point to the root; a ring is pointless at
this size).
"""
import numpy as np


class SequenceError(RuntimeError):
    """A rank of a distributed sequence failed (its pipeline raised, or its grids did not fit the agreed gather capacity).
    Raised on EVERY rank by :func:`agree_ok`, so that the whole job ends with a non-zero code instead of some ranks returning
    while others wait in a collective.  `ranks`: the ranks that reported a failure, `messages`: what they said."""

    def __init__(self, ranks, messages):
        RuntimeError.__init__(self, 'rank(s) %s of the sequence failed: %s' % (ranks, '; '.join(messages)))
        self.ranks, self.messages = ranks, messages


def agree_ok(error, device, group=None):
    """
    Every rank calls this at the same point of the program with its own failure — an exception, a message, or None.  When any
    rank reports one, EVERY rank raises :class:`SequenceError` naming those ranks (one all_gather of a flag, and of the
    messages only when something failed); otherwise returns.  The reference has no counterpart: its frames are a ``map`` in one
    process (mapping/spacecraft.py:326-332) and an exception simply ends it.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    flag = torch.tensor([0 if error is None else 1], dtype=torch.int32, device=device)
    flags = [torch.zeros_like(flag) for _ in range(world)]
    dist.all_gather(flags, flag, group=group)
    failed = [r for r, f in enumerate(flags) if int(f.item()) != 0]
    if not failed:
        return
    messages = [None] * world
    dist.all_gather_object(messages, None if error is None else '%s: %s' % (type(error).__name__, error) if isinstance(error, BaseException)
                           else str(error), group=group)
    raise SequenceError(failed, ['rank %d: %s' % (r, messages[r]) for r in failed])


def shard(n_frames, rank, world_size):
    """Contiguous block of frame indices for `rank` (sizes differ by at most one)."""
    base, extra = divmod(n_frames, world_size)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


# ny, nx, nchan+1, lat of first row centre, lon of first column centre, dlat, dlon, frame index,
# contains_pole, contains_discontinuity, altitude [km], magnetic.  The grid of a frame that straddles the 180 deg
# discontinuity is laid out for longitudes shifted by 180 deg and that of a pole frame for coordinates rotated by
# +90 deg about x at `altitude` (reference resample.py:176-218, 262-277): with the two flags (and the altitude) the
# receiver can undo either, see :func:`frame_coordinates`.  ny == 0 marks a frame without any valid pixel.
DESC_LEN = 12


def pack_results(results, indices, device):
    """
    Flatten per-frame results (dicts of resample_frame with keep_on_device=True or host arrays)
    into one float64 payload + one descriptor table.  Layout per frame: mean (ny*nx*(C+1)) then
    count (ny*nx).
    """
    import torch
    descs = describe_results(results, indices)
    parts, _ = payload_parts(results, device)
    payload = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.float64, device=device)
    return torch.from_numpy(descs).to(device), payload


def payload_parts(results, device):
    """The pieces of the payload in order (1-d float64 tensors on `device`) and their total length."""
    import torch
    if hasattr(results, 'payload'):
        # results of the native runner: mean | count of all frames already lie back to back in one arena
        whole = results.payload()
        if whole is not None:
            buf, total = whole
            return [buf if buf.device == torch.device(device) else buf.to(device)], total
    parts, total = [], 0
    for res in results:
        if res is None:
            # no valid pixel in this frame (maskedByElevation would raise ValueError, reference mapping.py:858-859)
            continue
        mean, count = res['mean'], res['count']
        if not isinstance(mean, torch.Tensor):
            mean, count = torch.from_numpy(np.ascontiguousarray(mean)), torch.from_numpy(np.ascontiguousarray(count))
        ny, nx, nc = mean.shape
        packed = res.get('packed')
        if packed is not None and packed.device == mean.device and packed.numel() == ny * nx * (nc + 1):
            # the single-pass plan lays mean and count out one after the other already
            parts.append(packed if packed.device == torch.device(device) else packed.to(device))
        else:
            parts += [mean.reshape(-1).to(device), count.reshape(-1).to(device)]
        total += ny * nx * (nc + 1)
    return parts, total


_ZEROS = {}


def _zero_pad(device, n):
    """n float64 zeros on `device` without a kernel: a slice of a buffer that is kept (and only ever read)."""
    import torch
    key = str(device)
    z = _ZEROS.get(key)
    if z is None or z.numel() < n:
        z = _ZEROS[key] = torch.zeros(max(n, 1 << 16), dtype=torch.float64, device=device)
    return z[:n]


def unpack_results(descs, payload, failed=None):
    """Inverse of :func:`pack_results` on host tensors -> list of dict(index, mean, count, lat0, lon0, dlat, dlon,
    contains_pole, contains_discontinuity, altitude, magnetic).  Frames without a valid pixel are left out; their
    indices are appended to `failed` when a list is given."""
    out, off = [], 0
    descs = descs.cpu().numpy()
    payload = payload.cpu().numpy()
    for d in descs:
        ny, nx, nc = int(d[0]), int(d[1]), int(d[2])
        if ny == 0:
            if failed is not None:
                failed.append(int(d[7]))
            continue
        n_mean, n_cnt = ny * nx * nc, ny * nx
        mean = payload[off:off + n_mean].reshape(ny, nx, nc)
        count = payload[off + n_mean:off + n_mean + n_cnt].reshape(ny, nx)
        off += n_mean + n_cnt
        out.append(dict(index=int(d[7]), mean=mean, count=count, lat0=d[3], lon0=d[4], dlat=d[5], dlon=d[6],
                        contains_pole=bool(d[8]), contains_discontinuity=bool(d[9]), altitude=float(d[10]),
                        magnetic=bool(d[11])))
    return out


def frame_coordinates(frame):
    """
    True cell-centre coordinates (lat_c, lon_c), each (ny, nx), of one unpacked frame: geodetic degrees, or (MLat,
    SM longitude) for a `magnetic` frame.  Undoes the 180 deg shift of a date-line frame and the pole rotation of a
    pole frame exactly as the reference does after binning (resample.py:262-277).
    """
    from .resample import _rotate_pole_host
    from .mapping.mapping import wrap_at_180
    ny, nx = frame['count'].shape
    lat = frame['lat0'] + frame['dlat'] * np.arange(ny)
    lon = frame['lon0'] + frame['dlon'] * np.arange(nx)
    lon_c, lat_c = np.meshgrid(lon, lat)
    if frame['contains_pole']:
        lat_c, lon_c = _rotate_pole_host(lat_c, lon_c, frame['altitude'], -90)
    elif frame['contains_discontinuity']:
        lon_c = wrap_at_180(lon_c - 180)
    return lat_c, lon_c


class Gathered(object):
    """
    What :func:`gather_device` leaves on the destination rank: one [descriptors | payload] buffer per rank, still
    in the memory of `device`, plus the sizes.  :meth:`unpack` copies to the host and splits into frames.
    """

    def __init__(self, bufs, sizes, max_frames):
        self.bufs, self._sizes, self.max_frames = bufs, sizes, max_frames
        self.failed = []            # indices of the frames without any valid pixel (filled by unpack)

    @property
    def sizes(self):
        if self._sizes is None:
            # gathered with an agreed capacity: every buffer ends with its own (frames, payload length)
            import torch
            tail = torch.stack([b[-2:] for b in self.bufs]).cpu().numpy()
            if (tail[:, 0] == -2).any():
                raise ValueError('gather_device: rank(s) %s failed before the gather and sent nothing'
                                 % np.nonzero(tail[:, 0] == -2)[0].tolist())
            if (tail[:, 0] < 0).any():
                raise ValueError('gather_device: the grids of rank(s) %s did not fit the agreed capacity'
                                 % np.nonzero(tail[:, 0] < 0)[0].tolist())
            self._sizes = tail.astype(np.int64)
        return self._sizes

    @property
    def n_frames(self):
        return int(self.sizes[:, 0].sum())

    def unpack(self):
        out = []
        for r, buf in enumerate(self.bufs):
            nf, npay = int(self.sizes[r, 0]), int(self.sizes[r, 1])
            d = buf[:nf * DESC_LEN].reshape(nf, DESC_LEN)
            p = buf[self.max_frames * DESC_LEN:self.max_frames * DESC_LEN + npay]
            out += unpack_results(d, p, self.failed)
        self.failed.sort()
        return sorted(out, key=lambda f: f['index'])


class Packer(object):
    """
    The send buffer of :func:`gather_device` with an agreed capacity, filled WHILE the sequence is processed:
    ``SequencePipeline.process(frames, on_batch=packer.add)`` hands over the frames of every finished launch, and their
    (mean | count) slices are copied to their place in the buffer on the stream that finalises them (a few microseconds
    of host time per frame, hidden behind the next launch's kernel), instead of one concatenation of all frames after
    the last kernel, where nothing hides it.  ``gather_device(..., packer=packer)`` then only writes the descriptors.
    Frames that did not take the single-pass plan (no `packed` slice) are copied at the end.
    """

    def __init__(self, capacity, device, stream=None):
        import torch
        self.max_frames, self.max_payload = capacity
        self.device, self.stream = device, stream
        n = self.max_frames * DESC_LEN + self.max_payload
        if stream is not None:
            with torch.cuda.stream(stream):
                self.buf = torch.empty(n + 2, dtype=torch.float64, device=device)
        else:
            self.buf = torch.empty(n + 2, dtype=torch.float64, device=device)
        self.offset = 0             # payload doubles laid out so far
        self.frames = 0
        self.late = []              # (offset, result) of frames copied in finish()
        self.overflow = False

    def add(self, k0, results):
        """Frames k0, k0+1, ... of the call (in order, every frame exactly once)."""
        import torch
        assert k0 == self.frames, 'frames must arrive in order'
        todo = []
        for res in results:
            self.frames += 1
            if res is None:
                continue
            ny, nx, nc = res['mean'].shape
            m = ny * nx * (nc + 1)
            if self.offset + m > self.max_payload or self.frames > self.max_frames:
                self.overflow = True
                continue
            packed = res.get('packed')
            if packed is not None and packed.numel() == m and packed.device == self.buf.device:
                todo.append((self.offset, packed))
            else:
                self.late.append((self.offset, res))
            self.offset += m
        if not todo:
            return
        base = self.max_frames * DESC_LEN
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                for off, packed in todo:
                    self.buf[base + off:base + off + packed.numel()].copy_(packed, non_blocking=True)
        else:
            for off, packed in todo:
                self.buf[base + off:base + off + packed.numel()].copy_(packed)

    def finish(self, results, indices):
        """Descriptors, late frames and the trailer -> the buffer to send, written on the current stream.  The slices
        that add() copied sit on the packer's own stream (the pipeline's finalise stream), which process() does NOT join
        to the caller's stream for work enqueued by on_batch: the current stream is made to wait for it here, so the
        collective (or any reader on the current stream) sees every frame's payload."""
        import torch
        assert self.frames == len(results), 'every frame of the call must have been added'
        buf, base = self.buf, self.max_frames * DESC_LEN
        if self.stream is not None:
            current = torch.cuda.current_stream(buf.device)
            current.wait_stream(self.stream)
            buf.record_stream(current)
        if self.overflow or len(results) > self.max_frames:
            tail = [-1.0, 0.0]
        else:
            descs = describe_results(results, indices)
            for off, res in self.late:
                mean, count = res['mean'], res['count']
                if not isinstance(mean, torch.Tensor):
                    mean, count = torch.from_numpy(np.ascontiguousarray(mean)), torch.from_numpy(np.ascontiguousarray(count))
                buf[base + off:base + off + mean.numel()] = mean.reshape(-1).to(buf.device)
                buf[base + off + mean.numel():base + off + mean.numel() + count.numel()] = count.reshape(-1).to(buf.device)
            buf[:descs.size].copy_(torch.from_numpy(descs.reshape(-1)), non_blocking=True)
            tail = [float(len(results)), float(self.offset)]
        buf[-2:].copy_(torch.tensor(tail, dtype=torch.float64), non_blocking=True)
        return buf


def describe_results(results, indices):
    """The descriptor table ((n, DESC_LEN) float64, host) of :func:`pack_results`."""
    if hasattr(results, 'descriptors') and results.payload() is not None:
        return results.descriptors(np.asarray(indices, dtype=np.float64))
    descs = np.zeros((len(results), DESC_LEN), dtype=np.float64)
    for i, (res, idx) in enumerate(zip(results, indices)):
        if res is None:
            descs[i, 7] = idx
            continue
        ny, nx, nc = res['mean'].shape
        g = res['grid']
        descs[i] = (ny, nx, nc, g.lat0, g.lon0, g.latStep, g.lonStep, idx,
                    1.0 if res.get('contains_pole') else 0.0, 1.0 if res.get('contains_discontinuity') else 0.0,
                    float(res.get('altitude') or 0.0), 1.0 if res.get('magnetic') else 0.0)
    return descs


def agree_capacity(results, indices, device, margin=1.25, group=None):
    """
    (max_frames, max_payload) over all ranks for results like these, the payload with `margin`: what
    :func:`gather_device` takes as `capacity` to gather later results of the same kind with ONE collective.  One
    all_gather of two numbers and a host synchronisation, on every rank.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n_payload = 0
    whole = results.payload() if hasattr(results, 'payload') else None
    if whole is not None:
        n_payload = whole[1]
    else:
        for res in results:
            if res is not None:
                ny, nx, nc = res['mean'].shape
                n_payload += ny * nx * (nc + 1)
    sizes = torch.tensor([len(results), n_payload], dtype=torch.int64, device=device)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)
    all_sizes = torch.stack(all_sizes).cpu().numpy()
    return int(all_sizes[:, 0].max()), int(all_sizes[:, 1].max() * margin) + 1


def gather_checked(results, indices, device, dst=0, group=None, capacity=None, packer=None, error=None):
    """
    :func:`gather_device` for a job in which a rank may have FAILED before the gather (`error`: its exception; `results` is
    then ignored) or may find that its grids do not fit the agreed capacity: the failed rank still takes part in the
    collective (it sends a buffer that says so: nobody waits for it), the destination reads the sizes, and then all ranks
    agree (:func:`agree_ok`) — every rank raises :class:`SequenceError` when something went wrong anywhere, the destination's
    knowledge of an overflow included.  Two small collectives more than gather_device (outside of a benchmark's timed region
    only when the caller puts them there).
    """
    import torch.distributed as dist
    rank = dist.get_rank(group)
    problem = error
    if error is not None:
        results, indices, packer = _FailedRank(), [], None
    g = None
    try:
        g = gather_device(results, indices, device, dst, group, capacity, packer)
        if g is not None and rank == dst:
            g.sizes                              # (reads the trailers: an overflow or a failed rank shows here)
    except ValueError as e:
        problem = problem or e
    agree_ok(problem, device, group)
    return g


class _FailedRank(list):
    """The results of a rank that failed: no frames; with an agreed capacity its buffer's trailer says (-2, 0)."""
    failed = True


def gather_device(results, indices, device, dst=0, group=None, capacity=None, packer=None):
    """
    Gather every rank's per-frame grids on rank `dst`, device to device.  Two collectives: an all_gather of the
    (frames, payload length) pair, then one gather of [descriptors | payload] padded to the longest.
    With `capacity` = (max_frames, max_payload) agreed before (:func:`agree_capacity`; the same on every rank) it is
    ONE collective and no host synchronisation: every rank sends a buffer of that size which ends with its own
    (frames, payload length).  A rank whose grids do not fit sends (-1, 0) there and nothing else; the destination
    raises ValueError when it reads the sizes (no rank is left waiting in a collective).
    With `packer` (a :class:`Packer` that was handed these very results while they were computed) nothing is left to
    pack but the descriptors.
    Returns a :class:`Gathered` on `dst`, None elsewhere.
    """
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if packer is not None:
        buf = packer.finish(results, indices)
        if rank == dst:
            bufs = [torch.empty_like(buf) for _ in range(world)]
            dist.gather(buf, bufs, dst=dst, group=group)
            return Gathered(bufs, None, packer.max_frames)
        dist.gather(buf, None, dst=dst, group=group)
        return None
    if capacity is not None:
        # [descriptors, padded | payload, padded | frames, payload length]: ONE host-to-device copy (descriptors and
        # trailer) and ONE concatenation straight into that layout (the padding is a slice of a kept zero buffer)
        max_frames, max_payload = capacity
        parts, total = payload_parts(results, device)
        head = np.zeros(max_frames * DESC_LEN + 2, dtype=np.float64)
        if getattr(results, 'failed', False):
            head[-2:] = (-2.0, 0.0)              # this rank failed before the gather (gather_checked)
            parts, total = [], 0
        elif len(results) <= max_frames and total <= max_payload:
            head[:len(results) * DESC_LEN] = describe_results(results, indices).reshape(-1)
            head[-2:] = (len(results), total)
        else:
            head[-2:] = (-1.0, 0.0)
            parts, total = [], 0
        head = torch.from_numpy(head).to(device)
        buf = torch.cat([head[:-2]] + parts + [_zero_pad(device, max_payload - total), head[-2:]])
        if rank == dst:
            bufs = [torch.empty_like(buf) for _ in range(world)]
            dist.gather(buf, bufs, dst=dst, group=group)
            return Gathered(bufs, None, max_frames)
        dist.gather(buf, None, dst=dst, group=group)
        return None
    descs, payload = pack_results(results, indices, device)
    sizes = torch.tensor([descs.shape[0], payload.numel()], dtype=torch.int64, device=device)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)
    all_sizes = torch.stack(all_sizes).cpu().numpy()
    max_frames, max_payload = int(all_sizes[:, 0].max()), int(all_sizes[:, 1].max())
    buf = torch.empty(max_frames * DESC_LEN + max_payload, dtype=torch.float64, device=device)
    buf[:descs.numel()] = descs.reshape(-1)
    buf[descs.numel():max_frames * DESC_LEN] = 0
    buf[max_frames * DESC_LEN:max_frames * DESC_LEN + payload.numel()] = payload
    buf[max_frames * DESC_LEN + payload.numel():] = 0
    if rank == dst:
        bufs = [torch.empty_like(buf) for _ in range(world)]
        dist.gather(buf, bufs, dst=dst, group=group)
        return Gathered(bufs, all_sizes, max_frames)
    dist.gather(buf, None, dst=dst, group=group)
    return None


def collective_device(compute_device, group=None):
    """Where the tensors of the gather live: the GPU for "nccl" (= RCCL: device to device over xGMI), the host for
    "gloo", whose gather takes CPU tensors only (the grids are a few hundred KB per frame)."""
    import torch
    import torch.distributed as dist
    return compute_device if dist.get_backend(group) == 'nccl' else torch.device('cpu')


def gather_results(results, indices, device, dst=0, group=None):
    """:func:`gather_device` + unpacking on the host: the list of frame results sorted by frame index on `dst`."""
    g = gather_device(results, indices, device, dst, group)
    return g.unpack() if g is not None else None


def run_sequence(frames, width, height, altitude=110, fast=True, min_elevation=10.0, pxPerDeg=10,
                 magnetic=False, gather=True, device=None, return_failed=False):
    """
    Process this rank's share of `frames` — a list of (wcsHeader, cameraPosGCRS, photoTime, image)
    tuples, identical on every rank — and gather the grids on rank 0.  Works without an initialised
    process group (single GPU).  Frames without any valid pixel (the reference raises ValueError for them,
    mapping.py:858-859) are left out of the list; with `return_failed` the result is (list, their indices).
    """
    import torch.distributed as dist
    from .pipeline import SequencePipeline
    distributed = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank() if distributed else 0
    world = dist.get_world_size() if distributed else 1
    mine = shard(len(frames), rank, world)
    first = frames[mine[0]][3] if len(mine) else None
    seq = SequencePipeline(width, height, nchan=first.shape[2] if first is not None else 3,
                           img_dtype=first.dtype if first is not None else np.uint16, device=device,
                           altitude=altitude, fast=fast, min_elevation=min_elevation, pxPerDeg=pxPerDeg,
                           magnetic=magnetic)
    # every rank takes part in the collectives whatever its frames did: a frame without a valid pixel travels as
    # an empty descriptor and is reported in `failed`; a rank whose pipeline RAISES (a date outside the IGRF table, a HIP
    # error, no memory) takes part too and says so — then every rank raises SequenceError (gather_checked / agree_ok)
    error = None
    try:
        results = seq.process([frames[k] for k in mine])
    except Exception as e:
        if not distributed:
            raise
        error, results = e, []
    dev = collective_device(seq.ctx.device) if distributed else seq.ctx.device
    failed = []
    if not distributed:
        descs, payload = pack_results(results, mine, dev)
        out = unpack_results(descs, payload, failed)
    elif not gather:
        agree_ok(error, dev)
        descs, payload = pack_results(results, mine, dev)
        out = unpack_results(descs, payload, failed)
    else:
        g = gather_checked(results, mine, dev, error=error)
        out = g.unpack() if g is not None else None
        failed = g.failed if g is not None else failed
    if return_failed:
        return out, failed
    return out


# ---- ONE frame over several GPUs ("a single very large frame") ----------------------------------------

def row_band(height, rank, world_size, multiple=16):
    """Rows [y0, y1) of a frame of `height` rows for `rank`: bands of whole chunks of `multiple` rows (the row-marching
    kernel's work items are 16 rows tall, so a band is cut exactly where the whole frame's chunks are), sizes differing by
    at most one chunk; the last band takes the remainder."""
    chunks = (height + multiple - 1) // multiple
    base, extra = divmod(chunks, world_size)
    c0 = rank * base + min(rank, extra)
    c1 = c0 + base + (1 if rank < extra else 0)
    return min(c0 * multiple, height), min(c1 * multiple, height)


class RowShard(object):
    """The two exchange steps of a frame whose rows are spread over the ranks of `group` (what FramePipeline.shard and
    resample_frame(shard=...) call): the bounding-box reduction (min / max / count: one all_gather of 8 doubles) and
    the all-reduce(sum) of the integer accumulators.  Integer sums do not depend on the order of the ranks, so the
    result is the one a single GPU computes."""

    def __init__(self, pole, group=None):
        import torch.distributed as dist
        self.pole = bool(pole)
        self.group = group
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)

    def box(self, red):
        import torch
        import torch.distributed as dist
        mine = torch.as_tensor(np.asarray(red, dtype=np.float64))
        if self.backend == 'nccl':
            mine = mine.cuda()
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine, group=self.group)
        a = torch.stack(parts).cpu().numpy()
        out = np.array(red, dtype=np.float64)
        for k in (0, 2, 4):
            out[k] = a[:, k].min()          # lat_min, lon_min, smallest positive longitude (+inf where a band has none)
        for k in (1, 3, 5):
            out[k] = a[:, k].max()          # lat_max, lon_max, largest non-positive longitude
        out[6] = a[:, 6].sum()              # corners kept
        out[7] = a[:, 7].max()
        return out

    def acc(self, acc):
        import torch.distributed as dist
        if self.backend == 'nccl':
            dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=self.group)
            return
        host = acc.cpu()                    # (gloo in the tests: through the host)
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
        acc.copy_(host)


def resample_frame_sharded(wcsHeader, altitude, cameraPosGCRS, photoTime, img, pxPerDeg=10, min_elevation=10.0,
                           fast=True, containsPole=None, group=None, device=None, keep_on_device=False):
    """
    Georeference and mean-resample ONE frame with its rows spread over the ranks of `group` (every rank calls this
    with the same arguments; `img` may be the whole (h, w, 3) image or only this rank's rows): each rank computes the
    coordinate arrays of its band — the corner row between two bands is computed by both, nothing is exchanged for
    it —, the ranks agree on the bounding box (min / max), bin their pixels on the common grid and all-reduce the
    integer accumulators; every rank returns the complete grids (reference: resample.py:159-279 on the whole frame).

    :return: (result dict of :func:`auromat_amd.resample.resample_frame`, FramePipeline holding this rank's band of
             the per-pixel arrays, (y0, y1))
    """
    import torch.distributed as dist
    from .pipeline import FramePipeline, pole_in_view
    from .mapping.astrometry import frame_params
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    w, h = int(wcsHeader['IMAGEW']), int(wcsHeader['IMAGEH'])
    y0, y1 = row_band(h, rank, world)
    assert y1 > y0, 'more ranks than 16-row chunks'
    band = dict(wcsHeader)
    band['CRPIX2'] = wcsHeader['CRPIX2'] - y0           # the same camera, rows counted from y0
    band['IMAGEH'] = y1 - y0
    img = np.asarray(img)
    rows = img[y0:y1] if img.shape[0] == h else img
    assert rows.shape[0] == y1 - y0 and rows.shape[1] == w
    pipe = FramePipeline(w, y1 - y0, nchan=rows.shape[2], img_dtype=rows.dtype, device=device)
    # (the pole is in view of the frame, not of a band: decided from the whole frame's camera model on every rank)
    whole = frame_params(wcsHeader, altitude, cameraPosGCRS, photoTime, fast)
    pipe.shard = RowShard(pole_in_view(whole, min_elevation), group)
    res = pipe.run(band, altitude, cameraPosGCRS, photoTime, img=rows, fast=fast, min_elevation=min_elevation,
                   pxPerDeg=pxPerDeg, containsPole=containsPole, keep_on_device=keep_on_device, fuse=False)
    return res, pipe, (y0, y1)
