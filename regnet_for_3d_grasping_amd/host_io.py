"""Small host -> device transfers of the region stage through page-locked memory (train.py / test.py's host-side control flow:
gripper_region_network.py:92-184, :233-309, :532-544 and get_regiondataset.py:311-434 build index / flag arrays with numpy and
``.cuda()`` them one by one)."""
import threading

import numpy as np
import torch

# ---- small host -> device transfers of the region stage ---------------------------------------------------------------------
# The stage's control flow is the reference's host code: a dozen-odd small arrays per batch (drawn row ids, class-balancing picks,
# valid-crop ids, loss scale vectors; a few hundred bytes to a few KiB each) go up.  ``torch.from_numpy(a).to(dev)`` copies from
# PAGEABLE memory: the runtime stages the bytes and the call returns only when they have left (60-80 us of the launching
# thread each, 1.3 ms per training iteration -- on the iteration's critical path once the trunk is replayed from hipGraphs --
# and a blit kernel on the CUs per copy: the `__amd_rocclr_copyBuffer` launches of the kernel statistics).  ``upload`` goes through
# a ring of page-locked slots instead: one memcpy into the slot, an asynchronous DMA transfer out of it, an event per slot so
# that a slot is not rewritten before its transfer has run.
UPLOAD_SLOT_BYTES = 1 << 16
UPLOAD_SLOTS = 64


class _UploadRing(threading.local):
    """Per THREAD (the pipeline's region worker and a trainer may upload at the same time)."""
    slots = None
    events = None
    at = 0


_upload_ring = _UploadRing()


def _slot():
    """The next pinned slot of this thread's ring (its previous transfer, 64 uploads ago, has run) and its index."""
    ring = _upload_ring
    if ring.slots is None:
        ring.slots = [torch.empty((UPLOAD_SLOT_BYTES,), dtype=torch.uint8, pin_memory=True) for _ in range(UPLOAD_SLOTS)]
        ring.events = [None] * UPLOAD_SLOTS
    i = ring.at % UPLOAD_SLOTS
    ring.at += 1
    if ring.events[i] is not None:
        ring.events[i].synchronize()         # (64 uploads ago: long done)
    return ring.slots[i], i


def _sent(i, device):
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    _upload_ring.events[i] = ev


def _as_cpu_tensor(host, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(host)) if isinstance(host, np.ndarray) else host
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t


def upload(host, device, dtype=None):
    """``host`` (numpy array or CPU tensor) -> a tensor on ``device``; small arrays through the pinned ring (asynchronous on the
    current stream), anything larger than a slot -- or a non-GPU ``device`` -- by a plain ``.to``."""
    t = _as_cpu_tensor(host, dtype)
    device = torch.device(device)
    nbytes = t.numel() * t.element_size()
    if device.type != "cuda" or t.is_cuda or nbytes == 0 or nbytes > UPLOAD_SLOT_BYTES:
        return t.to(device)
    slot, i = _slot()
    t = t.contiguous()
    stage = slot[:nbytes].view(t.dtype).view(t.shape)
    stage.copy_(t)
    out = stage.to(device, non_blocking=True)
    _sent(i, device)
    return out


def upload_many(hosts, device):
    """Several small host arrays that exist at the same moment -> their device tensors through ONE transfer (a transfer costs the
    launching thread 60-100 us whatever its size): the arrays are laid into one pinned slot at 16-byte aligned offsets and the
    results are views of the one device buffer that arrives.  Falls back to one ``upload`` each when they do not fit a slot."""
    ts = [_as_cpu_tensor(h).contiguous() for h in hosts]
    device = torch.device(device)
    sizes = [t.numel() * t.element_size() for t in ts]
    offsets, total = [], 0
    for n in sizes:
        offsets.append(total)
        total += (n + 15) // 16 * 16
    if device.type != "cuda" or total == 0 or total > UPLOAD_SLOT_BYTES or any(t.is_cuda for t in ts):
        return [upload(t, device) for t in ts]
    slot, i = _slot()
    for t, o, n in zip(ts, offsets, sizes):
        if n:
            slot[o:o + n].view(t.dtype).view(t.shape).copy_(t)
    dev = slot[:total].to(device, non_blocking=True)
    _sent(i, device)
    return [dev[o:o + n].view(t.dtype).view(t.shape) if n else torch.empty(t.shape, dtype=t.dtype, device=device)
            for t, o, n in zip(ts, offsets, sizes)]
