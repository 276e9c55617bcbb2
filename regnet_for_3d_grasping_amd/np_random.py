"""numpy-compatible batched ``np.random.choice`` draws on numpy's GLOBAL generator, executed by
native host code (csrc/np_random.hip).  Semantics and stream consumption equal the reference's
per-row Python loops; see regnet_np_choice_rows in include/regnet_hip.h."""
import ctypes

import numpy as np

from . import _lib

_check = _lib.check
_L = _lib.lib


def choice_rows(counts, size, mode):
    """counts: int array (any shape) of candidate-list lengths, processed in C order.
    Returns (positions int64 counts.shape + (size,), valid bool counts.shape)."""
    counts32 = np.ascontiguousarray(counts, dtype=np.int32)
    rows = counts32.size
    out = np.empty((rows, size), dtype=np.int64)
    valid = np.empty((rows,), dtype=np.uint8)
    name, key, pos, has_gauss, cached = np.random.get_state()
    if name != "MT19937":
        raise RuntimeError("numpy's global generator is not MT19937")
    key = np.ascontiguousarray(key, dtype=np.uint32).copy()
    cpos = ctypes.c_int32(int(pos))
    _check(_L.regnet_np_choice_rows(key.ctypes.data, ctypes.addressof(cpos), counts32.ctypes.data, rows, int(size),
                                    int(mode), out.ctypes.data, valid.ctypes.data), "np_choice_rows")
    np.random.set_state((name, key, int(cpos.value), has_gauss, cached))
    return out.reshape(counts32.shape + (size,)), valid.astype(bool).reshape(counts32.shape)


def choice_rows_pinned(counts, size, mode):
    """``choice_rows`` with the positions written straight into PINNED host memory: -> (positions int64 torch tensor
    counts.shape + (size,) in page-locked memory, valid bool ndarray).  ``positions.to(device, non_blocking=True)`` is then an
    asynchronous DMA transfer (no staging copy, no blit kernel on the CUs: the drawn positions of a batch of 8 are 5 MB);
    torch's host allocator keeps the block alive until that copy has run."""
    import torch
    counts32 = np.ascontiguousarray(counts, dtype=np.int32)
    rows = counts32.size
    out = torch.empty((rows, int(size)), dtype=torch.int64, pin_memory=True)
    valid = np.empty((rows,), dtype=np.uint8)
    name, key, pos, has_gauss, cached = np.random.get_state()
    if name != "MT19937":
        raise RuntimeError("numpy's global generator is not MT19937")
    key = np.ascontiguousarray(key, dtype=np.uint32).copy()
    cpos = ctypes.c_int32(int(pos))
    _check(_L.regnet_np_choice_rows(key.ctypes.data, ctypes.addressof(cpos), counts32.ctypes.data, rows, int(size),
                                    int(mode), out.data_ptr(), valid.ctypes.data), "np_choice_rows")
    np.random.set_state((name, key, int(cpos.value), has_gauss, cached))
    return out.view(tuple(counts32.shape) + (int(size),)), valid.astype(bool).reshape(counts32.shape)


# ---- the same draws on the DEVICE (csrc/np_random_dev.hip): numpy's generator state lives in HBM -----------------------
class _DeviceStream:
    """numpy's global MT19937 state, resident on one GPU.

    The device copy becomes authoritative with the first device draw and stays so until ``flush()`` hands the state back
    to ``np.random`` (one small device->host copy, the only synchronisation).  If the host's generator was touched in
    between -- ``np.random.seed``, a host-side draw -- the signature taken at the last hand-over no longer matches and the
    next device draw starts from the HOST's state again, as the reference's code (which only knows the host generator)
    would."""

    def __init__(self, device):
        import torch
        self.device = device
        self.key = torch.empty((624,), dtype=torch.int32, device=device)
        self.pos = torch.empty((1,), dtype=torch.int32, device=device)
        self.ahead = False          # the device has consumed words the host generator does not know about
        self.host_sig = None        # signature of the host state the device copy descends from
        self.last_event = None      # orders consecutive uses issued on different torch streams
        self.status = []            # workspaces whose status word has not been checked yet

    @staticmethod
    def _signature(state):
        return (int(state[2]), hash(state[1].tobytes()), int(state[3]), float(state[4]))

    def acquire(self):
        """Make the device state current on torch's current stream."""
        import torch
        state = np.random.get_state()
        if state[0] != "MT19937":
            raise RuntimeError("numpy's global generator is not MT19937")
        sig = self._signature(state)
        cur = torch.cuda.current_stream(self.device)
        if self.last_event is not None:
            cur.wait_event(self.last_event)
        if not self.ahead or sig != self.host_sig:
            key = torch.from_numpy(np.ascontiguousarray(state[1], dtype=np.uint32).view(np.int32).copy())
            self.key.copy_(key)
            self.pos.fill_(int(state[2]))
            self.host_sig = sig
            self.ahead = False
        return cur

    def release(self, cur):
        import torch
        self.ahead = True
        self.last_event = torch.cuda.Event()
        self.last_event.record(cur)

    def flush(self):
        """Hand the state back to ``np.random`` (synchronises with the device work that produced it)."""
        import torch
        if not self.ahead:
            return
        if self._signature(np.random.get_state()) != self.host_sig:
            # the host generator was re-seeded (or used) after the device copy was taken: the device copy is obsolete --
            # never overwrite what the caller put into np.random
            self.ahead = False
            self.status = []
            return
        with torch.cuda.device(self.device):
            if self.last_event is not None:
                torch.cuda.current_stream(self.device).wait_event(self.last_event)
            key = self.key.cpu().numpy().view(np.uint32).copy()      # synchronising copies
            pos = int(self.pos.cpu())
            bad = [w for w in self.status if int(w[-1].cpu()) != 0]
        self.status = []
        if bad:
            raise RuntimeError("device numpy draws: a candidate count exceeded the max_count the workspace was sized for")
        name, _, _, has_gauss, cached = np.random.get_state()
        np.random.set_state((name, key, pos, has_gauss, cached))
        self.host_sig = self._signature(np.random.get_state())
        self.ahead = False


_streams = {}
_deferred = 0


def _stream_for(device):
    import torch
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _streams:
        _streams[idx] = _DeviceStream(torch.device("cuda", idx))
    return _streams[idx]


def choice_rows_device(counts, size, mode, max_count):
    """``choice_rows`` with the counts, the draws and the generator state on the GPU: counts int32 GPU tensor (any
    shape, rows in C order), every count <= max_count -> (positions int64 counts.shape + (size,), valid bool counts.shape),
    GPU tensors, no host synchronisation.  ``flush()`` hands the generator state back to ``np.random``."""
    import torch
    if counts.dtype != torch.int32 or not counts.is_cuda:
        raise TypeError("counts must be an int32 GPU tensor")
    counts = counts.contiguous()
    rows = counts.numel()
    st = _stream_for(counts.device)
    with torch.cuda.device(counts.device):
        out = torch.empty((rows, int(size)), dtype=torch.int64, device=counts.device)
        valid = torch.empty((rows,), dtype=torch.uint8, device=counts.device)
        if rows:
            ws = torch.empty((_L.regnet_np_choice_rows_dev_workspace_ints(rows, int(max_count)),), dtype=torch.int32,
                             device=counts.device)
            cur = st.acquire()
            _check(_L.regnet_np_choice_rows_dev(st.key.data_ptr(), st.pos.data_ptr(), counts.data_ptr(), rows, int(size),
                                                int(max_count), int(mode), out.data_ptr(), valid.data_ptr(),
                                                ws.data_ptr(), cur.cuda_stream), "np_choice_rows_dev")
            st.release(cur)
            st.status.append(ws)
            if len(st.status) > 64:      # bound what an unflushed (deferred) run keeps alive
                st.status = st.status[-64:]
    return out.view(tuple(counts.shape) + (int(size),)), valid.bool().view(counts.shape)


def rand_device(count, device):
    """``np.random.rand(count)`` drawn from the device-resident generator -> float64 GPU tensor (no synchronisation)."""
    import torch
    st = _stream_for(device)
    with torch.cuda.device(st.device):
        out = torch.empty((int(count),), dtype=torch.float64, device=st.device)
        if count:
            cur = st.acquire()
            _check(_L.regnet_np_rand_doubles_dev(st.key.data_ptr(), st.pos.data_ptr(), int(count), out.data_ptr(),
                                                 cur.cuda_stream), "np_rand_doubles_dev")
            st.release(cur)
    return out


def flush(device=None):
    """Give ``np.random`` the state the device draws left behind (no-op when the host state is current)."""
    for st in list(_streams.values()):
        if device is None or st.device == device:
            st.flush()


def flush_unless_deferred():
    if not _deferred:
        flush()


class deferred:
    """``with np_random.deferred():`` -- the region stages inside do not hand the generator back after every call (that
    would be one synchronisation per call); the state is flushed once when the outermost block ends, or whenever host
    code asks for it with ``flush()``."""

    def __enter__(self):
        global _deferred
        _deferred += 1
        return self

    def __exit__(self, *exc):
        global _deferred
        _deferred -= 1
        if not _deferred:
            flush()
        return False
