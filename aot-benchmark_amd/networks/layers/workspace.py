"""Persistent device scratch buffers.  Buffers are allocated once per (stream, name, shape) and reused every
frame, so the steady state makes no allocator calls for intermediates and all pointers are stable
(a prerequisite for hipGraph capture of the frame).  Keying on the current HIP stream gives every
concurrently running clip (one stream each, see bench.py --streams) its own scratch set."""
import torch


class Workspace:
    def __init__(self):
        self._bufs = {}
        self.salt = ''     # appended to every name while set: a second, disjoint set of the same buffers (the engine's overlapped look-ahead)

    def get(self, name, shape, device, dtype=torch.float32):
        # (raw stream handle through the C binding: torch.cuda.current_stream() builds a Stream object per call, which at
        #  ~110 lookups per frame cost 0.5 ms of host time per frame)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        key = (name + self.salt if self.salt else name, tuple(shape), dtype, idx, torch._C._cuda_getCurrentRawStream(idx))
        buf = self._bufs.get(key)
        if buf is None:
            buf = torch.empty(shape, dtype=dtype, device=device)
            self._bufs[key] = buf
        return buf

    def get_zeroed(self, name, shape, device, dtype=torch.float32):
        """Like get(), but the buffer is zero-filled when it is first created (ticket words of in-launch reductions)."""
        n = len(self._bufs)
        buf = self.get(name, shape, device, dtype)
        if len(self._bufs) != n:
            buf.zero_()
        return buf

    def clear(self, stream=None):
        """Drops every buffer (or only those of one raw stream handle): the next get() allocates again."""
        if stream is None:
            self._bufs.clear()
        else:
            for key in [k for k in self._bufs if k[4] == stream]:
                del self._bufs[key]

    def nbytes(self):
        return sum(b.numel() * b.element_size() for b in self._bufs.values())
