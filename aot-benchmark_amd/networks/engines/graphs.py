"""hipGraph replay of the per-frame launch sequences.

A propagated frame of R50-AOTL is ~130 kernel launches issued through ctypes (~1.5 ms of host time, more than half of
the frame's GPU time), and nothing in them depends on data: pointers, grid sizes and the bank length are functions of
the engine's STATE (frame geometry, bank slots in use, which buffer holds the previous frame's K/V).  So every distinct
state is captured once as a hipGraph and replayed afterwards -- the 70-frame clips of a sequence set walk through the
same states, clip after clip.

  * capture happens on a stream PRIVATE to the cache (HIP forbids capturing the default stream, and the model's scratch
    is keyed by stream, so a private stream also gives the graphs a scratch set that no eager launch touches); replay is
    a single hipGraphLaunch on whatever stream is current;
  * tensors created while capturing come out of one private memory pool; what a captured stage returns is kept with
    the graph, so it is never handed out again -- temporaries of different graphs may share memory, which is safe because
    the graphs of one engine replay one after the other on one stream;
  * the key of a graph must name everything its launches depend on: every external pointer and every integer argument.
    `ptr_key` builds that from tensors.

torch.cuda.CUDAGraph is only the handle on hipStreamBeginCapture / hipGraphInstantiate / hipGraphLaunch (and tells
torch's caching allocator that a capture is in progress); no tracing is involved, the captured work is the same C-ABI
launches."""
import torch


def ptr_key(*items):
    """Hashable identity of tensors (pointer, shape, strides) and plain values, nested lists/tuples allowed."""
    out = []
    for it in items:
        if isinstance(it, torch.Tensor):
            out.append((it.data_ptr(), tuple(it.shape), tuple(it.stride())))
        elif isinstance(it, (list, tuple)):
            out.append(ptr_key(*it))
        else:
            out.append(it)
    return tuple(out)


class FrameGraphs:
    MAX_GRAPHS = 2048      # states of a handful of clip geometries; beyond that everything is dropped and captured afresh

    def __init__(self, device):
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(self.device)
        self.pool = torch.cuda.graph_pool_handle()
        self._g = {}
        self.captures = 0
        self.replays = 0

    def __len__(self):
        return len(self._g)

    def run(self, key, fn):
        """Replays the graph of `key`, capturing fn() first when the key is new.  Returns what fn returned at capture
        time (tensors whose contents the replay has just refreshed)."""
        ent = self._g.get(key)
        if ent is None:
            if len(self._g) >= self.MAX_GRAPHS:     # e.g. a re-allocated memory bank left every old key unreachable
                self._g.clear()
            g = torch.cuda.CUDAGraph()
            # (capture_begin / capture_end directly: the torch.cuda.graph context manager also synchronises the device,
            #  runs the garbage collector and empties the allocator cache on entry -- per-capture costs that would stall
            #  the clips running on the other streams)
            with torch.cuda.stream(self.stream):
                g.capture_begin(pool=self.pool)
                try:
                    ret = fn()
                finally:
                    g.capture_end()
            ent = self._g[key] = (g, ret)
            self.captures += 1
        ent[0].replay()
        self.replays += 1
        return ent[1]

    def clear(self):
        self._g.clear()
