"""A rasterizer step replayed from a HIP graph.

Small scenes (BASELINE config c1: 10,000 Gaussians at 256x256; early training; inference on small scenes) are bound by
the HOST: a forward + backward of the op is ~30 kernel launches of 5 - 40 us each, and enqueueing them takes longer than
running them.  The reference cannot capture its step in a graph - `CudaRasterizer::Rasterizer::forward` reads
`num_rendered` back in the middle of its enqueue (rasterizer_impl.cu:283) to size the binning buffer.  With the option
`sync_free` (include/f3dgs.h) this library's forward call carves that buffer for a provision and lets the kernels read the
count on the device, so a whole step - forward, loss, backward - is capturable with `torch.cuda.graph`.

`CapturedStep(fn)` does the bookkeeping a capture needs:
  * warm-up calls of `fn` on the capture stream with `sync_free = 1` (they allocate the pinned count words, read the scene's
    instance count and so size the provision: 1.25 x the count + 4096 entries, or option `instance_capacity`);
  * the capture itself; `fn` must read and write STATIC tensors (PyTorch's rule for graphs: new inputs are copied into them);
  * after a replay, `check()` (it synchronises) reads the words the captured forward call reports in: a frame whose lists
    did not fit the provision is void - its emit waves stored nothing - and `check()` re-captures with room and replays.

`fn` may call the rasterizer once (several calls on one stream share the pinned words: the sticky no-room word still catches
any of them, the counts are the last call's).
"""
from __future__ import annotations

from typing import Callable, Optional

import torch


class CapturedStep:
    def __init__(self, fn: Callable[[], object], warmup: int = 2, stream: Optional["torch.cuda.Stream"] = None,
                 capacity: int = 0):
        from diff_gaussian_rasterization import _C
        self._C = _C
        self.fn = fn
        self.warmup = max(1, int(warmup))
        self.stream = stream if stream is not None else torch.cuda.Stream()
        self.capacity = int(capacity)
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.result = None
        self.counts_address = 0
        self.captures = 0
        self._saved = None

    # -- options: sync_free for the warm-up and the capture, the caller's settings afterwards ----------------------------
    def _enter(self):
        self._saved = (self._C.get_option("sync_free"), self._C.get_option("instance_capacity"))
        self._C.set_option("sync_free", 1)
        self._C.set_option("instance_capacity", self.capacity)

    def _leave(self):
        self._C.set_option("sync_free", self._saved[0])
        self._C.set_option("instance_capacity", self._saved[1])

    def capture(self) -> "CapturedStep":
        self._enter()
        try:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                for _ in range(self.warmup):
                    self.fn()
            self.stream.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self.stream):
                self.result = self.fn()
            self.counts_address = self._C.forward_counts_address()
            torch.cuda.current_stream().wait_stream(self.stream)
        finally:
            self._leave()
        self.graph = g
        self.captures += 1
        return self

    def replay(self):
        if self.graph is None:
            self.capture()
        self.graph.replay()
        return self.result

    def counts(self):
        """(list entries, the reference's num_rendered, long-axis flag, entries provided for, no-room word) of the last replayed
        frame; synchronises."""
        torch.cuda.synchronize()
        import ctypes
        if not self.counts_address:
            return None
        words = (ctypes.c_uint32 * 5).from_address(self.counts_address)
        return tuple(int(w) for w in words)

    def check(self) -> bool:
        """True if the last replayed frame(s) fitted the provision.  Otherwise the step is captured again with room for what
        the frame needed (1.25 x) and replayed once; returns False to say that the earlier results were void."""
        c = self.counts()
        if c is None or (c[4] == 0 and c[0] <= c[3]):
            return True
        self._C.clear_forward_overflow(self.counts_address)
        self.capacity = int(c[0] * 1.25) + 4096
        self.graph = None
        self.capture()
        self.graph.replay()
        return False
