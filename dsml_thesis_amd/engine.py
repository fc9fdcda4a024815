"""Launch programs: a model forward is compiled once per input shape into a flat list of C-ABI
calls over a static workspace (no allocation, no host sync inside a step), which is what makes
the per-step work capturable in a hipGraph and cheap to replay from Python.

Buffers come from a size-keyed pool with explicit release, so the working set of a whole UNet
evaluation stays small enough to live in the 256 MiB Infinity Cache / L2 between producer and
consumer kernels instead of streaming ~1 GB of fresh HBM lines per step.
"""
import ctypes as C

import torch

from . import lib as L


class Program:
    def __init__(self, device):
        self.device = torch.device(device)
        self.lib = L.load()
        self.calls = []          # (fn, args, keepalive)
        self._free = {}          # numel -> [tensor]
        self._all = []
        self.inputs = {}
        self.outputs = {}

    # ---- workspace -----------------------------------------------------------------------
    def alloc(self, *shape, dtype=torch.float32):
        n = 1
        for s in shape:
            n *= int(s)
        key = (n, dtype)
        lst = self._free.get(key)
        if lst:
            return lst.pop().view(*shape)
        t = torch.empty(n, device=self.device, dtype=dtype)
        self._all.append(t)
        return t.view(*shape)

    def release(self, *tensors):
        for t in tensors:
            if t is None:
                continue
            self._free.setdefault((t.numel(), t.dtype), []).append(t.reshape(-1))

    def workspace_bytes(self):
        return sum(t.numel() * t.element_size() for t in self._all)

    # ---- calls ---------------------------------------------------------------------------
    def add(self, name, *args, keep=None):
        self.calls.append((getattr(self.lib, name), args, keep, name))

    def igemm(self, args, scale_m=None):
        """scale_m = (num, den): pin the tile shape this GEMM would get at M*num/den rows, so that results
        are bitwise independent of how a batch is split across calls / ranks (same K-summation order)."""
        if scale_m is not None and scale_m[0] != scale_m[1]:
            m = args.M
            args.M = max(1, m * scale_m[0] // scale_m[1])
            args.tile_cfg = self.lib.ldmk_igemm_pick_config(C.byref(args))
            args.M = m
        self.calls.append((self.lib.ldmk_igemm, (C.byref(args),), args, "ldmk_igemm"))

    def run(self, stream=None):
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        for fn, args, _, name in self.calls:
            rc = fn(*args, st)
            if rc != 0:
                L.check(rc, name)


class GraphedProgram:
    """A Program (or any callable that only enqueues on the current stream) captured into a hipGraph
    through torch.cuda.CUDAGraph (hipStreamBeginCapture/hipGraphLaunch underneath)."""

    def __init__(self, fn, warmup=2):
        self.fn = fn
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            fn()

    def replay(self):
        self.graph.replay()
