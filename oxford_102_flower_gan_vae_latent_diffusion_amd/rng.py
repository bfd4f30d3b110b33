"""Graph-safe counter RNG (Philox4x32-10 kernels in csrc/elementwise.hip).

The state is (seed, device-resident int64 counter): draws advance the counter with a tiny kernel on the
same stream, so a captured hipGraph replays fresh numbers without any host involvement. CPU (mt19937) and
GPU (Philox) streams can never agree bit-for-bit, so every stochastic site of the module surface
also accepts explicit noise / mask tensors; this class is the default source when none is given."""
import torch

from . import ops


class Philox:
    def __init__(self, seed=0):
        self.seed = int(seed)
        self._state = {}

    def manual_seed(self, seed):
        self.seed = int(seed)
        for s in self._state.values():
            s.zero_()
        return self

    def state(self, device):
        device = torch.device(device)
        key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
        s = self._state.get(key)
        if s is None:
            s = torch.zeros(1, dtype=torch.int64, device=device)
            self._state[key] = s
        return s

    # offset / advance: a caller that draws several tensors in a row and has a kernel of its own to bump the counter
    # afterwards (the persistent training program's post hook) passes the counters consumed so far as `offset` and
    # advance=False, then adds `fill_count(numel)` / `numel` (randint) itself - same numbers, fewer tiny launches.
    @staticmethod
    def fill_count(numel):
        return (numel + 3) // 4

    def _fill(self, out, mode, p=0.0, offset=0, advance=True):
        st = self.state(out.device)
        ops.philox_fill(out, self.seed, offset, st, mode, p)
        if advance:
            ops.advance_offset(st, offset + (out.numel() + 3) // 4)
        return out

    def normal(self, shape, device, out=None, offset=0, advance=True):
        out = out if out is not None else torch.empty(tuple(shape), dtype=torch.float32, device=device)
        return self._fill(out, 0, 0.0, offset, advance)

    def uniform(self, shape, device):
        return self._fill(torch.empty(tuple(shape), dtype=torch.float32, device=device), 2)

    def dropout(self, shape, device, p, offset=0, advance=True):
        """scale tensor: 0 with probability p, else 1/(1-p)."""
        return self._fill(torch.empty(tuple(shape), dtype=torch.float32, device=device), 1, p, offset, advance)

    def randint(self, shape, device, hi, offset=0, advance=True):
        out = torch.empty(tuple(shape), dtype=torch.int64, device=device)
        st = self.state(device)
        ops.philox_randint(out, self.seed, hi, offset, st)
        if advance:
            ops.advance_offset(st, offset + out.numel())
        return out
