"""Hidden-RNG pinning for multi-step goldens (shared by tests/golden/make_golden.py and the GPU tests).  TEST INFRASTRUCTURE ONLY.

The reference draws its hidden randomness (Cell.fill's subset, the partial point cloud's subset, SconeOcc.forward's three
down-samples) with `torch.randperm` on the global CPU generator; a golden of ONE call can replay those draws from a seed.  A
golden of a ten-step trajectory cannot be *generated* that way: the generator re-runs the trajectory while it redraws the few
proxy points that sit on a numerical decision boundary (kNN k / k+1 ties, bin edges), every redraw changes how much of the
global stream the earlier calls consume, hence every later permutation, hence every later tie -- it never converges.

Here every draw is a function of its POSITION in the run instead: (base seed, step, stage, index of the call inside the stage,
size).  The harness names the stages; the code under test (the reference there, macarons_amd here) just calls torch.randperm
as it always does.  Both sides make the same calls in the same order with the same sizes exactly when they consume the hidden
randomness identically -- which is the contract being tested -- and a mismatch in order or size raises at once when a log of
the expected calls is given.
"""
import contextlib

import numpy as np
import torch


class KeyedRandperm:
    def __init__(self, base, expect=None):
        self.base, self.step, self.stage, self.k = int(base), 0, "", 0
        self.log = []                                  # (step, stage, k, n) of every call made
        self.expect = expect                           # optional: {(step, stage): [n, n, ...]} recorded by the generator
        self._real = torch.randperm

    def at(self, step, stage):
        self.step, self.stage, self.k = int(step), str(stage), 0

    def _seed(self, n):
        h = (self.base * 1000003 + self.step * 7919 + sum(ord(c) * (i + 1) for i, c in enumerate(self.stage)) * 104729 + self.k * 31337) % (2 ** 31 - 1)
        return h

    def __call__(self, n, *a, **kw):
        n = int(n)
        if self.expect is not None:
            want = self.expect.get((self.step, self.stage))
            if want is None or self.k >= len(want) or int(want[self.k]) != n:
                raise AssertionError(f"hidden-RNG contract: call {self.k} of stage '{self.stage}' (step {self.step}) asks for a permutation of "
                                     f"{n}; the reference asked for {None if want is None else list(want)[self.k:self.k + 1]}")
        g = torch.Generator().manual_seed(self._seed(n))
        self.log.append((self.step, self.stage, self.k, n))
        self.k += 1
        return self._real(n, generator=g)

    def peek(self, step, stage, k, n):
        """The permutation call k of (step, stage) returns / returned for size n, without touching the position."""
        keep = (self.step, self.stage, self.k)
        self.step, self.stage, self.k = int(step), str(stage), int(k)
        g = torch.Generator().manual_seed(self._seed(int(n)))
        self.step, self.stage, self.k = keep
        return self._real(int(n), generator=g)

    def sizes(self):
        """{(step, stage): [n of call 0, n of call 1, ...]} of the calls made so far."""
        out = {}
        for step, stage, _, n in self.log:
            out.setdefault((step, stage), []).append(n)
        return out

    @contextlib.contextmanager
    def installed(self):
        torch.randperm = self
        try:
            yield self
        finally:
            torch.randperm = self._real


def keyed_uniforms(base, step, cam, n=2048):
    """The sampling uniforms of neighbour camera `cam` at step `step` (what torch.rand(n, 1) returns to the sampler)."""
    g = torch.Generator().manual_seed((int(base) * 2654435761 + 97 * int(step) + int(cam) + 12345) % (2 ** 31 - 1))
    return torch.rand(n, 1, generator=g)


def pack_sizes(sizes, stages):
    """sizes dict -> (flat int32 array, index array [n_steps*len(stages)+1]) for an .npz; inverse: unpack_sizes."""
    steps = sorted({s for s, _ in sizes})
    flat, off = [], [0]
    for s in steps:
        for st in stages:
            flat += sizes.get((s, st), [])
            off.append(len(flat))
    return np.asarray(flat, np.int32), np.asarray(off, np.int32)


def unpack_sizes(flat, off, n_steps, stages):
    out, i = {}, 0
    for s in range(n_steps):
        for st in stages:
            out[(s, st)] = [int(v) for v in flat[off[i]:off[i + 1]]]
            i += 1
    return out
