"""Shared by the trace tools: where the NBV steps of a rocprofv3 kernel trace begin.  A step ends with the ONE device-to-host copy that
carries its decision (`__amd_rocclr_copyBuffer`); the next step begins with the first kernel after it.  (Until round 6 the tools cut at
`view_state_kernel` -- which has run BESIDE the step's first search since the split forward of round 4, so that search fell into the
previous segment and left a fake 100-150 us "hole" in front of the parked kNN groups in every gap report.)"""


def step_starts(rows):
    """rows: sorted (start, end, name, ...) tuples -> indices of every step's first kernel (the first non-copy kernel behind a
    device-to-host copy; fragments of fewer than half the usual number of kernels -- uploads between steps -- are merged forward)."""
    is_copy = lambda r: "__amd_rocclr_copyBuffer" in r[2]
    cand = [i for i in range(1, len(rows)) if is_copy(rows[i - 1]) and not is_copy(rows[i])]
    if len(cand) < 3:
        return cand
    sizes = sorted(b - a for a, b in zip(cand[:-1], cand[1:]))
    usual = sizes[len(sizes) // 2]
    out = [cand[0]]
    for c in cand[1:]:
        if c - out[-1] >= usual // 2:
            out.append(c)
    return out
