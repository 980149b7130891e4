"""oracle/transition_oracle.py -- TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench cpu_baseline).

CPU restatement of msmbuilder.msm._transition_counts
(/root/reference/msmbuilder/msm/core.py:487-596): plain Python/numpy loops, every step citing
the reference line it follows.  Pinned against golden vectors produced by the reference
function itself (tests/golden/make_golden.py -> transition_golden.npz) and against the
known-answer cases of the reference's tests/test_transition_counts.py.
"""
import numpy as np


def _invalid(v):
    return v is None or (isinstance(v, (float, np.floating)) and np.isnan(v))


def transition_counts(sequences, lag_time=1, sliding_window=True):
    # core.py:540-542: non-sliding window = stride first, then lag 1
    if (not sliding_window) and lag_time > 1:
        return transition_counts([X[::lag_time] for X in sequences], lag_time=1)
    # core.py:544-555: sorted unique labels without NaN / None
    classes = np.unique(np.concatenate(sequences))
    classes = [c for c in classes if not _invalid(c)]
    n_states = len(classes)
    mapping = dict(zip(classes, range(n_states)))                     # core.py:557
    counts = np.zeros((n_states, n_states), dtype=float)              # core.py:564
    for y in sequences:                                               # core.py:567-585
        y = list(np.asarray(y))
        for t in range(len(y) - lag_time):
            a, b = y[t], y[t + lag_time]
            if _invalid(a) or _invalid(b):                            # core.py:575-579
                continue
            counts[mapping[a], mapping[b]] += 1
    counts /= float(lag_time)                                         # core.py:594
    return counts, mapping
