"""Host-side bookkeeping of make_von_mises(state="resident") without a GPU: `_StateMirror` against a recording stand-in for
`_lib.VmState` — when the holders are uploaded, when they are not, what commit_state / state_changed do, and the sampled
tripwire. The device side of the same protocol is tests/test_vm_state_gpu.py."""
import warnings

import numpy as np
import pytest

from dolfinx_external_operator_amd.operators import _StateMirror


class _FakeState:
    def __init__(self, ctx, d, n):
        self.ctx, self.d, self.n = ctx, d, n
        self.uploads, self.commits, self.closed = [], 0, False

    def upload(self, sigma_n, p):
        self.uploads.append((sigma_n.copy(), p.copy()))

    def commit(self):
        self.commits += 1

    def close(self):
        self.closed = True


class _FakeCtx:
    def __init__(self):
        self.states = []

    def vm_state(self, d, n):
        self.states.append(_FakeState(self, d, n))
        return self.states[-1]


def test_upload_once_then_commit_then_tripwire():
    n, d = 10_000, 6
    rng = np.random.Generator(np.random.PCG64(1))
    sigma_n, p = rng.normal(size=n * d), np.abs(rng.normal(size=n))
    c = _FakeCtx()
    m = _StateMirror(sigma_n, p)
    m.commit()                                          # nothing on the device yet: a no-op, not an error
    st = m.sync(c, d, n, sigma_n, p)
    assert len(st.uploads) == 1 and len(c.states) == 1
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        for _ in range(3):                              # Newton iterations of one load step: no traffic
            assert m.sync(c, d, n, sigma_n, p) is st
    assert len(st.uploads) == 1
    sigma_n *= 1.01                                     # the reference's load-step update on the host ...
    p += 0.5
    m.commit()                                          # ... announced: device-side commit, no upload, no warning
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        m.sync(c, d, n, sigma_n, p)
        m.sync(c, d, n, sigma_n, p)
    assert st.commits == 1 and len(st.uploads) == 1
    sigma_n[::7] += 1.0                                 # changed behind the operator's back: tripwire -> warn + upload
    with pytest.warns(RuntimeWarning, match="re-uploading"):
        m.sync(c, d, n, sigma_n, p)
    assert len(st.uploads) == 2 and np.array_equal(st.uploads[-1][0], sigma_n)
    p[5] = 9.0                                          # a single entry the samples may miss: announced by the caller
    m.invalidate()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        m.sync(c, d, n, sigma_n, p)
    assert len(st.uploads) == 3 and st.uploads[-1][1][5] == 9.0


def test_a_new_batch_size_or_context_gets_a_new_mirror():
    c1, c2 = _FakeCtx(), _FakeCtx()
    a, b = np.zeros(600), np.zeros(100)
    m = _StateMirror(a, b)
    s1 = m.sync(c1, 6, 100, a, b)
    a2, b2 = np.zeros(1200), np.zeros(200)
    s2 = m.sync(c1, 6, 200, a2, b2)
    assert s2 is not s1 and s1.closed and len(s2.uploads) == 1
    s3 = m.sync(c2, 6, 200, a2, b2)
    assert s3 is not s2 and s2.closed and s3.ctx is c2 and len(s3.uploads) == 1
    with pytest.raises(RuntimeError, match="no call"):
        _StateMirror(a, b).check()


def test_nan_state_does_not_trip_the_wire():
    a, b = np.full(600, np.nan), np.zeros(100)
    c = _FakeCtx()
    m = _StateMirror(a, b)
    st = m.sync(c, 6, 100, a, b)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        m.sync(c, 6, 100, a, b)                          # NaN == NaN for the comparison of samples
    assert len(st.uploads) == 1
