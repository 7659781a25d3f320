"""Shared comparison helpers for the parity tests."""
import hashlib

import numpy as np

TOL = 1e-6   # north_star: dequantized output and gradients within 1e-6 relative of the reference CPU path


def bits_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def assert_bits_equal(got, want, what):
    got = np.ascontiguousarray(got)
    want = np.ascontiguousarray(want).reshape(got.shape)
    if got.tobytes() != want.tobytes():
        gi = got.view(np.uint8).reshape(got.size, -1)
        wi = want.view(np.uint8).reshape(want.size, -1)
        bad = np.nonzero((gi != wi).any(axis=1))[0]
        raise AssertionError("%s: %d of %d elements differ bitwise; first at %d: got %r want %r" %
                             (what, bad.size, got.size, bad[0], got.reshape(-1)[bad[0]], want.reshape(-1)[bad[0]]))


def assert_reduction_close(got, want, abs_terms, what, tol=TOL):
    """|got - want| <= tol * sum|terms| (== tol * |want| when the terms do not cancel); NaN == NaN."""
    got = np.asarray(got, dtype=np.float64).reshape(-1)
    want = np.asarray(want, dtype=np.float64).reshape(-1)
    scale = np.maximum(np.asarray(abs_terms, dtype=np.float64).reshape(-1), 1e-300)
    ok = (np.abs(got - want) <= tol * scale) | (got == want) | (np.isnan(got) & np.isnan(want))
    if not ok.all():
        i = int(np.nonzero(~ok)[0][0])
        raise AssertionError("%s: element %d got %.12g want %.12g (|diff| %.3g, allowed %.3g)" %
                             (what, i, got[i], want[i], abs(got[i] - want[i]), tol * scale[i]))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
