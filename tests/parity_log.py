"""What the GPU parity tests actually OBSERVED, not only whether they passed: every tolerance check
that goes through `check_close` / `note` leaves its largest error here, and tests/conftest.py writes the collection to
profiles/r06_parity_observed.json at the end of a `-m gpu` session (merged over what earlier sessions wrote, so a run of
a single test file does not erase the rest)."""
import numpy as np

OBSERVED = {}


def note(key, value, bound=None):
    """Keep the largest `value` seen under `key` (and the bound it was held to, if any)."""
    e = OBSERVED.setdefault(key, {'max': 0.0})
    e['max'] = max(e['max'], float(value))
    if bound is not None:
        e['bound'] = float(bound)
    e['checks'] = e.get('checks', 0) + 1


def check_close(key, got, ref, rtol=0.0, atol=0.0):
    """np.testing.assert_allclose(got, ref, rtol, atol) that also records max |got - ref| and the share of the
    tolerance it used (max over elements of |got - ref| / (atol + rtol |ref|))."""
    got, ref = np.asarray(got), np.asarray(ref)
    if got.size:
        err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
        tol = atol + rtol * np.abs(ref.astype(np.float64))
        e = OBSERVED.setdefault(key, {'max': 0.0})
        e['max'] = max(e['max'], float(err.max()))
        e['atol'], e['rtol'] = float(atol), float(rtol)
        e['tolerance_used'] = max(e.get('tolerance_used', 0.0), float((err / np.maximum(tol, 1e-300)).max()))
        e['checks'] = e.get('checks', 0) + 1
    np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol)
