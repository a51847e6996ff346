"""Per-tensor parity tables for the end-to-end gradient tests.

Every end-to-end test holds each of the 44 (52) parameter gradients to a bound on
`max |got - ref| / max(max |ref|, floor)`.  `check_gradients` computes that figure for every tensor, prints the worst three
on success, and on failure raises with the whole table (name, largest reference entry, absolute and relative error) so that
a red run says WHICH tensor moved and by how much.  When `gpurun_out/` exists the table is also appended to
`gpurun_out/parity_tables.txt` (the GPU box merges that directory back), which is where the bounds below were read from.

fp32 bounds (round 6; SURVEY section 8c suggests "gradients rel <= 1e-4" against real TensorFlow):
  FP32_GRAD_TOL  = 2e-4 of each tensor's largest entry (measured 7e-6 ... 2.4e-5 single-head)
  FP32_NORMAL_TOL = 8e-6 absolute on unit normals (measured 1e-6 ... 2.2e-6)
"""
import os

import numpy as np

FP32_GRAD_TOL = 2e-4
FP32_NORMAL_TOL = 8e-6
GRAD_FLOOR = 1e-3

_REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _np(a):
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.asarray(a)


def gradient_table(names, got, ref, floor=GRAD_FLOOR):
    """rows of (index, name, max |ref|, max abs error, error / max(max |ref|, floor)) for every tensor"""
    rows = []
    for i, (g, r) in enumerate(zip(got, ref)):
        g, r = _np(g).astype(np.float64), _np(r).astype(np.float64)
        assert g.shape == r.shape, "tensor %d (%s): shape %s vs %s" % (i, names[i], g.shape, r.shape)
        big = float(np.abs(r).max()) if r.size else 0.0
        err = float(np.abs(g - r).max()) if r.size else 0.0
        rows.append((i, str(names[i]), big, err, err / max(big, floor)))
    return rows


def format_table(rows, limit=None):
    rows = sorted(rows, key=lambda t: -t[4])
    if limit is not None:
        rows = rows[:limit]
    out = ["  %3s  %-34s %12s %12s %12s" % ("#", "tensor", "max|ref|", "abs err", "rel err")]
    for i, name, big, err, rel in rows:
        out.append("  %3d  %-34s %12.4e %12.4e %12.4e" % (i, name[:34], big, err, rel))
    return "\n".join(out)


def _log(label, text):
    d = os.path.join(_REPO, "gpurun_out")
    if os.path.isdir(d):
        try:
            with open(os.path.join(d, "parity_tables.txt"), "a") as f:
                f.write("== %s\n%s\n" % (label, text))
        except OSError:
            pass


def check_gradients(names, got, ref, tol, label, floor=GRAD_FLOOR):
    """Assert every tensor's relative error < tol; returns the worst one.  Prints the worst three on success, the whole
    table on failure."""
    rows = gradient_table(names, got, ref, floor)
    worst = max(r[4] for r in rows)
    _log("%s (bound %.1e, worst %.3e)" % (label, tol, worst), format_table(rows))
    if worst >= tol:
        raise AssertionError("%s: gradient parity above %.1e (worst %.3e)\n%s" % (label, tol, worst, format_table(rows)))
    print("%s: worst relative gradient error over %d tensors %.3e (bound %.1e); worst three:\n%s"
          % (label, len(rows), worst, tol, format_table(rows, 3)))
    return worst


def check_normals(got, ref, tol, label):
    got, ref = _np(got).astype(np.float64), _np(ref).astype(np.float64)
    err = float(np.abs(got - ref).max())
    _log("%s normals (bound %.1e)" % (label, tol), "  max abs error %.4e" % err)
    assert err < tol, "%s: normals max abs error %.3e >= %.1e" % (label, err, tol)
    print("%s: normals max abs error %.3e (bound %.1e)" % (label, err, tol))
    return err
