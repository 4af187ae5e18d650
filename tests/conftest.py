import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """npz fixture: arrays as torch tensors, `sd/...` entries collected into .sd."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.sd, self.grad, self.out, self.a = {}, {}, {}, {}
        for k in z.files:
            t = torch.from_numpy(z[k])
            if k.startswith("sd/"):
                self.sd[k[3:]] = t
            elif k.startswith("grad/"):
                self.grad[k[5:]] = t
            elif k.startswith("out/"):
                self.out[k[4:]] = t
            else:
                self.a[k] = t

    def __getitem__(self, k):
        return self.a[k]


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return load


def rel_err(a, b):
    """max |a-b| / max(|b|, 1) -- the 'relative fp32' measure used throughout.  For quantities below 1 (colours, alpha, sdf,
    roughness, weights) this is an ABSOLUTE bound; `true_rel_err` below is the relative one."""
    a, b = a.double(), b.double()
    v = float(((a - b).abs() / b.abs().clamp_min(1.0)).max()) if a.numel() else 0.0
    return v


def true_rel_err(a, b, floor=1e-3):
    """max |a-b| / max(|b|, floor * max|b|): a genuinely relative measure with a small floor against division by ~0 (a quantity
    of size 1e-2 held to 1e-4 here is held to 1e-6 absolute)."""
    a, b = a.double(), b.double()
    if not a.numel():
        return 0.0
    return float(((a - b).abs() / b.abs().clamp_min(floor * float(b.abs().max()) + 1e-300)).max())


def parity(got, ref, tol=1e-4, *, label="", floor=1e-3, rel_tol=None, why=None, truth=None, absolute=False, abs_tol=None):
    """The parity assertion of the GPU tests (round 5).  Prints BOTH measures -- `rel_err` (max |a-b| / max(|b|, 1): an absolute bound
    below 1) and `true_rel_err` (max |a-b| / max(|b|, floor max|b|): relative, with a floor at a thousandth of the array's scale) --
    and asserts on the RELATIVE one:

      * default: both measures < tol;
      * absolute=True: sRGB colours / values in [0,1] whose bar `north_star` states per pixel: the absolute form only;
      * truth=<the same function evaluated in fp64 by the oracle>: a miss of the relative bar is accepted only if this implementation is
        no farther from the fp64 evaluation than the REFERENCE's own fp32 value (`ref`, a golden) is -- worst case and 99th percentile
        of |got - truth| at most twice those of |ref - truth| (+ tol) -- i.e. where no fp32 implementation, the reference on another
        device included, holds tol against the golden; both spreads are printed;
      * rel_tol=<bound>, why=<measured value + cause>: a documented exception without an fp64 twin at hand.
    The absolute-floor measure is held to abs_tol (default: tol) in every mode."""
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    assert got.shape == ref.shape, (label, tuple(got.shape), tuple(ref.shape))
    if not got.numel():
        return 0.0
    a = float(((got - ref).abs() / ref.abs().clamp_min(1.0)).max())
    den = ref.abs().clamp_min(floor * float(ref.abs().max()) + 1e-300)
    e = (got - ref).abs() / den
    r = float(e.max())
    note = ""
    ok_rel = r < tol
    if not ok_rel and rel_tol is None and not absolute:
        from parity_exceptions import lookup          # the ONE table of documented exceptions (measured value + cause per entry)
        hit = lookup(label)
        if hit is not None:
            rel_tol, why = hit
    if ok_rel:
        pass
    elif absolute:
        ok_rel, note = True, "absolute bar (values in [0,1])"
    elif truth is not None:
        # The reference's fp32 value and this implementation's are two ROUNDINGS of one function (`truth`: its fp64 evaluation by the
        # oracle).  Rounding noise is per element, so the two error DISTRIBUTIONS are compared, not the errors element by element:
        # worst case and 99th percentile of |got - truth| may be at most twice the reference's own (+ tol), over the whole array.
        t = truth.detach().double().cpu().reshape(ref.shape)
        e_hip, e_ref = ((got - t).abs() / den).flatten(), ((ref - t).abs() / den).flatten()
        q = lambda x: float(torch.quantile(x, 0.99)) if x.numel() > 1 else float(x.max())
        note = (f"{int((e >= tol).sum())} of {e.numel()} elements beyond {tol:g} of the reference's fp32 value; against the fp64 evaluation the "
                f"reference is off by {float(e_ref.max()):.2e} (max) / {q(e_ref):.2e} (99 %), this implementation by {float(e_hip.max()):.2e} / {q(e_hip):.2e}")
        ok_rel = float(e_hip.max()) <= 2.0 * float(e_ref.max()) + tol and q(e_hip) <= 2.0 * q(e_ref) + tol
        if not ok_rel and rel_tol is not None:       # a documented exception on top (e.g. two rows of an edge-case fixture)
            assert why, "a relative exception needs its evidence"
            ok_rel, note = r < rel_tol, note + f"; documented bound {rel_tol:g}: {why}"
    elif rel_tol is not None:
        assert why, "a relative exception needs its evidence"
        ok_rel, note = r < rel_tol, f"documented bound {rel_tol:g}: {why}"
    print(f"[parity] {label or '?'}: abs-floor {a:.2e}, relative {r:.2e}" + (f"  ({note})" if note else ""))
    assert a < (tol if abs_tol is None else abs_tol), (label, "abs-floor measure", a)
    if os.environ.get("TF_PARITY_DISCOVER"):          # dev: collect what the relative bar would reject instead of stopping at the first
        with open(os.environ["TF_PARITY_DISCOVER"], "a") as f:
            f.write(f"{label}\t{a:.3e}\t{r:.3e}\t{'ok' if ok_rel else 'FAIL'}\t{note}\n")
        return r
    assert ok_rel, (label, "relative measure", r, note)
    return r


def in_fp64(fn, *args, **kw):
    """fn evaluated in double precision: float tensors (also inside dicts / lists / tuples) are widened, torch's default dtype is
    float64 for the duration of the call.  The oracle restates the reference's functions in plain torch, so this is the reference's
    FUNCTION without the reference's fp32 rounding -- the `truth` argument of `parity`."""
    def cv(x):
        if torch.is_tensor(x):
            return x.detach().cpu().double() if x.is_floating_point() else x.detach().cpu()
        if isinstance(x, dict):
            return {k: cv(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return type(x)(cv(v) for v in x)
        return x
    keep = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        with torch.no_grad():
            return fn(*[cv(a) for a in args], **{k: cv(v) for k, v in kw.items()})
    finally:
        torch.set_default_dtype(keep)


AABB = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])
