"""Oracle restatement of the closed-form third-party segment ops (TEST INFRASTRUCTURE).

  * torch_scatter.segment_coo(src, index, out, reduce='sum')  -- reference call sites
    network/fields.py:1225,1232-1234,1293 (index is sorted, fields.py:1210);
  * nerfacc.render_weight_from_alpha / accumulate_along_rays   -- reference call sites
    network/shapeRenderer.py:1094,1098,1166-1206,1249.
Neither package is under /root/reference; both ops are closed-form:
sorted-index segmented sum, and per-ray exclusive cumulative product of (1-alpha).
"""
import torch


def segment_coo(src, index, out=None, dim_size=None, reduce="sum"):
    assert reduce == "sum"
    if out is None:
        out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype)
    return out.index_add(0, index, src.to(out.dtype))


def render_weight_from_alpha(alphas, ray_indices=None, n_rays=None, packed_info=None):
    """w_i = alpha_i * prod_{j<i, same ray}(1 - alpha_j); returns (weights, trans)."""
    N = alphas.shape[0]
    if N == 0:
        return alphas.clone(), alphas.clone()
    # exclusive cumprod inside each (contiguous) ray segment, done in a python-free way:
    # log-space would lose exactness, so use a segmented scan by doubling.
    one_minus = 1.0 - alphas
    trans = torch.ones_like(alphas)
    # shift by one inside segment
    same_prev = torch.zeros(N, dtype=torch.bool)
    same_prev[1:] = ray_indices[1:] == ray_indices[:-1]
    shifted = torch.ones_like(alphas)
    shifted[1:] = torch.where(same_prev[1:], one_minus[:-1], torch.ones_like(one_minus[:-1]))
    # inclusive segmented cumprod of `shifted` (Hillis-Steele)
    trans = shifted
    seg_start = torch.arange(N)
    first = ~same_prev
    # index of the first element of each element's segment
    start_idx = torch.where(first, seg_start, torch.zeros_like(seg_start))
    start_idx = torch.cummax(start_idx, 0).values
    pos = seg_start - start_idx
    step = 1
    maxlen = int(pos.max()) + 1
    while step < maxlen:
        prev = torch.ones_like(trans)
        prev[step:] = trans[:-step]
        trans = torch.where(pos >= step, trans * prev, trans)
        step *= 2
    return alphas * trans, trans


def render_weight_from_alpha_seq(alphas, ray_indices):
    """Sequential (left-to-right) product -- the summation order a serial scan gives."""
    a = alphas.tolist()
    r = ray_indices.tolist()
    w, t = [], []
    cur, T = None, 1.0
    import numpy as np
    T = np.float32(1.0)
    for ai, ri in zip(a, r):
        if ri != cur:
            cur, T = ri, np.float32(1.0)
        t.append(float(T))
        w.append(float(np.float32(ai) * T))
        T = np.float32(T * (np.float32(1.0) - np.float32(ai)))
    return torch.tensor(w, dtype=alphas.dtype), torch.tensor(t, dtype=alphas.dtype)


def accumulate_along_rays(weights, values=None, ray_indices=None, n_rays=None):
    if values is None:
        src = weights[:, None]
    else:
        src = weights[:, None] * values
    out = torch.zeros(n_rays, src.shape[-1], dtype=src.dtype)
    if src.shape[0] == 0:
        return out
    return out.index_add(0, ray_indices, src)
