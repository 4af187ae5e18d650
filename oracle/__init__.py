"""CPU oracle for the TensoFlow hot path (TEST INFRASTRUCTURE, not product code).

Every module here is a plain PyTorch-CPU / numpy restatement of the reference's
algorithm for the path named in BASELINE.json `north_star`; each function cites the
reference file:line it follows.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import anything from this package, and only as the
checker -- `tensoflow_amd` never imports it.

Pinning status (see DESIGN.md "Oracle"):
  * everything that is plain Python/PyTorch in the reference is pinned by golden vectors
    generated from the imported reference (`tools/gen_golden.py`, fixtures under
    `tests/golden/`);
  * the third-party arithmetic that is absent from /root/reference (nvdiffrast
    `dr.texture`, nerfacc occupancy marcher, `_raytracing` BVH tie-breaking) is restated
    from published behaviour: PARITY UNPINNED at those boundaries.
"""
