"""The geometry hand-over between the two stages (mesh side): the shape stage's SDF -> a triangle mesh file
-> the material stage's BVH.

* `sdf_lattice`            -- extract_fields (utils/network_utils.py:204-222): the SDF on a resolution^3 lattice over [bound_min,
                              bound_max], values outside the unit sphere replaced by `outside_val`; evaluated on the device by
                              tf_sdf_forward through TensoSDF.sdf, slab by slab.
* `iso_surface`            -- extract_geometry (:224-231).  The reference hands the lattice to PyMCubes (third-party, absent here,
                              unpinned): this module extracts the same iso-surface with MARCHING CUBES (method="cubes", the
                              default: one vertex per sign-changing lattice edge, placed by linear interpolation exactly as
                              PyMCubes places it; the 256-case table is GENERATED below by linking the iso-segments of the six
                              cell faces into loops, with one rule for ambiguous faces, so neighbouring cells always agree and the
                              surface is watertight -- the classic table is not on ambiguous faces; loops are fan-triangulated,
                              so the triangle LIST differs from PyMCubes' while vertices and surface coincide) or with marching
                              TETRAHEDRA on the Kuhn 6-split of every cell (method="tetrahedra": about twice the triangles).
                              Both run in tensor ops on the lattice's device.  Winding follows PyMCubes: face normals point towards DEcreasing lattice values,
                              i.e. into the object for a positive-outside SDF -- the orientation MaterialRenderer.trace undoes
                              with its `-normals` (materialRenderer.py:256-259).  Vertices map to world units like :228-230.
* `write_ply` / `read_ply` -- the file extract_mesh.py:43-47 exports through trimesh (binary little-endian PLY: float x, y, z;
                              faces as `list uchar int vertex_indices`) and materialRenderer.py:148 reads through open3d.
                              The reader accepts ascii and binary_little_endian files with extra vertex properties (normals, colours).
* `extract_mesh`           -- extract_mesh.py:36-47 for a drop-in ShapeRenderer: lattice at level `blend_ratio`, iso-surface at 0.
"""
import numpy as np
import torch

# corner c of a cell = (c & 1, c >> 1 & 1, c >> 2 & 1) along (x, y, z); six tetrahedra around the main diagonal 0-7.  Every face
# diagonal they induce starts at the face's lowest corner, so neighbouring cells agree on it (no cracks).
_TETS = ((0, 1, 3, 7), (0, 1, 5, 7), (0, 2, 3, 7), (0, 2, 6, 7), (0, 4, 5, 7), (0, 4, 6, 7))


def _tet_case_table():
    """case (bit i = vertex i inside) -> up to 2 triangles, each 3 edges (a, b) of tet-local vertices, a inside / b outside."""
    tab = np.zeros((16, 2, 3, 2), np.int64)
    cnt = np.zeros(16, np.int64)
    for case in range(1, 15):
        ins = [i for i in range(4) if case >> i & 1]
        out = [i for i in range(4) if not case >> i & 1]
        if len(ins) == 1:
            tris = [[(ins[0], o) for o in out]]
        elif len(ins) == 3:
            tris = [[(i, out[0]) for i in ins]]
        else:
            (i, j), (k, l) = ins, out
            tris = [[(i, k), (i, l), (j, l)], [(i, k), (j, l), (j, k)]]
        cnt[case] = len(tris)
        for t, tri in enumerate(tris):
            tab[case, t] = tri
    return tab, cnt


_CASE_TAB, _CASE_CNT = _tet_case_table()


def _cube_case_table():
    """Marching-cubes table: case (bit c = corner c inside, corner c = (c & 1, c >> 1 & 1, c >> 2 & 1)) -> triangles, each three cell
    edges (a, b) with a inside / b outside, wound so that the normal points from the inside corners to the outside ones.
    Construction: on every cell face (corners in counter-clockwise order seen from outside) each boundary edge whose corners differ
    carries a vertex; walking the boundary, an inside -> outside crossing starts an iso-segment that ends at the NEXT outside -> inside
    crossing (on a face with four crossings this separates the two outside corners -- the same choice on both cells that share the
    face, because it only looks at the face).  Every crossed cell edge starts one segment (in one of its two faces) and ends one (in
    the other), so the segments chain into closed loops; each loop becomes a triangle fan."""
    corner = np.array([[c & 1, c >> 1 & 1, c >> 2 & 1] for c in range(8)])
    faces = []
    for a in range(3):
        b, c = (a + 1) % 3, (a + 2) % 3
        for side in (0, 1):
            quad = []
            for (ub, uc) in ((0, 0), (1, 0), (1, 1), (0, 1)):           # counter-clockwise seen from +a
                xyz = [0, 0, 0]
                xyz[a], xyz[b], xyz[c] = side, ub, uc
                quad.append(xyz[0] | xyz[1] << 1 | xyz[2] << 2)
            faces.append(quad if side == 1 else quad[::-1])             # seen from -a the same cycle runs clockwise: reverse it
    max_t = 0
    rows = []
    for case in range(256):
        ins = [(case >> c) & 1 for c in range(8)]
        nxt = {}
        for quad in faces:
            exits = [i for i in range(4) if ins[quad[i]] and not ins[quad[(i + 1) % 4]]]
            for i in exits:
                j = (i + 1) % 4
                while not (not ins[quad[j]] and ins[quad[(j + 1) % 4]]):   # the next outside -> inside crossing
                    j = (j + 1) % 4
                start = (quad[i], quad[(i + 1) % 4])                    # (inside corner, outside corner)
                end = (quad[(j + 1) % 4], quad[j])
                nxt[start] = end
        tris, seen = [], set()
        for e0 in list(nxt):
            if e0 in seen:
                continue
            loop, e = [], e0
            while e not in seen:
                seen.add(e)
                loop.append(e)
                e = nxt[e]
            # orientation: the loop's area vector against (centroid of its outside corners - centroid of its inside corners)
            mid = np.array([(corner[a_] + corner[b_]) * 0.5 for a_, b_ in loop])
            area = sum(np.cross(mid[k] - mid[0], mid[k + 1] - mid[0]) for k in range(1, len(loop) - 1))
            grad = np.mean([corner[b_] for _, b_ in loop], 0) - np.mean([corner[a_] for a_, _ in loop], 0)
            if float(np.dot(area, grad)) < 0:
                loop = loop[::-1]
            tris += [(loop[0], loop[k], loop[k + 1]) for k in range(1, len(loop) - 1)]
        rows.append(tris)
        max_t = max(max_t, len(tris))
    tab = np.zeros((256, max_t, 3, 2), np.int64)
    cnt = np.zeros(256, np.int64)
    for case, tris in enumerate(rows):
        cnt[case] = len(tris)
        for t, tri in enumerate(tris):
            tab[case, t] = tri
    return tab, cnt


_CUBE_TAB, _CUBE_CNT = _cube_case_table()


@torch.no_grad()
def sdf_lattice(sdf_fn, bound_min, bound_max, resolution, outside_val=1.0, device="cuda", slab=16):
    """-> u [res, res, res] float32 on `device`; sdf_fn(pts [n,3]) -> [n] or [n,1]."""
    ax = [torch.linspace(float(bound_min[k]), float(bound_max[k]), resolution, device=device) for k in range(3)]
    u = torch.empty(resolution, resolution, resolution, device=device)
    for x0 in range(0, resolution, slab):
        xs = ax[0][x0:x0 + slab]
        xx, yy, zz = torch.meshgrid(xs, ax[1], ax[2], indexing="ij")
        pts = torch.stack([xx, yy, zz], -1).reshape(-1, 3).contiguous()
        val = sdf_fn(pts).reshape(-1).float()
        val = torch.where(pts.norm(dim=-1) >= 1.0, torch.full_like(val, outside_val), val)
        u[x0:x0 + slab] = val.reshape(len(xs), resolution, resolution)
    return u


@torch.no_grad()
def iso_surface(u, threshold=0.0, bound_min=(-1.0, -1.0, -1.0), bound_max=(1.0, 1.0, 1.0), slab=32, normals_to_lower=True, method="cubes"):
    """u [nx, ny, nz] -> vertices [V,3] float32 (world units), triangles [F,3] int64 (both on u's device).
    normals_to_lower: face normals towards decreasing u (PyMCubes' winding); False: towards increasing u.
    method: "cubes" (marching cubes, what the reference's PyMCubes call does) or "tetrahedra"."""
    if method not in ("cubes", "tetrahedra"):
        raise ValueError(f"iso_surface: method {method!r}")
    dev = u.device
    nx, ny, nz = u.shape
    v = u.float() - threshold
    tab, cnt = torch.from_numpy(_CASE_TAB).to(dev), torch.from_numpy(_CASE_CNT).to(dev)
    corner = torch.tensor([[c & 1, c >> 1 & 1, c >> 2 & 1] for c in range(8)], device=dev)
    keys_a, keys_b, flips = [], [], []
    for x0 in range(0, nx - 1, slab):
        x1 = min(x0 + slab, nx - 1)
        # lattice ids and values of the 8 corners of every cell of the slab
        ii, jj, kk = torch.meshgrid(torch.arange(x0, x1, device=dev), torch.arange(ny - 1, device=dev), torch.arange(nz - 1, device=dev),
                                    indexing="ij")
        base = torch.stack([ii, jj, kk], -1).reshape(-1, 3)
        cval = torch.stack([v[x0 + c[0]:x1 + c[0], c[1]:ny - 1 + c[1], c[2]:nz - 1 + c[2]].reshape(-1) for c in corner.tolist()], -1)
        inside = cval < 0
        mixed = inside.any(-1) & ~inside.all(-1)
        if not bool(mixed.any()):
            continue
        base, cval, inside = base[mixed], cval[mixed], inside[mixed]
        if method == "cubes":
            ctab, ccnt = torch.from_numpy(_CUBE_TAB).to(dev), torch.from_numpy(_CUBE_CNT).to(dev)
            case = (inside.long() << torch.arange(8, device=dev)).sum(-1)
            for t in range(ctab.shape[1]):
                sel = ccnt[case] > t
                if not bool(sel.any()):
                    break
                edges = ctab[case[sel], t]                                    # [m,3,2] cell-local corners (inside, outside)
                bs = base[sel]
                pa = bs[:, None, :] + corner[edges[..., 0]]                   # [m,3,3] lattice coordinates
                pb = bs[:, None, :] + corner[edges[..., 1]]
                flips.append(torch.full((bs.shape[0],), bool(normals_to_lower), device=dev))   # the table's loops face the outside corners
                ida = (pa[..., 0] * ny + pa[..., 1]) * nz + pa[..., 2]
                idb = (pb[..., 0] * ny + pb[..., 1]) * nz + pb[..., 2]
                keys_a.append(torch.minimum(ida, idb)); keys_b.append(torch.maximum(ida, idb))
            continue
        for tet in _TETS:
            tv = cval[:, tet]                                               # [n,4]
            case = (inside[:, tet].long() << torch.arange(4, device=dev)).sum(-1)
            for t in range(2):
                sel = cnt[case] > t
                if not bool(sel.any()):
                    continue
                edges = tab[case[sel], t]                                     # [m,3,2] tet-local (inside, outside)
                tvs, bs = tv[sel], base[sel]
                tet_c = corner[list(tet)]                                     # [4,3] cell-local corner offsets
                pa = bs[:, None, :] + tet_c[edges[..., 0]]                    # [m,3,3] lattice coordinates
                pb = bs[:, None, :] + tet_c[edges[..., 1]]
                va, vb = torch.gather(tvs, 1, edges[..., 0]), torch.gather(tvs, 1, edges[..., 1])
                w = (va / (va - vb))[..., None]
                pos = pa + w * (pb - pa)
                # orientation: the linear interpolant's gradient points from inside to outside; M grad = sum v_i (p_i - centroid)
                cen = tet_c.float().mean(0)
                g = (tvs[:, :, None] * (tet_c.float() - cen)[None]).sum(1)
                n = torch.cross(pos[:, 1] - pos[:, 0], pos[:, 2] - pos[:, 0], dim=-1)
                flips.append(((n * g).sum(-1) < 0) != normals_to_lower)
                ida = (pa[..., 0] * ny + pa[..., 1]) * nz + pa[..., 2]
                idb = (pb[..., 0] * ny + pb[..., 1]) * nz + pb[..., 2]
                keys_a.append(torch.minimum(ida, idb)); keys_b.append(torch.maximum(ida, idb))
    if not keys_a:
        return torch.zeros(0, 3, device=dev), torch.zeros(0, 3, dtype=torch.int64, device=dev)
    ka, kb, flip = torch.cat(keys_a), torch.cat(keys_b), torch.cat(flips)
    ntot = nx * ny * nz
    uniq, inv = torch.unique((ka * ntot + kb).reshape(-1), return_inverse=True)
    tris = inv.reshape(-1, 3)
    tris = torch.where(flip[:, None], tris[:, [0, 2, 1]], tris)
    tris = tris[(tris[:, 0] != tris[:, 1]) & (tris[:, 1] != tris[:, 2]) & (tris[:, 0] != tris[:, 2])]
    # vertex positions from the unique edges (one interpolation per edge, so shared vertices are bit-identical)
    ea, eb = uniq // ntot, uniq % ntot
    unravel = lambda e: torch.stack([e // (ny * nz), (e // nz) % ny, e % nz], -1).float()
    fa, fb = v.reshape(-1)[ea], v.reshape(-1)[eb]
    w = (fa / (fa - fb))[:, None]
    lat = unravel(ea) + w * (unravel(eb) - unravel(ea))
    lo, hi = torch.tensor(bound_min, device=dev).float(), torch.tensor(bound_max, device=dev).float()
    scale = (hi - lo) / (torch.tensor([nx, ny, nz], device=dev).float() - 1.0)
    return lat * scale + lo, tris


def write_ply(path, vertices, triangles):
    v = np.ascontiguousarray(np.asarray(vertices, np.float32).reshape(-1, 3))
    f = np.asarray(triangles, np.int32).reshape(-1, 3)
    rec = np.empty(f.shape[0], dtype=[("n", "u1"), ("v", "<i4", (3,))])
    rec["n"], rec["v"] = 3, f
    head = ("ply\nformat binary_little_endian 1.0\ncomment tensoflow_amd.mesh\n"
            f"element vertex {v.shape[0]}\nproperty float x\nproperty float y\nproperty float z\n"
            f"element face {f.shape[0]}\nproperty list uchar int vertex_indices\nend_header\n")
    with open(path, "wb") as fh:
        fh.write(head.encode("ascii"))
        fh.write(v.astype("<f4").tobytes())
        fh.write(rec.tobytes())


_PLY_T = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4", "float": "f4", "double": "f8",
          "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2", "int32": "i4", "uint32": "u4", "float32": "f4", "float64": "f8"}


def read_ply(path):
    """-> vertices [V,3] float32, triangles [F,3] int32.  Triangles only (what extract_mesh.py writes); a polygon with another
    vertex count raises."""
    with open(path, "rb") as fh:
        data = fh.read()
    end = data.index(b"end_header")
    end = data.index(b"\n", end) + 1
    lines = data[:end].decode("ascii", "replace").split("\n")
    if lines[0].strip() != "ply":
        raise ValueError(f"{path}: not a PLY file")
    fmt, elems = None, []
    for ln in lines[1:]:
        tok = ln.split()
        if not tok or tok[0] == "comment":
            continue
        if tok[0] == "format":
            fmt = tok[1]
        elif tok[0] == "element":
            elems.append((tok[1], int(tok[2]), []))
        elif tok[0] == "property":
            elems[-1][2].append(tok[1:])
    if fmt not in ("binary_little_endian", "ascii"):
        raise NotImplementedError(f"{path}: PLY format {fmt}")
    verts = faces = None
    if fmt == "ascii":
        rows = data[end:].decode("ascii").split("\n")
        r = 0
        for name, n, props in elems:
            block = [rows[r + i].split() for i in range(n)]
            r += n
            if name == "vertex":
                names = [p[-1] for p in props]
                cols = [names.index(c) for c in "xyz"]
                verts = np.array([[float(b[c]) for c in cols] for b in block], np.float32).reshape(-1, 3)
            elif name == "face":
                if any(int(b[0]) != 3 for b in block):
                    raise NotImplementedError(f"{path}: only triangle faces are supported")
                faces = np.array([[int(x) for x in b[1:4]] for b in block], np.int32).reshape(-1, 3)
        return verts, faces
    off = end
    for name, n, props in elems:
        if all(p[0] != "list" for p in props):
            dt = np.dtype([(p[1], "<" + _PLY_T[p[0]]) for p in props])
            arr = np.frombuffer(data, dt, n, off)
            off += n * dt.itemsize
            if name == "vertex":
                verts = np.stack([arr["x"], arr["y"], arr["z"]], -1).astype(np.float32)
        else:
            if len(props) != 1:
                raise NotImplementedError(f"{path}: element {name} mixes list and scalar properties")
            _, tc, ti, _ = props[0]
            dt = np.dtype([("n", "<" + _PLY_T[tc]), ("v", "<" + _PLY_T[ti], (3,))])
            arr = np.frombuffer(data, dt, n, off)
            if n and not (arr["n"] == 3).all():
                raise NotImplementedError(f"{path}: only triangle faces are supported")
            off += n * dt.itemsize
            if name == "face":
                faces = arr["v"].astype(np.int32)
    if verts is None or faces is None:
        raise ValueError(f"{path}: needs a vertex and a face element")
    return verts, faces


@torch.no_grad()
def extract_mesh(renderer, resolution=512, path=None, threshold=0.0):
    """extract_mesh.py:36-47 -> (vertices [V,3] float32, triangles [F,3] int32) as numpy; written to `path` (PLY) when given."""
    ratio = float(renderer.cfg.get("blend_ratio", 0))
    dev = renderer.aabb.device
    fn = lambda x: renderer.sdf_network.sdf(x, torch.full((x.shape[0],), ratio, device=x.device))
    u = sdf_lattice(fn, (-1.0, -1.0, -1.0), (1.0, 1.0, 1.0), resolution, device=dev)
    v, f = iso_surface(u, threshold)                     # marching cubes, like the reference's mcubes.marching_cubes
    v, f = v.cpu().numpy().astype(np.float32), f.cpu().numpy().astype(np.int32)
    if path is not None:
        write_ply(path, v, f)
    return v, f
