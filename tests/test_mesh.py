"""Mesh hand-over between the stages (tensoflow_amd/mesh.py): iso-surface extraction + PLY files on CPU, the SDF lattice ->
mesh -> BVH -> MaterialRenderer chain on the GPU."""
import numpy as np
import pytest
import torch


def _sphere_lattice(res, r=0.6, center=(0.05, -0.1, 0.02)):
    ax = torch.linspace(-1, 1, res)
    xx, yy, zz = torch.meshgrid(ax, ax, ax, indexing="ij")
    c = torch.tensor(center)
    return torch.sqrt((xx - c[0]) ** 2 + (yy - c[1]) ** 2 + (zz - c[2]) ** 2) - r, c


def _edge_counts(f):
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    und = np.sort(e, 1)
    _, cnt = np.unique(und, axis=0, return_counts=True)
    # directed edges: a consistently oriented closed surface has every directed edge once (its reverse belongs to the neighbour)
    _, dcnt = np.unique(e, axis=0, return_counts=True)
    return cnt, dcnt


METHODS = ("cubes", "tetrahedra")


def test_cube_table_is_closed_complete_and_symmetric():
    """The generated 256-case table: every sign-changing cell edge is used, inside each cell every triangle edge is either shared by
    two triangles (a fan's diagonals) or lies on a cell face; complementary cases use the same cell edges."""
    from tensoflow_amd.mesh import _CUBE_CNT, _CUBE_TAB
    corner = np.array([[c & 1, c >> 1 & 1, c >> 2 & 1] for c in range(8)])
    assert _CUBE_CNT[0] == 0 and _CUBE_CNT[255] == 0 and _CUBE_CNT.max() <= 5
    for case in range(1, 255):
        ins = [(case >> c) & 1 for c in range(8)]
        cut = {(a, b) for a in range(8) for b in range(8) if ins[a] and not ins[b] and np.abs(corner[a] - corner[b]).sum() == 1}
        tris = _CUBE_TAB[case, :_CUBE_CNT[case]]
        used = {tuple(e) for t in tris for e in t.tolist()}
        assert used == cut, case
        # triangle count of a set of loops over len(cut) vertices: sum(len - 2) -> between len/3 and len - 2
        assert len(cut) / 3 <= len(tris) <= len(cut) - 2
        comp = _CUBE_TAB[255 - case, :_CUBE_CNT[255 - case]]
        assert {tuple(e[::-1]) for t in comp for e in t.tolist()} == cut


@pytest.mark.parametrize("method", METHODS)
def test_iso_surface_of_a_random_field_is_watertight(method):
    """White noise exercises every case, ambiguous faces included: neighbouring cells must agree on every shared face."""
    from tensoflow_amd.mesh import iso_surface
    g = torch.Generator().manual_seed(3)
    u = torch.rand(14, 13, 12, generator=g) - 0.5
    u = torch.nn.functional.pad(u, (1, 1, 1, 1, 1, 1), value=1.0)        # positive shell: every component closes inside the lattice
    v, f = iso_surface(u, 0.0, slab=5, method=method)
    cnt, dcnt = _edge_counts(f.numpy())
    assert (cnt % 2 == 0).all() and (dcnt <= 2).all()                    # closed (an edge may be shared by two sheets that touch)
    und = np.sort(np.concatenate([f.numpy()[:, [0, 1]], f.numpy()[:, [1, 2]], f.numpy()[:, [2, 0]]]), 1)
    fwd = np.concatenate([f.numpy()[:, [0, 1]], f.numpy()[:, [1, 2]], f.numpy()[:, [2, 0]]])
    sign = np.where(fwd[:, 0] < fwd[:, 1], 1, -1)
    _, inv = np.unique(und, axis=0, return_inverse=True)
    assert (np.bincount(inv.ravel(), weights=sign) == 0).all()           # oriented: every edge as often forwards as backwards
    # enclosed volume = volume of {u < 0} to the accuracy of trilinear cells: compare with a fine resampling
    a, b, c = (v[f[:, k]].double() for k in range(3))
    vol = float(-(a * torch.cross(b, c, dim=-1)).sum() / 6)              # default winding faces decreasing u: inwards -> negate
    assert vol > 0
    if method == "cubes":
        assert f.shape[0] < iso_surface(u, 0.0, method="tetrahedra")[1].shape[0] * 0.62


@pytest.mark.parametrize("method", METHODS)
def test_iso_surface_sphere_is_closed_oriented_and_accurate(method):
    from tensoflow_amd import mesh
    iso_surface = lambda *a, **k: mesh.iso_surface(*a, method=method, **k)
    res, r = 48, 0.6
    u, c = _sphere_lattice(res, r)
    v, f = iso_surface(u, 0.0, slab=7, normals_to_lower=False)           # a slab size that does not divide the lattice
    v, f = v.numpy().astype(np.float64), f.numpy()
    assert f.shape[0] > (5000 if method == "tetrahedra" else 2500) and f.min() == 0 and f.max() == v.shape[0] - 1
    cnt, dcnt = _edge_counts(f)
    assert (cnt == 2).all() and (dcnt == 1).all()                       # watertight 2-manifold, consistent winding
    h = 2.0 / (res - 1)
    rad = np.linalg.norm(v - c.numpy(), axis=1)
    assert np.abs(rad - r).max() < 0.5 * h * h / r + 1e-6               # linear interpolation of a distance field: O(h^2 / r)
    a, b, cc = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    n = np.cross(b - a, cc - a)
    cen = (a + b + cc) / 3 - c.numpy()
    big = np.linalg.norm(n, axis=1) > 1e-12
    assert (np.einsum("ij,ij->i", n, cen)[big] > 0).all()               # normals point to positive SDF (outwards)
    area = 0.5 * np.linalg.norm(n, axis=1).sum()
    assert abs(area - 4 * np.pi * r * r) / (4 * np.pi * r * r) < 5e-3
    vol = np.einsum("ij,ij->i", a - c.numpy(), np.cross(b - c.numpy(), cc - c.numpy())).sum() / 6
    assert abs(vol - 4 / 3 * np.pi * r ** 3) / (4 / 3 * np.pi * r ** 3) < 5e-3
    # default winding (PyMCubes': normals towards decreasing values) is the mirror image: same vertices, every triangle reversed
    vd, fd = iso_surface(u, 0.0)                                          # (default slab: the triangles come in another order)
    rows = lambda t: np.unique(np.sort(t, 1), axis=0)
    assert np.array_equal(vd.numpy().astype(np.float64), v) and np.array_equal(rows(fd.numpy()), rows(f))
    fd = fd.numpy()
    ad, bd, cd = v[fd[:, 0]], v[fd[:, 1]], v[fd[:, 2]]
    assert (np.einsum("ij,ij->i", np.cross(bd - ad, cd - ad), (ad + bd + cd) / 3 - c.numpy())[big] < 0).all()
    # threshold and bounds: the 0.1 level set of the same lattice mapped to a [0,2]^3 box
    v2, f2 = iso_surface(u, 0.1, bound_min=(0.0, 0.0, 0.0), bound_max=(2.0, 2.0, 2.0))
    rad2 = np.linalg.norm(v2.numpy() - (c.numpy() + 1.0), axis=1)
    assert np.abs(rad2 - (r + 0.1)).max() < 1e-3
    # nothing to extract
    v0, f0 = iso_surface(torch.ones(5, 5, 5))
    assert v0.shape == (0, 3) and f0.shape == (0, 3)


@pytest.mark.parametrize("method", METHODS)
def test_iso_surface_handles_lattice_points_on_the_surface_and_two_components(method):
    from tensoflow_amd import mesh
    iso_surface = lambda *a, **k: mesh.iso_surface(*a, method=method, **k)
    ax = torch.linspace(-1, 1, 33)                                       # h = 1/16: the planes x = +-0.5 pass through lattice points
    xx, yy, zz = torch.meshgrid(ax, ax, ax, indexing="ij")
    box = torch.maximum(torch.maximum(xx.abs() - 0.5, yy.abs() - 0.25), zz.abs() - 0.75)
    ball = torch.sqrt((xx - 0.8) ** 2 + yy ** 2 + zz ** 2) - 0.15
    v, f = iso_surface(torch.minimum(box, ball), normals_to_lower=False)
    assert torch.isfinite(v).all() and f.shape[0] > 0
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    vol = float((a * torch.cross(b, c, dim=-1)).sum() / 6)
    # the box's 12 edges come out chamfered by one cell (cross-section h^2/2 over a total edge length of 12: 0.023 for whole-cell
    # chamfers, which is what marching cubes cuts; the tetrahedra's diagonals cut about half of that)
    assert abs(vol - (1.0 * 0.5 * 1.5 + 4 / 3 * np.pi * 0.15 ** 3)) < (0.03 if method == "cubes" else 0.02)


def test_ply_round_trip_and_foreign_layouts(tmp_path):
    from tensoflow_amd.mesh import read_ply, write_ply
    rng = np.random.default_rng(0)
    v = rng.standard_normal((50, 3)).astype(np.float32)
    f = rng.integers(0, 50, (80, 3)).astype(np.int32)
    p = str(tmp_path / "m.ply")
    write_ply(p, v, f)
    head = open(p, "rb").read(200)
    assert head.startswith(b"ply\nformat binary_little_endian 1.0\n") and b"property list uchar int vertex_indices" in head
    v2, f2 = read_ply(p)
    assert v2.dtype == np.float32 and f2.dtype == np.int32 and np.array_equal(v2, v) and np.array_equal(f2, f)
    # a binary file with extra vertex properties (normals + colours, as open3d / MeshLab write) and uint face indices
    dt = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    rec = np.zeros(50, dt)
    rec["x"], rec["y"], rec["z"] = v.T
    fr = np.zeros(80, np.dtype([("n", "u1"), ("v", "<u4", (3,))]))
    fr["n"], fr["v"] = 3, f
    hdr = ("ply\nformat binary_little_endian 1.0\ncomment made elsewhere\nelement vertex 50\n" + "".join(f"property float {k}\n" for k in ("x", "y", "z", "nx", "ny", "nz"))
           + "property uchar red\nproperty uchar green\nproperty uchar blue\nelement face 80\nproperty list uchar uint vertex_indices\nend_header\n")
    p2 = str(tmp_path / "n.ply")
    open(p2, "wb").write(hdr.encode() + rec.tobytes() + fr.tobytes())
    v3, f3 = read_ply(p2)
    assert np.array_equal(v3, v) and np.array_equal(f3, f)
    # ascii
    p3 = str(tmp_path / "a.ply")
    with open(p3, "w") as fh:
        fh.write("ply\nformat ascii 1.0\nelement vertex 50\nproperty float x\nproperty float y\nproperty float z\nelement face 80\n"
                 "property list uchar int vertex_indices\nend_header\n")
        fh.writelines(f"{a:.9g} {b:.9g} {c:.9g}\n" for a, b, c in v)
        fh.writelines(f"3 {a} {b} {c}\n" for a, b, c in f)
    v4, f4 = read_ply(p3)
    assert np.array_equal(v4, v) and np.array_equal(f4, f)
    # quads are refused, garbage is refused
    fr4 = np.zeros(1, np.dtype([("n", "u1"), ("v", "<i4", (3,))])); fr4["n"] = 4
    p4 = str(tmp_path / "q.ply")
    open(p4, "wb").write(b"ply\nformat binary_little_endian 1.0\nelement vertex 0\nproperty float x\nproperty float y\nproperty float z\n"
                         b"element face 1\nproperty list uchar int vertex_indices\nend_header\n" + fr4.tobytes())
    with pytest.raises(NotImplementedError):
        read_ply(p4)
    open(p4, "wb").write(b"solid stl\nend_header\n")
    with pytest.raises(ValueError):
        read_ply(p4)


@pytest.mark.gpu
def test_extract_mesh_feeds_the_material_stage(tmp_path):
    """ShapeRenderer SDF -> lattice (tf_sdf_forward) -> iso-surface -> PLY -> MaterialRenderer's BVH: rays through the centre hit
    the extracted surface where the SDF changes sign."""
    from tensoflow_amd import ops
    from tensoflow_amd.mesh import extract_mesh, read_ply
    from tensoflow_amd.network.shapeRenderer import ShapeRenderer
    from tensoflow_amd.synth import random_sdf_state
    dev = torch.device("cuda:0")
    R = 64
    r = ShapeRenderer(dict(gridSize=[R, R, R], max_levels=3, sdf_n_comp=36, sdf_dim=256, app_dim=128, predict_BG=False, device="cuda",
                           nerfDataType=True, blend_ratio=0.2), training=False)
    sd = random_sdf_state(seed=1, R=R)
    sd["sdf_mat.2.bias"][0] -= 0.35                                      # the initial blob has radius ~0.05: take its 0.35 level set
    r.load_state_dict({"sdf_network." + k: v for k, v in sd.items()}, strict=False)
    path = str(tmp_path / "shape.ply")
    v, f = extract_mesh(r, resolution=96, path=path)
    assert f.shape[0] > 1000 and np.isfinite(v).all() and np.abs(v).max() <= 1.0
    # every vertex lies on the zero set of the field (to the lattice's interpolation error)
    with torch.no_grad():
        s = r.sdf_network.sdf(torch.from_numpy(v).to(dev), torch.full((v.shape[0],), 0.2, device=dev))[:, 0]
    assert float(s.abs().max()) < 5e-3, float(s.abs().max())
    v2, f2 = read_ply(path)
    assert np.array_equal(v2, v) and np.array_equal(f2, f)
    # closed and consistently wound (apart from the cut by the unit sphere, where the lattice is set to +1 -> still closed)
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    _, dcnt = np.unique(e, axis=0, return_counts=True)
    assert (dcnt == 1).all()
    bvh = ops.Bvh(v2, f2, dev)
    # rays from outside the unit sphere aimed at the most negative point of the field must enter the surface
    from tensoflow_amd.mesh import sdf_lattice
    u = sdf_lattice(lambda x: r.sdf_network.sdf(x, torch.full((x.shape[0],), 0.2, device=dev)), (-1, -1, -1), (1, 1, 1), 96, device=dev)
    assert float(u.min()) < 0
    idx = np.unravel_index(int(u.argmin()), u.shape)
    p_in = torch.tensor([-1 + 2 * i / 95 for i in idx], device=dev)
    g = torch.Generator().manual_seed(0)
    d = torch.nn.functional.normalize(torch.randn(4096, 3, generator=g), dim=-1).to(dev)
    o = (p_in[None] - 2.5 * d).contiguous()
    pos, nrm, depth, hit = bvh.trace(o, d)
    assert bool(hit.all())
    with torch.no_grad():
        s_hit = r.sdf_network.sdf(pos, torch.full((pos.shape[0],), 0.2, device=dev))[:, 0]
    assert float(s_hit.abs().max()) < 5e-3
    # the tracer reports MINUS the face normal (materialRenderer.py:256); with PyMCubes' winding that is the outward normal, so
    # a ray entering the object sees it against its direction
    assert float((nrm * d).sum(-1).max()) < 0
