"""CPU tests around tests/golden/reference_pipeline.npz -- vectors produced by THE REFERENCE ITSELF (its ops.py on its
own C++/CUDA sources compiled for the host; tests/golden/make_reference_fixture.py).

 * the committed vectors are what the reference produces today (regenerated here when the checkout is present);
 * BASELINE config 1: samples/torch/triangle.py's inputs through the reference's own Python layer on the CPU build
   reproduce docs/img/tri.png;
 * the C oracle reproduces the reference's vectors (so the fixture pins the oracle even where oracle/_ref is not
   available, e.g. in a checkout without /root/reference)."""
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "reference_pipeline.npz")
RES = (40, 56)


def _gen():
    spec = importlib.util.spec_from_file_location("make_reference_fixture", os.path.join(HERE, "golden", "make_reference_fixture.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def fx():
    z = np.load(FIXTURE)
    return {k[3:]: z[k] for k in z.files if k.startswith("in_")}, {k[4:]: z[k] for k in z.files if k.startswith("out_")}


def _tol(x, r=2e-5):
    return r * max(1.0, float(np.abs(x).max()))


def test_reference_stack_reproduces_the_committed_vectors(ref, fx):
    from oracle import ref_torch
    if not ref_torch.reference_ops_available():
        pytest.skip("reference ops.py absent")
    gen = _gen()
    i, o = fx
    assert all(np.array_equal(i[k], v) for k, v in gen.inputs().items())
    new = gen.run(ref_torch.reference_on_cpu("fma"), i)
    assert set(new) == set(o)
    for k, v in new.items():
        assert np.array_equal(v, o[k]), k                    # single-threaded, fixed launch order: bit-reproducible


def test_triangle_sample_through_the_reference_python_layer_on_cpu(ref):
    """BASELINE config 1 ("samples/torch/triangle.py ... on reference CPU/GL context"): the reference's ops.py driving
    the reference's rasterizer on the host -- samples/torch/triangle.py:19-30 -- equals docs/img/tri.png."""
    import torch
    from PIL import Image
    from oracle import ref_torch
    if not ref_torch.reference_ops_available():
        pytest.skip("reference ops.py absent")
    dr = ref_torch.reference_on_cpu()
    pos = torch.tensor([[[-0.8, -0.8, 0, 1], [0.8, -0.8, 0, 1], [-0.8, 0.8, 0, 1]]], dtype=torch.float32)
    col = torch.tensor([[[1, 0, 0], [0, 1, 0], [0, 0, 1]]], dtype=torch.float32)
    tri = torch.tensor([[0, 1, 2]], dtype=torch.int32)
    glctx = dr.RasterizeCudaContext()
    rast, _ = dr.rasterize(glctx, pos, tri, resolution=[256, 256])
    out, _ = dr.interpolate(col, rast, tri)
    img = np.clip(np.rint(out.numpy()[0, ::-1, :, :] * 255), 0, 255).astype(np.uint8)
    assert (img != np.array(Image.open(os.path.join(HERE, "golden", "tri.png")))).sum() == 0
    # the reference's quirks survive: rasterize() during peeling RETURNS an exception object (ops.py:131-132)
    with dr.DepthPeeler(glctx, pos, tri, (8, 8)):
        assert isinstance(dr.rasterize(glctx, pos, tri, (8, 8)), RuntimeError)


def test_oracle_reproduces_the_reference_vectors(raw_oracle, fx):
    o_ = raw_oracle
    i, o = fx
    rast, rast_db = o_.rasterize(i["pos"], i["tri"], RES)
    assert (rast[..., 3] != o["rast"][..., 3]).sum() == 0
    assert np.abs(rast - o["rast"]).max() <= 1e-5 and np.abs(rast_db - o["rast_db"]).max() <= _tol(o["rast_db"], 1e-5)
    uv, uv_da = o_.interpolate(i["uv"], rast, i["tri"], rast_db=rast_db, diff_attrs="all")
    col = o_.texture(i["tex"], uv, uv_da, filter_mode="linear-mipmap-linear")
    out = o_.antialias(col, rast, i["pos"], i["tri"])
    assert np.abs(uv - o["uv"]).max() <= 1e-5 and np.abs(uv_da - o["uv_da"]).max() <= _tol(o["uv_da"], 1e-5)
    assert np.abs(col - o["col"]).max() <= 1e-5 and np.abs(out - o["out"]).max() <= 1e-5
    g_col, g_pos_aa = o_.antialias_grad(col, rast, i["pos"], i["tri"], i["g_out"])
    gt = o_.texture_grad(i["tex"], uv, g_col, uv_da, filter_mode="linear-mipmap-linear")
    g_uvattr, g_rast, g_rast_db = o_.interpolate_grad(i["uv"], rast, i["tri"], gt["uv"], rast_db=rast_db, dda=gt["uv_da"], diff_attrs="all")
    g_pos = g_pos_aa + o_.rasterize_grad(i["pos"], i["tri"], rast, g_rast, g_rast_db)
    assert np.abs(gt["tex"] - o["g_tex"]).max() <= _tol(o["g_tex"])
    assert np.abs(g_uvattr - o["g_uvattr"]).max() <= _tol(o["g_uvattr"])
    assert np.abs(g_pos - o["g_pos"]).max() <= _tol(o["g_pos"])
    # headline chain
    a2, _ = o_.interpolate(i["attr"], rast, i["tri"])
    g_attr, g_rast2, _ = o_.interpolate_grad(i["attr"], rast, i["tri"], i["g_attr_out"])
    assert np.abs(a2 - o["h_out"]).max() <= 1e-5
    assert np.abs(g_attr - o["h_g_attr"]).max() <= _tol(o["h_g_attr"])
    assert np.abs(o_.rasterize_grad(i["pos"], i["tri"], rast, g_rast2) - o["h_g_pos"]).max() <= _tol(o["h_g_pos"])
    # cube map with two slices
    kw = dict(filter_mode="linear-mipmap-linear", boundary_mode="cube")
    assert np.abs(o_.texture(i["cube_tex"], i["cube_dir"], i["cube_da"], **kw) - o["cube_out"]).max() <= 1e-5
    g = o_.texture_grad(i["cube_tex"], i["cube_dir"], i["g_cube"], i["cube_da"], **kw)
    assert np.abs(g["tex"] - o["cube_g_tex"]).max() <= _tol(o["cube_g_tex"])
    assert np.abs(g["uv"] - o["cube_g_dir"]).max() <= _tol(o["cube_g_dir"])
    assert np.abs(g["uv_da"] - o["cube_g_da"]).max() <= _tol(o["cube_g_da"])
    # depth peeling, range mode
    peel = None
    for k in range(3):
        r, _, depth = o_.rasterize(i["peel_pos"], i["peel_tri"], (48, 48), peel_depth=peel, return_depth=True)
        peel = depth
        assert (r[..., 3] != o["peel%d" % k][..., 3]).sum() == 0
    r, rdb = o_.rasterize(i["pos"][0], i["tri"], RES, ranges=i["ranges"])
    assert (r[..., 3] != o["range_rast"][..., 3]).sum() == 0 and np.abs(r - o["range_rast"]).max() <= 1e-5


def test_ops_transcript_replays_exactly_on_the_reference_itself(ref):
    """tests/golden/reference_ops_transcript.{json,npz} (the calls of the reference's ops.py into _nvdiffrast_c, recorded
    with the reference's results) replayed against the reference's own code on the CPU: the replay engine that the GPU
    test uses against the HIP plugin must reproduce every recorded tensor bit for bit here."""
    import numpy as np
    from oracle import ref_torch
    from replay_ops import replay
    n = [0]

    def on_tensor(call, idx, got, want, chained):
        n[0] += 1
        assert np.array_equal(got, want), (call["fn"], idx)

    doc, _ = replay(ref_torch.cpu_plugin("fma"), "cpu", on_tensor)
    assert n[0] >= 60 and doc["calls_from_fixture_scenes"] == 22


@pytest.mark.parametrize("name", ["t1m", "t1m_shuffled", "s10k_1024"])
def test_oracle_equals_the_reference_at_a_million_triangles(name, raw_oracle):
    """tests/golden/t1m_reference.npz holds what the REFERENCE's rasterizer (oracle/_ref, tests/golden/make_t1m_fixture.py) makes
    of item 0 of bench.py's million-triangle scenes and of an S10k stress item at 1024^2: SHA-256 of the id image, 256 sampled
    (u, v, z/w, id).  The C oracle must reproduce it -- which pins the checker of bench.py's t1m blocks to the reference at that
    size (VERDICT r5 "missing" 4); tests/test_gpu_full_size.py asks the same of the HIP path."""
    spec = importlib.util.spec_from_file_location("make_t1m_fixture", os.path.join(HERE, "golden", "make_t1m_fixture.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    fx = np.load(os.path.join(HERE, "golden", "t1m_reference.npz"))
    (pos, tri, res), = [(p, t, r) for n, p, t, r in mk.scenes() if n == name]
    rast, _ = raw_oracle.rasterize(pos, tri, res)
    assert mk.digest(rast) == bytes(fx[name + "/ids_sha256"]).decode()
    assert int((rast[0, ..., 3] > 0).sum()) == int(fx[name + "/covered"])
    yx, want = fx[name + "/sample_yx"], fx[name + "/sample_rast"]
    got = rast[0, yx[:, 0], yx[:, 1]]
    assert (got[:, 3] != want[:, 3]).sum() == 0 and np.abs(got[:, :3] - want[:, :3]).max() <= 1e-5
