"""GPU parity against THE REFERENCE ITSELF, without the oracle in between: the HIP path (through the C ABI) is
compared with oracle/_ref -- the reference's own glue, kernels and CudaRaster compiled for the host
(oracle/ref.py) -- on the headline op graph and on the config-3 chain.  This is where BASELINE.json's
"grad max-abs-err vs ref" is measured.  The prebuilt oracle/_ref travels to the GPU box with the snapshot.

Also here: behaviours added in round 2 whose definition is the reference's (cube corners for texture slices
>= 1, the opt-in corner fix) or the advisor's (hipGraphs over mixed layouts, very wide vertices)."""
import numpy as np
import pytest
from conftest import CHAIN_OPS, CHAIN_VALUE_TOL, grad_tol, within
import torch

from nvdiffrast_amd.utils import m10k_batch

pytestmark = pytest.mark.gpu
ATOL = 1e-5


def _t(a, dev="cuda"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _tol(x):
    return ATOL * max(1.0, float(np.abs(x).max()))


def test_headline_chain_against_the_reference(dr, ref):
    """rasterize + interpolate fwd + bwd, the metric's op graph, 3 x 256^2 of the benchmark mesh:
    ids bit-exact, barycentrics / attributes 1e-5 abs, gradients 1e-5 of their magnitude."""
    N, res = 3, (256, 256)
    b = m10k_batch(N, seed=21)
    rng = np.random.default_rng(0)
    G = rng.normal(size=(N,) + res + (4,)).astype(np.float32)
    pos = _t(b["pos"]).requires_grad_(True)
    attr = _t(b["attr"]).requires_grad_(True)
    tri = _t(b["tri"])
    ctx = dr.RasterizeCudaContext()
    rast, rast_db = dr.rasterize(ctx, pos, tri, res)
    out, _ = dr.interpolate(attr, rast, tri)
    torch.autograd.backward(out, _t(G))

    r, rdb = ref.rasterize(b["pos"], b["tri"], res)
    o, _ = ref.interpolate(b["attr"], r, b["tri"])
    g_attr, g_rast, _ = ref.interpolate_grad(b["attr"], r, b["tri"], G)
    g_pos = ref.rasterize_grad(b["pos"], b["tri"], r, g_rast)
    h = rast.detach().cpu().numpy()
    assert (h[..., 3] != r[..., 3]).sum() == 0, "triangle ids differ from the reference"
    assert np.abs(h[..., :3] - r[..., :3]).max() <= ATOL
    assert np.abs(rast_db.detach().cpu().numpy() - rdb).max() <= _tol(rdb)
    assert np.abs(out.detach().cpu().numpy() - o).max() <= ATOL
    err_attr = np.abs(attr.grad.cpu().numpy() - g_attr).max()
    err_pos = np.abs(pos.grad.cpu().numpy() - g_pos).max()
    print("grad max-abs-err vs reference: g_attr %.3g (|g| %.3g), g_pos %.3g (|g| %.3g)" % (err_attr, np.abs(g_attr).max(), err_pos, np.abs(g_pos).max()))
    assert err_attr <= _tol(g_attr) and err_pos <= _tol(g_pos)


def test_four_op_chain_against_the_reference(dr, ref):
    rng = np.random.default_rng(17)
    N, res = 2, (160, 160)
    b = m10k_batch(N, seed=23)
    tex_np = rng.uniform(size=(1, 256, 256, 3)).astype(np.float32)
    pos = _t(b["pos"]).requires_grad_(True)
    tex = _t(tex_np).requires_grad_(True)
    tri = _t(b["tri"])
    ctx = dr.RasterizeCudaContext()
    rast, rdb = dr.rasterize(ctx, pos, tri, res)
    uv, uvda = dr.interpolate(_t(b["uv"]), rast, tri, rast_db=rdb, diff_attrs="all")
    col = dr.texture(tex, uv, uvda, filter_mode="linear-mipmap-linear")
    aa = dr.antialias(col, rast, pos, tri)
    dy = rng.normal(size=tuple(aa.shape)).astype(np.float32)
    aa.backward(_t(dy))

    r, rdb_r = ref.rasterize(b["pos"], b["tri"], res)
    uv_r, uvda_r = ref.interpolate(b["uv"], r, b["tri"], rdb_r, "all")
    col_r = ref.texture(tex_np, uv_r, uvda_r, filter_mode="linear-mipmap-linear")
    aa_r = ref.antialias(col_r, r, b["pos"], b["tri"])
    g_col, g_pos_aa = ref.antialias_grad(col_r, r, b["pos"], b["tri"], dy)
    g = ref.texture_grad(tex_np, uv_r, g_col, uvda_r, filter_mode="linear-mipmap-linear")
    _ga, g_rast, g_rdb = ref.interpolate_grad(b["uv"], r, b["tri"], g["uv"], rdb_r, g["uv_da"], "all")
    g_pos = ref.rasterize_grad(b["pos"], b["tri"], r, g_rast, g_rdb) + g_pos_aa

    assert (rast.detach().cpu().numpy()[..., 3] != r[..., 3]).sum() == 0
    # every element, no exemptions; the bars of a four-op chain compared end to end (tests/conftest.py)
    within("chain vs ref: col", col.detach().cpu().numpy(), col_r, CHAIN_VALUE_TOL)
    within("chain vs ref: aa", aa.detach().cpu().numpy(), aa_r, CHAIN_VALUE_TOL)
    within("chain vs ref: g_tex", tex.grad.cpu().numpy(), g["tex"], grad_tol(g["tex"]))
    within("chain vs ref: g_pos", pos.grad.cpu().numpy(), g_pos, grad_tol(g_pos, CHAIN_OPS))


def test_four_op_chain_at_config3_scale_against_the_reference(dr, ref, raw_oracle):
    """BASELINE configs[2] at ITS OWN size: one item at 1024^2 of the 10k-triangle benchmark mesh with a
    2048^2 mipmapped texture, the four ops forward and backward, against the reference itself (about 5 s of CPU).  Every op
    is compared ON THE INPUTS THE HIP PATH GAVE IT, with the single-op bars: a chain through a 2048^2 white-noise texture is
    not defined end to end (oracle/chain.py: a few hundred of the million pixels lie within an ulp of a texel boundary,
    where the uv gradient of bilinear sampling jumps) -- the end-to-end differences are printed, not asserted.
    Gradients that are sums over very many pixels (the texture gradient of the four texels every background pixel hits:
    6e5 terms each) are held to 1e-5 against the oracle, which sums in f64, and to 2e-5 against the reference, whose own
    f32 atomic sum in launch order carries a rounding noise of that size (the bracket oracle/pinned.py uses to pin one
    to the other)."""
    from oracle.chain import four_op_chain
    rng = np.random.default_rng(5)
    res = (1024, 1024)
    b = m10k_batch(1, seed=20240, attrs=2)
    assert b["tri"].shape[0] == 10000
    tex_np = rng.uniform(size=(1, 2048, 2048, 3)).astype(np.float32)
    dy = _t(rng.normal(size=(1,) + res + (3,)).astype(np.float32))
    ctx = dr.RasterizeCudaContext()
    for chk, who, summed in ((ref, "ref", 2.0), (raw_oracle, "oracle", 1.0)):
        e = four_op_chain(dr, ctx, None, chk, b["pos"], b["tri"], b["uv"], tex_np, dy, res, end_to_end=(who == "ref"))
        assert e["tri_id_mismatches"] == 0 and e["coverage"] > 0.1
        if who == "ref":
            print("c3-scale chain, end to end (conditioning, not parity):", e["end_to_end"])
        name = "c3 chain vs %s, op by op: " % who
        for k in ("bary_max_abs", "uv", "col", "aa"):
            within(name + k, e[k + "_err"], 0.0, ATOL)
        for k in ("rast_db", "uv_da", "g_col", "g_uv", "g_uv_da", "g_rast", "g_rast_db"):            # per-pixel results
            within(name + k, e[k + "_err"], 0.0, ATOL * max(1.0, e[k + "_max"]))
        for k in ("g_tex", "g_uvattr"):                                                              # sums over many pixels
            within(name + k, e[k + "_err"], 0.0, summed * ATOL * max(1.0, e[k + "_max"]))
        within(name + "g_pos", e["g_pos_err"], 0.0, 2 * summed * ATOL * max(1.0, e["g_pos_max"]))   # two ops' sums added


@pytest.mark.parametrize("fix", [False, True])
def test_cube_corner_texels_for_texture_slices_above_zero(dr, oracle, fix):
    """texture_kernel.cu:85-88,431-432: for slices >= 1 the reference loses the corner flag and samples texel (0,0)
    of face 5 of the previous slice.  Default = that behaviour (the pinned oracle checks it against the reference
    itself); set_cube_corner_fix(True) keeps the corner average for every slice."""
    from nvdiffrast_amd.torch import _plugin
    rng = np.random.default_rng(77)
    N, H, W, C = 2, 40, 40, 3
    tex = rng.uniform(size=(2, 6, 4, 4, C)).astype(np.float32)
    v = rng.normal(size=(N, H, W, 3)).astype(np.float32)
    v = (np.sign(v) * rng.uniform(0.9, 1.0, size=v.shape)).astype(np.float32)        # all near cube corners
    dy = rng.normal(size=(N, H, W, C)).astype(np.float32)
    _plugin.set_cube_corner_fix(fix)
    oracle.set_cube_corner_fix(fix)
    pin = oracle.enabled
    if fix:
        oracle.enabled = False                       # the fix is not reference behaviour: compare with the raw oracle
    try:
        for fm in ("linear", "linear-mipmap-linear"):
            da = (rng.normal(size=(N, H, W, 6)) * 0.05).astype(np.float32) if "mipmap" in fm else None
            t_tex = _t(tex).requires_grad_(True)
            out = dr.texture(t_tex, _t(v), None if da is None else _t(da), filter_mode=fm, boundary_mode="cube")
            out.backward(_t(dy))
            oo = oracle.texture(tex, v, da, filter_mode=fm, boundary_mode="cube")
            g = oracle.texture_grad(tex, v, dy, da, filter_mode=fm, boundary_mode="cube")
            # every element (an earlier version exempted 0.3 % of the corner pixels; the margins printed by `within` showed none used it)
            within("cube corners %s fix=%d: out" % (fm, fix), out.detach().cpu().numpy(), oo, ATOL)
            within("cube corners %s fix=%d: g_tex" % (fm, fix), t_tex.grad.cpu().numpy(), g["tex"], grad_tol(g["tex"]))
    finally:
        _plugin.set_cube_corner_fix(False)
        oracle.set_cube_corner_fix(False)
        oracle.enabled = pin
    if not fix:
        # the two behaviours do differ on this input (the test would be vacuous otherwise)
        oracle.set_cube_corner_fix(True)
        try:
            alt = oracle._o.texture(tex, v, filter_mode="linear", boundary_mode="cube")
        finally:
            oracle.set_cube_corner_fix(False)
        assert np.abs(alt[1] - oracle._o.texture(tex, v, filter_mode="linear", boundary_mode="cube")[1]).max() > 1e-3


def test_interpolate_backward_with_very_wide_vertices(dr, oracle):
    """A = 300 attributes per vertex: no LDS vertex table fits, the gradient kernel runs on plain atomics."""
    rng = np.random.default_rng(5)
    A = 300
    b = m10k_batch(1, seed=9, nx=10, ny=6)
    attr = rng.uniform(size=(1, b["pos"].shape[1], A)).astype(np.float32)
    ro, _ = oracle.rasterize(b["pos"], b["tri"], (48, 64))
    dy = rng.normal(size=(1, 48, 64, A)).astype(np.float32)
    t_attr = _t(attr).requires_grad_(True)
    t_rast = _t(ro).requires_grad_(True)
    out, _ = dr.interpolate(t_attr, t_rast, _t(b["tri"]))
    out.backward(_t(dy))
    oo, _ = oracle.interpolate(attr, ro, b["tri"])
    g_attr, g_rast, _ = oracle.interpolate_grad(attr, ro, b["tri"], dy)
    assert np.abs(out.detach().cpu().numpy() - oo).max() <= ATOL
    assert np.abs(t_attr.grad.cpu().numpy() - g_attr).max() <= _tol(g_attr)
    assert np.abs(t_rast.grad.cpu().numpy()[..., :2] - g_rast[..., :2]).max() <= _tol(g_rast)


def test_graphs_and_eager_calls_with_mixed_layouts_on_one_context(dr, oracle):
    """a hipGraph freezes the scratch pointer and the scratch_clean flag.  Two graphs with different
    layouts and eager calls with a third, interleaved on ONE context (the third outgrows the scratch buffer the
    graphs point to), must all keep producing the right ids."""
    ctx = dr.RasterizeCudaContext()
    scenes = []
    for seed, N, res, kw in ((1, 2, (64, 64), dict(nx=16, ny=8)), (2, 1, (96, 80), dict(nx=24, ny=12)), (3, 4, (200, 200), dict(nx=60, ny=30))):
        b = m10k_batch(N, seed=seed, **kw)
        scenes.append((b, res, oracle.rasterize(b["pos"], b["tri"], res)[0]))
    static = []
    for b, res, _ro in scenes[:2]:
        pos, tri = _t(b["pos"]), _t(b["tri"])
        dr.rasterize(ctx, pos, tri, res)                      # warm-up outside capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            r, _ = dr.rasterize(ctx, pos, tri, res)
        static.append((g, r, pos, tri))                       # (a graph's inputs must outlive it: replays read these very buffers)
    b3, res3, ro3 = scenes[2]
    pos3, tri3 = _t(b3["pos"]), _t(b3["tri"])
    for _round in range(3):
        for k in (0, 1):
            static[k][0].replay()
            torch.cuda.synchronize()
            assert (static[k][1].cpu().numpy()[..., 3] != scenes[k][2][..., 3]).sum() == 0
            r3, _ = dr.rasterize(ctx, pos3, tri3, res3)       # eager, other layout, larger scratch
            assert (r3.cpu().numpy()[..., 3] != ro3[..., 3]).sum() == 0


def test_growing_clip_pool_gives_the_same_image(dr, oracle, capfd):
    """NVDR_OPT_SCRATCH_LIMIT_MB = 0 forces the growing-pool scratch policy (what meshes of millions of triangles get);
    a scene in which every triangle crosses frustum planes overflows the initial pool, the glue reads the demand back,
    grows the pool and repeats the call: ids must equal the oracle's (and the reference's) like in worst-case mode."""
    from nvdiffrast_amd import _capi
    from nvdiffrast_amd.torch import _plugin
    rng = np.random.default_rng(12)
    T = 9000
    pos = rng.normal(size=(2, 3 * T, 4)).astype(np.float32) * np.array([3.0, 3.0, 1.5, 1.0], np.float32)
    pos[..., 3] = rng.uniform(0.05, 1.5, size=pos.shape[:2])
    tri = np.arange(3 * T, dtype=np.int32).reshape(T, 3)
    ro, _ = oracle.rasterize(pos, tri, (96, 128))
    lib = _capi.load()
    old = lib.nvdr_get_option(_capi.OPT_SCRATCH_LIMIT_MB)
    lib.nvdr_set_option(_capi.OPT_SCRATCH_LIMIT_MB, 0)
    _plugin.set_log_level(0)
    try:
        ctx = dr.RasterizeCudaContext()
        ctx.cpp_wrapper.set_pool_hint(2, T, 64)                  # start far too small
        r, _ = dr.rasterize(ctx, _t(pos), _t(tri), (96, 128))
        assert (r.cpu().numpy()[..., 3] != ro[..., 3]).sum() == 0
        grown = ctx.cpp_wrapper.pool_slots(2, T)
        assert 64 < grown <= 6 * T
        assert "Clip pool grown" in capfd.readouterr().err
        r2, _ = dr.rasterize(ctx, _t(pos), _t(tri), (96, 128))    # second call: the remembered size fits at once
        assert torch.equal(r, r2) and ctx.cpp_wrapper.pool_slots(2, T) == grown
        assert "Clip pool grown" not in capfd.readouterr().err
        t_pos, t_tri = _t(pos), _t(tri)
        with pytest.raises(RuntimeError, match="cannot be captured"):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                dr.rasterize(ctx, t_pos, t_tri, (96, 128))
    finally:
        lib.nvdr_set_option(_capi.OPT_SCRATCH_LIMIT_MB, old)
        _plugin.set_log_level(1)


def test_growing_pool_mode_with_the_worst_case_pool_does_not_read_garbage(dr, oracle):
    """in growing-pool mode the glue read the demand counter back even when the pool was already the
    clipper's worst case (meshes of <= 4096 triangles start there; grow_pool caps there) -- a counter the library neither
    clears nor writes in that case, so a stale value above the pool made the retry loop spin for ever.  The buffer is
    poisoned first so that a read of the counter would see such a value."""
    from nvdiffrast_amd import _capi
    from nvdiffrast_amd.torch import _plugin
    rng = np.random.default_rng(13)
    T = 500
    pos = rng.normal(size=(3, 3 * T, 4)).astype(np.float32) * np.array([3.0, 3.0, 1.5, 1.0], np.float32)
    pos[..., 3] = rng.uniform(0.05, 1.5, size=pos.shape[:2])
    tri = np.arange(3 * T, dtype=np.int32).reshape(T, 3)
    ro, _ = oracle.rasterize(pos, tri, (64, 64))
    lib = _capi.load()
    old = lib.nvdr_get_option(_capi.OPT_SCRATCH_LIMIT_MB)
    lib.nvdr_set_option(_capi.OPT_SCRATCH_LIMIT_MB, 0)
    try:
        ctx = dr.RasterizeCudaContext()
        st = ctx.cpp_wrapper
        assert st.pool_hint(3, T) == 6 * T
        nbytes = st.scratch_bytes(lib, 3, T, 64, 64, 6 * T)
        st.scratch = torch.full((nbytes,), 0x7F, dtype=torch.uint8, device="cuda")      # every int reads 0x7F7F7F7F
        if _plugin.host_layer() is not None:
            st.host_state(_plugin.host_layer()).set_scratch(torch.full((nbytes,), 0x7F, dtype=torch.uint8, device="cuda"))
        for _ in range(2):
            r, _ = dr.rasterize(ctx, _t(pos), _t(tri), (64, 64))
            assert (r.cpu().numpy()[..., 3] != ro[..., 3]).sum() == 0
        # a pool that has grown to the cap behaves the same
        T2 = 6000
        pos2 = rng.normal(size=(1, 3 * T2, 4)).astype(np.float32) * np.array([3.0, 3.0, 1.5, 1.0], np.float32)
        pos2[..., 3] = rng.uniform(0.05, 1.5, size=pos2.shape[:2])
        tri2 = np.arange(3 * T2, dtype=np.int32).reshape(T2, 3)
        st.set_pool_hint(1, T2, 6 * T2)
        ro2, _ = oracle.rasterize(pos2, tri2, (64, 64))
        r2, _ = dr.rasterize(ctx, _t(pos2), _t(tri2), (64, 64))
        assert (r2.cpu().numpy()[..., 3] != ro2[..., 3]).sum() == 0
    finally:
        lib.nvdr_set_option(_capi.OPT_SCRATCH_LIMIT_MB, old)
