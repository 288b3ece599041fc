"""The compiled host layer (csrc_host/nvdr_torch_host.cpp) on the GPU: it is what serves rasterize() / interpolate() by default,
its results are the Python layer's (same kernels) and the oracle's, and its way of fusing the backward pass -- interpolate's
backward leaves its share of the position gradient with the rasterize node, other consumers of rast are ADDED on top -- gives
the reference's gradients in every autograd situation the Python layer's stand-in gradient was built for.
tests/test_host_layer_logic.py asks the same questions on the CPU against a stub of the C ABI."""
import numpy as np
import pytest
import torch
from conftest import ATOL, grad_tol, within

from nvdiffrast_amd import _capi
from nvdiffrast_amd.torch import _plugin
from nvdiffrast_amd.utils import m10k_batch

pytestmark = pytest.mark.gpu


def _t(a, grad=False):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda").requires_grad_(grad)


def _np(t):
    return t.detach().cpu().numpy()


def _scene(n=3, seed=3, res=(120, 136)):
    b = m10k_batch(n, seed=seed, nx=14, ny=9)
    G = np.random.default_rng(seed).normal(size=(n,) + res + (b["attr"].shape[-1],)).astype(np.float32)
    return b, res, G


def _chain(oracle, b, res, G, extra=None):
    ro, rdbo = oracle.rasterize(b["pos"], b["tri"], res)
    oo, _ = oracle.interpolate(b["attr"], ro, b["tri"])
    ga, gr, _ = oracle.interpolate_grad(b["attr"], ro, b["tri"], G)
    gp = oracle.rasterize_grad(b["pos"], b["tri"], ro, gr if extra is None else gr + extra)
    return ro, rdbo, oo, ga, gr, gp


def test_the_compiled_layer_is_built_loaded_and_the_default(dr):
    assert _capi.host() is not None, "nvdiffrast_amd/_nvdr_host.so is missing on the GPU box (python -m nvdiffrast_amd._build)"
    assert _plugin.host_layer_name() == "compiled"
    b, res, G = _scene()
    ctx = dr.RasterizeCudaContext()
    pos, attr, tri = _t(b["pos"], True), _t(b["attr"], True), _t(b["tri"])
    rast, _ = dr.rasterize(ctx, pos, tri, res)
    out, _ = dr.interpolate(attr, rast, tri)
    assert rast.grad_fn.name() == "NvdrRasterizeBackward" and out.grad_fn.name() == "NvdrInterpolateBackward"


def test_same_results_as_the_python_layer_and_the_oracle(dr, oracle):
    b, res, G = _scene(seed=5)
    ro, rdbo, oo, ga, gr, gp = _chain(oracle, b, res, G)
    got = {}
    for layer in ("compiled", "python"):
        _plugin.set_host_layer(layer)
        try:
            ctx = dr.RasterizeCudaContext()
            pos, attr, tri = _t(b["pos"], True), _t(b["attr"], True), _t(b["tri"])
            c0 = _plugin.fused_backward_count()
            rast, rast_db = dr.rasterize(ctx, pos, tri, res)
            out, _ = dr.interpolate(attr, rast, tri)
            torch.autograd.backward(out, _t(G))
            assert _plugin.fused_backward_count()["used"] == c0["used"] + 1
            got[layer] = [_np(x) for x in (rast, rast_db, out, attr.grad, pos.grad)]
        finally:
            _plugin.set_host_layer("compiled")
    for a, p in zip(got["compiled"][:3], got["python"][:3]):
        assert np.array_equal(a, p)                                      # the same kernels on the same inputs
    r, rdb, o, g_attr, g_pos = got["compiled"]
    assert (r[..., 3] != ro[..., 3]).sum() == 0
    within("host layer: rast", r[..., :3], ro[..., :3], ATOL); within("host layer: rast_db", rdb, rdbo, grad_tol(rdbo))
    within("host layer: out", o, oo, ATOL)
    within("host layer: g_attr", g_attr, ga, grad_tol(ga)); within("host layer: g_pos", g_pos, gp, grad_tol(gp))
    within("compiled vs python: g_pos", g_pos, got["python"][4], grad_tol(gp))


def test_other_consumers_of_rast_are_added_to_the_prepared_share(dr, oracle):
    b, res, G = _scene(seed=7)
    Wm = np.random.default_rng(0).normal(size=(3,) + res + (4,)).astype(np.float32)
    ctx = dr.RasterizeCudaContext()
    pos, attr, tri = _t(b["pos"], True), _t(b["attr"], True), _t(b["tri"])
    c0 = _capi.host().counters()
    rast, _ = dr.rasterize(ctx, pos, tri, res)
    out, _ = dr.interpolate(attr, rast, tri)
    ((out * _t(G)).sum() + (rast * _t(Wm)).sum()).backward()
    _, _, _, ga, gr, gp = _chain(oracle, b, res, G, extra=Wm)
    within("two contributors: g_attr", _np(attr.grad), ga, grad_tol(ga)); within("two contributors: g_pos", _np(pos.grad), gp, 2 * grad_tol(gp))
    c1 = _capi.host().counters()
    assert c1["fused"] == c0["fused"] + 1 and c1["fused_plus"] == c0["fused_plus"] + 1


@pytest.mark.parametrize("how", ["hook", "retain_grad", "autograd_grad"])
def test_whoever_looks_at_rasts_gradient_sees_the_reference_values(dr, oracle, how):
    b, res, G = _scene(seed=11)
    ctx = dr.RasterizeCudaContext()
    pos, attr, tri = _t(b["pos"], True), _t(b["attr"], True), _t(b["tri"])
    rast, _ = dr.rasterize(ctx, pos, tri, res)
    out, _ = dr.interpolate(attr, rast, tri)
    _, _, _, ga, gr, gp = _chain(oracle, b, res, G)
    seen = []
    c0 = _capi.host().counters()
    if how == "hook":
        rast.register_hook(lambda g: seen.append(g.clone()))
        torch.autograd.backward(out, _t(G))
    elif how == "retain_grad":
        rast.retain_grad()
        torch.autograd.backward(out, _t(G))
        seen.append(rast.grad)
    else:
        g_rast, g_pos = torch.autograd.grad(out, [rast, pos], _t(G))
        seen.append(g_rast)
        within("autograd.grad: g_pos", _np(g_pos), gp, grad_tol(gp))
    assert _capi.host().counters()["fused"] == c0["fused"]
    within("g_rast as seen by " + how, _np(seen[0]), gr, grad_tol(gr))
    if how != "autograd_grad":
        within(how + ": g_pos", _np(pos.grad), gp, grad_tol(gp)); within(how + ": g_attr", _np(attr.grad), ga, grad_tol(ga))


def test_pixel_differentials_range_mode_grad_db_false(dr, oracle):
    b, res, G = _scene(n=2, seed=17)
    rng = np.random.default_rng(1)
    pos2, attr2 = b["pos"][0].copy(), b["attr"][0].copy()
    T, A = b["tri"].shape[0], attr2.shape[-1]
    ranges = np.array([[0, T // 2], [T // 2, T - T // 2]], np.int32)
    Gda = rng.normal(size=(2,) + res + (2 * A,)).astype(np.float32)
    ro, rdbo = oracle.rasterize(pos2, b["tri"], res, ranges=ranges)
    ctx = dr.RasterizeCudaContext()
    tri = _t(b["tri"])
    for grad_db in (True, False):
        pos, attr = _t(pos2, True), _t(attr2, True)
        c0 = _capi.host().counters()
        rast, rast_db = dr.rasterize(ctx, pos, tri, res, ranges=torch.from_numpy(ranges), grad_db=grad_db)
        out, out_da = dr.interpolate(attr, rast, tri, rast_db=rast_db, diff_attrs="all")
        torch.autograd.backward([out, out_da], [_t(G), _t(Gda)])
        assert _capi.host().counters()["fused"] == c0["fused"] + 1
        assert (_np(rast)[..., 3] != ro[..., 3]).sum() == 0
        oo, odao = oracle.interpolate(attr2, ro, b["tri"], rast_db=rdbo, diff_attrs="all")
        ga, gr, grdb = oracle.interpolate_grad(attr2, ro, b["tri"], G, rast_db=rdbo, dda=Gda, diff_attrs="all")
        gp = oracle.rasterize_grad(pos2, b["tri"], ro, gr, grdb if grad_db else None)
        within("range mode: out_da", _np(out_da), odao, grad_tol(odao))
        within("range mode: g_attr", _np(attr.grad), ga, grad_tol(ga)); within("range mode: g_pos grad_db=%s" % grad_db, _np(pos.grad), gp, 2 * grad_tol(gp))


def test_two_interpolations_and_antialias_on_one_rast(dr, oracle):
    """Config 3's shape: rast is read by two interpolations and by antialias (which gives rast no gradient); both interpolations
    prepare their share, pos also receives antialias' own gradient through autograd's sum."""
    b, res, G = _scene(seed=19)
    rng = np.random.default_rng(2)
    attr_b = rng.normal(size=b["attr"].shape[:-1] + (3,)).astype(np.float32)
    Gb = rng.normal(size=(3,) + res + (3,)).astype(np.float32)
    ctx = dr.RasterizeCudaContext()
    pos, attr, attr2, tri = _t(b["pos"], True), _t(b["attr"], True), _t(attr_b, True), _t(b["tri"])
    c0 = _capi.host().counters()
    rast, _ = dr.rasterize(ctx, pos, tri, res)
    o1, _ = dr.interpolate(attr, rast, tri)
    o2, _ = dr.interpolate(attr2, rast, tri)
    aa = dr.antialias(o2, rast, pos, tri)
    torch.autograd.backward([o1, aa], [_t(G), _t(Gb)])
    ro, _ = oracle.rasterize(b["pos"], b["tri"], res)
    o2o, _ = oracle.interpolate(attr_b, ro, b["tri"])
    g_col, g_pos_aa = oracle.antialias_grad(o2o, ro, b["pos"], b["tri"], Gb)
    ga1, gr1, _ = oracle.interpolate_grad(b["attr"], ro, b["tri"], G)
    ga2, gr2, _ = oracle.interpolate_grad(attr_b, ro, b["tri"], g_col)
    gp = oracle.rasterize_grad(b["pos"], b["tri"], ro, gr1 + gr2) + g_pos_aa
    within("two interpolations: g_attr 1", _np(attr.grad), ga1, grad_tol(ga1)); within("two interpolations: g_attr 2", _np(attr2.grad), ga2, 2 * grad_tol(ga2))
    within("two interpolations + antialias: g_pos", _np(pos.grad), gp, 3 * grad_tol(gp))
    c1 = _capi.host().counters()
    assert c1["fused"] == c0["fused"] + 2 and c1["fused_alone"] == c0["fused_alone"] + 1


def test_records_and_texture_find_the_flags(dr, oracle):
    b, res, G = _scene(seed=23)
    rng = np.random.default_rng(3)
    V = b["pos"].shape[1]
    uvattr = rng.uniform(0, 1, size=(V, 2)).astype(np.float32)
    tex_np = rng.uniform(size=(1, 64, 64, 3)).astype(np.float32)
    ctx = dr.RasterizeCudaContext()
    pos, uva, tex, tri = _t(b["pos"], True), _t(uvattr, True), _t(tex_np, True), _t(b["tri"])
    rast, rast_db = dr.rasterize(ctx, pos, tri, res)
    f = _plugin.flags_of(rast)
    assert f is not None
    n, (h, w) = rast.shape[0], res
    want = torch.nn.functional.max_pool2d((rast[..., 3] > 0).float()[:, None], 8, ceil_mode=True)[:, 0] > 0
    assert torch.equal(_plugin.tile_flags_grid(f, n, h, w) != 0, want)
    assert _plugin.flags_of(rast.detach()) is not None and _plugin.flags_of(rast.clone()) is None and _plugin.flags_of(rast_db) is None
    uv, uv_da = dr.interpolate(uva, rast, tri, rast_db=rast_db, diff_attrs="all")
    assert _plugin.flags_of(uv, "zero").data_ptr() == f.data_ptr() and _plugin.flags_of(uv_da, "zero").data_ptr() == f.data_ptr()
    col = dr.texture(tex, uv, uv_da, filter_mode="linear-mipmap-linear")
    aa = dr.antialias(col, rast, pos, tri)
    dy = rng.normal(size=aa.shape).astype(np.float32)
    aa.backward(_t(dy))
    # the same with every flag ignored: identical forward values (skipping never changes what is computed)
    _plugin.set_tile_skipping(False)
    try:
        assert _plugin.flags_of(rast) is None
        pos2, uva2, tex2 = _t(b["pos"], True), _t(uvattr, True), _t(tex_np, True)
        r2, rdb2 = dr.rasterize(ctx, pos2, tri, res)
        uv2, uvda2 = dr.interpolate(uva2, r2, tri, rast_db=rdb2, diff_attrs="all")
        col2 = dr.texture(tex2, uv2, uvda2, filter_mode="linear-mipmap-linear")
        aa2 = dr.antialias(col2, r2, pos2, tri)
        aa2.backward(_t(dy))
    finally:
        _plugin.set_tile_skipping(True)
    assert torch.equal(uv, uv2) and torch.equal(uv_da, uvda2)             # zeros are written either way
    within("flags vs none: antialiased colour", _np(aa), _np(aa2), ATOL)  # (a tile of known-zero uv is sampled once, by scalar arithmetic)
    within("flags vs none: g_tex", _np(tex.grad), _np(tex2.grad), grad_tol(_np(tex2.grad)))
    within("flags vs none: g_pos", _np(pos.grad), _np(pos2.grad), grad_tol(_np(pos2.grad)))
    within("flags vs none: g_uvattr", _np(uva.grad), _np(uva2.grad), grad_tol(_np(uva2.grad)))
    with torch.no_grad():
        rast.mul_(1.0)
    assert _plugin.flags_of(rast) is None                                 # version counter moved


def test_depth_peeling_and_errors_still_worded_by_the_python_layer(dr, oracle):
    b, res, G = _scene(seed=41)
    ctx = dr.RasterizeCudaContext()
    pos, tri = _t(b["pos"]), _t(b["tri"])
    depth = None
    with dr.DepthPeeler(ctx, pos, tri, res) as peeler:
        for k in range(3):
            rast, _ = peeler.rasterize_next_layer()
            want, _, depth = oracle.rasterize(b["pos"], b["tri"], res, peel_depth=depth, return_depth=True)
            assert (_np(rast)[..., 3] != want[..., 3]).sum() == 0, "layer %d" % k
    with pytest.raises(RuntimeError, match="must reside on the same GPU device"):
        dr.rasterize(ctx, pos.cpu(), tri, res)
    with pytest.raises(RuntimeError, match="must be float32 tensors"):
        dr.rasterize(ctx, pos.double(), tri, res)
    rast, _ = dr.rasterize(ctx, pos, tri, res)
    with pytest.raises(RuntimeError, match="must be contiguous tensors"):
        dr.interpolate(_t(b["attr"])[..., :2], rast, tri)


def test_the_step_captures_into_a_graph_and_replays(dr, oracle):
    b, res, G = _scene(seed=43)
    ctx = dr.RasterizeCudaContext()
    pos, attr, tri, Gt = _t(b["pos"], True), _t(b["attr"], True), _t(b["tri"]), _t(G)

    def step():
        rast, _ = dr.rasterize(ctx, pos, tri, res)
        out, _ = dr.interpolate(attr, rast, tri)
        torch.autograd.backward(out, Gt)
        return out

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            pos.grad = attr.grad = None
            step()
    torch.cuda.current_stream().wait_stream(s)
    pos.grad = attr.grad = None
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
    with torch.no_grad():
        pos.mul_(0.97)                                                   # new inputs in the captured tensors
        pos.grad.zero_(); attr.grad.zero_()
    g.replay()
    torch.cuda.synchronize()
    b2 = dict(b, pos=_np(pos))
    _, _, oo, ga, _, gp = _chain(oracle, b2, res, G)
    within("graph replay: out", _np(out), oo, ATOL)
    within("graph replay: g_attr", _np(attr.grad), ga, grad_tol(ga)); within("graph replay: g_pos", _np(pos.grad), gp, grad_tol(gp))
    assert ctx.cpp_wrapper.host_state(_capi.host()).captured


@pytest.mark.parametrize("seed", range(24))
def test_random_graphs_compiled_layer_equals_python_layer(dr, seed):
    """Property test across the two host layers: a random op graph over one rasterization -- plain / differentiated / listed
    interpolations (one or two of them), texture, antialias, a user-made consumer of rast, hooks, grad_db on or off, backward() or
    autograd.grad() over a random subset of the inputs -- gives bit-identical forward values and gradients equal to the
    summation-order bar on both layers (the Python layer is the one the earlier rounds' parity tests pinned to the reference)."""
    rng = np.random.default_rng(4200 + seed)
    n = int(rng.integers(1, 4))
    res = (int(rng.integers(5, 20)) * 8 + int(rng.integers(0, 8)), int(rng.integers(5, 20)) * 8 + int(rng.integers(0, 8)))
    b = m10k_batch(n, seed=int(rng.integers(1, 1000)), nx=int(rng.integers(4, 20)), ny=int(rng.integers(3, 12)))
    V = b["pos"].shape[1]
    A = int(rng.integers(1, 6))
    attr_np = rng.normal(size=(V, A) if rng.uniform() < 0.5 else (n, V, A)).astype(np.float32)
    uv_np = rng.uniform(0, 1, size=(V, 2)).astype(np.float32)
    tex_np = rng.uniform(size=(1, 32, 64, 3)).astype(np.float32)
    opt = dict(grad_db=bool(rng.uniform() < 0.7), da=str(rng.choice(["none", "all", "list"])), second=bool(rng.uniform() < 0.4),
               texture=bool(rng.uniform() < 0.6), antialias=bool(rng.uniform() < 0.6), mask=bool(rng.uniform() < 0.4),
               hook=str(rng.choice(["none", "none", "rast", "out"])), mode=str(rng.choice(["backward", "backward", "grad_subset"])))
    diff = None if opt["da"] == "none" else "all" if opt["da"] == "all" else [int(x) for x in rng.integers(-A, A, size=int(rng.integers(1, 4)))]
    G = {}

    def upstream(name, t):
        if name not in G:
            G[name] = rng.normal(size=tuple(t.shape)).astype(np.float32)
        return _t(G[name])

    def run(layer):
        _plugin.set_host_layer(layer)
        try:
            ctx = dr.RasterizeCudaContext()
            pos, attr, uva, tex, tri = _t(b["pos"], True), _t(attr_np, True), _t(uv_np, True), _t(tex_np, True), _t(b["tri"])
            rast, rast_db = dr.rasterize(ctx, pos, tri, res, grad_db=opt["grad_db"])
            fwd, outs = [rast, rast_db], []
            o, oda = dr.interpolate(attr, rast, tri, rast_db=rast_db if diff is not None else None, diff_attrs=diff)
            fwd += [o, oda]
            outs.append(("o", o))
            if oda.numel():
                outs.append(("oda", oda))
            if opt["hook"] == "rast":
                rast.register_hook(lambda g: g * 1.0)
            if opt["hook"] == "out":
                o.register_hook(lambda g: g * 0.5)
            if opt["second"] or opt["texture"]:
                uv, uvda = dr.interpolate(uva, rast, tri, rast_db=rast_db, diff_attrs="all")
                fwd += [uv, uvda]
                if opt["texture"]:
                    col = dr.texture(tex, uv, uvda, filter_mode="linear-mipmap-linear")
                    fwd.append(col)
                    if opt["antialias"]:
                        col = dr.antialias(col, rast, pos, tri)
                        fwd.append(col)
                    outs.append(("col", col))
                else:
                    outs += [("uv", uv), ("uvda", uvda)]
            elif opt["antialias"] and A >= 1:
                aa = dr.antialias(o, rast, pos, tri)
                fwd.append(aa)
                outs.append(("aa", aa))
            if opt["mask"]:
                outs.append(("mask", rast[..., :2] * rast_db[..., 1:3]))
            tensors = [t for _, t in outs]
            grads = [upstream(k, t) for k, t in outs]
            leaves = [pos, attr, uva, tex]
            if opt["mode"] == "backward":
                torch.autograd.backward(tensors, grads)
                got = [x.grad for x in leaves]
            else:
                pick = [x for k, x in enumerate(leaves) if asked[k]]
                it = iter(torch.autograd.grad(tensors, pick, grads, allow_unused=True))
                got = [next(it) if asked[k] else None for k in range(len(leaves))]
            torch.cuda.synchronize()
            return [_np(f) for f in fwd], [None if g is None else _np(g) for g in got], (pos, attr, uva, tex)
        finally:
            _plugin.set_host_layer("compiled")

    asked = [True] + [bool(rng.uniform() < 0.5) for _ in range(3)]             # which of (pos, attr, uva, tex) autograd.grad asks for
    f_c, g_c, _ = run("compiled")
    f_p, g_p, _ = run("python")
    assert len(f_c) == len(f_p)
    for k, (a, p) in enumerate(zip(f_c, f_p)):
        if opt["antialias"] and k == len(f_c) - 1:
            # (a pixel blended from two sides receives two f32 atomics: their order is the hardware's, run to run)
            within("random graph: antialiased image", a, p, ATOL)
            continue
        assert np.array_equal(a, p, equal_nan=True), (opt, "forward tensor %d" % k, float(np.nanmax(np.abs(a - p))), int((a != p).sum()))
    for k, (a, p) in enumerate(zip(g_c, g_p)):
        assert (a is None) == (p is None), (opt, k)
        if a is not None:
            within("random graph %s leaf %d" % (opt["mode"], k), a, p, 2 * grad_tol(p))


def test_a_long_loop_leaks_nothing(dr):
    """2000 steps on one context: allocated memory returns to where it was (records of dead tensors are swept, nodes die with
    their graphs) and a dropped graph that was never run backward leaves nothing behind either."""
    b, res, G = _scene(seed=51)
    ctx = dr.RasterizeCudaContext()
    pos, attr, tri, Gt = _t(b["pos"], True), _t(b["attr"], True), _t(b["tri"]), _t(G)

    def step(backward=True):
        pos.grad = attr.grad = None
        rast, rast_db = dr.rasterize(ctx, pos, tri, res)
        out, _ = dr.interpolate(attr, rast, tri, rast_db=rast_db, diff_attrs=[0])
        if backward:
            torch.autograd.backward(out, Gt)

    for _ in range(20):
        step()
    pos.grad = attr.grad = None
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    for k in range(2000):
        step(backward=k % 7 != 3)
    pos.grad = attr.grad = None
    torch.cuda.synchronize()
    # (the registry may hold the flags of the last forward call's tensors until the next attach: a few KB)
    assert torch.cuda.memory_allocated() - base <= 1 << 20, (torch.cuda.memory_allocated(), base)


def test_two_threads_two_contexts(dr, oracle):
    """The compiled layer releases the GIL around its work: two Python threads, each with its own context and stream, render and
    differentiate different scenes concurrently and both get the oracle's results (the record registry is shared between them)."""
    import threading
    scenes = [_scene(seed=61), _scene(n=2, seed=62, res=(96, 160))]
    want = [_chain(oracle, b, res, G) for b, res, G in scenes]
    out = [None, None]
    err = []

    def work(k):
        try:
            b, res, G = scenes[k]
            with torch.cuda.stream(torch.cuda.Stream()):
                ctx = dr.RasterizeCudaContext()
                tri, Gt = _t(b["tri"]), _t(G)
                torch.cuda.current_stream().synchronize()
                for _ in range(60):
                    pos, attr = _t(b["pos"], True), _t(b["attr"], True)
                    rast, _ = dr.rasterize(ctx, pos, tri, res)
                    o, _ = dr.interpolate(attr, rast, tri)
                    torch.autograd.backward(o, Gt)
                torch.cuda.current_stream().synchronize()
                out[k] = (_np(rast), _np(o), _np(attr.grad), _np(pos.grad))
        except Exception as e:                                   # noqa: BLE001
            err.append(repr(e))

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not err, err
    for k in range(2):
        ro, _, oo, ga, _, gp = want[k]
        r, o, g_attr, g_pos = out[k]
        assert (r[..., 3] != ro[..., 3]).sum() == 0
        within("threads: out", o, oo, ATOL); within("threads: g_attr", g_attr, ga, grad_tol(ga)); within("threads: g_pos", g_pos, gp, grad_tol(gp))
