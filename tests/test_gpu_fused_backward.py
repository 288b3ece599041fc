"""GPU parity and legality of the fused backward pass of rasterize -> interpolate (csrc/backward_fused.hip;
ops.py `_RasterOrigin`): one kernel computes interpolate's and rasterize's gradients, and the operator layer uses its
position gradient only when autograd shows that interpolate was the sole contributor to rast's gradient.

Bars of tests/conftest.py: the fused kernel against the oracle (pinned to the reference) on identical inputs."""
import numpy as np
import pytest
import torch
from conftest import ATOL, grad_tol, within

from nvdiffrast_amd.utils import m10k_batch

# These tests look INSIDE the Python host layer (the records on the tensors, the stand-in gradient, the discard rule): they run
# with that layer serving every call.  tests/test_gpu_host_layer.py asks the same questions of the compiled layer.
pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("python_host_layer")]


def _t(a, dev="cuda"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _kernels(lib, _capi, fn):
    """Names of the library kernels launched while fn() runs."""
    lib.nvdr_profile_reset()
    lib.nvdr_profile_enable(1)
    try:
        fn()
        torch.cuda.synchronize()
        return set(_capi.profile_read())
    finally:
        lib.nvdr_profile_enable(0)
        lib.nvdr_profile_reset()


@pytest.mark.parametrize("A,attr_mode", [(4, "broadcast"), (4, "instance"), (2, "broadcast"), (3, "shared"), (7, "instance"), (1, "broadcast")])
@pytest.mark.parametrize("with_g_rast", [True, False])
def test_fused_kernel_against_the_oracle(dr, oracle, A, attr_mode, with_g_rast):
    from nvdiffrast_amd.torch import _plugin
    N, res = 3, (136, 200)                                   # not multiples of the 64x16 block
    b = m10k_batch(N, seed=31, nx=36, ny=18)
    rng = np.random.default_rng(A)
    V = b["pos"].shape[1]
    attr = rng.uniform(-1, 1, size={"broadcast": (1, V, A), "instance": (N, V, A), "shared": (V, A)}[attr_mode]).astype(np.float32)
    dy = rng.normal(size=(N,) + res + (A,)).astype(np.float32)
    dy[1, 40:60] = 0.0                                       # rows without upstream gradient
    ro, _ = oracle.rasterize(b["pos"], b["tri"], res)
    ga, gr, _ = oracle.interpolate_grad(attr, ro, b["tri"], dy)
    gp = oracle.rasterize_grad(b["pos"], b["tri"], ro, gr)
    g_attr, g_rast, _, g_pos = _plugin.interpolate_rasterize_grad(_t(attr), _t(ro), _t(b["tri"]), _t(b["pos"]), _t(dy), with_g_rast=with_g_rast)
    within("fused: g_attr", g_attr.cpu().numpy(), ga, grad_tol(ga))
    within("fused: g_pos", g_pos.cpu().numpy(), gp, grad_tol(gp))
    if with_g_rast:
        within("fused: g_rast", g_rast.cpu().numpy(), gr, grad_tol(gr))
    else:
        assert g_rast is None
    # and against the two separate kernels of the same library (same per-pixel arithmetic: only the summation differs)
    s_attr, s_rast = _plugin.interpolate_grad(_t(attr), _t(ro), _t(b["tri"]), _t(dy))
    s_pos = _plugin.rasterize_grad(_t(b["pos"]), _t(b["tri"]), _t(ro), s_rast)
    if with_g_rast:
        assert torch.equal(g_rast, s_rast)
    within("fused vs separate: g_pos", g_pos.cpu().numpy(), s_pos.cpu().numpy(), grad_tol(gp))
    within("fused vs separate: g_attr", g_attr.cpu().numpy(), s_attr.cpu().numpy(), grad_tol(ga))


def test_fused_kernel_in_range_mode_and_with_non_finite_gradients(dr, oracle):
    from nvdiffrast_amd.torch import _plugin
    b = m10k_batch(1, seed=32, nx=20, ny=10)
    pos, tri = b["pos"][0], b["tri"]
    ranges = np.array([[0, tri.shape[0]], [30, 200]], np.int32)
    res = (72, 96)
    ro, _ = oracle.rasterize(pos, tri, res, ranges=ranges)
    rng = np.random.default_rng(1)
    attr = rng.uniform(size=(pos.shape[0], 4)).astype(np.float32)
    dy = rng.normal(size=(2,) + res + (4,)).astype(np.float32)
    ga, gr, _ = oracle.interpolate_grad(attr, ro, tri, dy)
    gp = oracle.rasterize_grad(pos, tri, ro, gr)
    g_attr, g_rast, _, g_pos = _plugin.interpolate_rasterize_grad(_t(attr), _t(ro), _t(tri), _t(pos), _t(dy))
    within("fused range mode: g_attr", g_attr.cpu().numpy(), ga, grad_tol(ga))
    within("fused range mode: g_pos", g_pos.cpu().numpy(), gp, grad_tol(gp))
    within("fused range mode: g_rast", g_rast.cpu().numpy(), gr, grad_tol(gr))
    # an infinite upstream gradient: the block falls back to plain f32 atomics; finite elsewhere, inf/nan where the
    # separate kernels put them
    covered = np.argwhere(ro[0, ..., 3] > 0)
    y, x = covered[len(covered) // 2]
    dy2 = dy.copy(); dy2[0, y, x, 1] = np.inf
    f_attr, f_rast, _, f_pos = _plugin.interpolate_rasterize_grad(_t(attr), _t(ro), _t(tri), _t(pos), _t(dy2))
    s_attr, s_rast = _plugin.interpolate_grad(_t(attr), _t(ro), _t(tri), _t(dy2))
    s_pos = _plugin.rasterize_grad(_t(pos), _t(tri), _t(ro), s_rast)
    assert torch.equal(torch.isfinite(f_attr), torch.isfinite(s_attr)) and torch.equal(torch.isfinite(f_pos), torch.isfinite(s_pos))
    assert not torch.isfinite(f_attr).all()
    fin = torch.isfinite(s_pos)
    assert torch.allclose(f_pos[fin], s_pos[fin], rtol=1e-4, atol=1e-4 * float(s_pos[fin].abs().max()))


def test_operator_layer_uses_the_fused_gradient_only_when_it_is_legal(dr, oracle):
    """(1) rasterize -> interpolate with nothing else on rast: one backward kernel, no k_raster_grad
    launch, gradients equal to the oracle's; (2) a second consumer of rast (a coverage mask, as the reference's samples
    build): autograd sums two gradients for rast, the prepared position gradient is discarded, k_raster_grad runs on the
    summed gradient and the result is still the oracle's; (3) after that the context no longer prepares anything;
    (4) an interpolation with another index buffer (pose.py style) is never fused."""
    from nvdiffrast_amd import _capi
    from nvdiffrast_amd.torch import _plugin
    lib = _capi.load()
    N, res = 2, (128, 128)
    b = m10k_batch(N, seed=33, nx=30, ny=15)
    rng = np.random.default_rng(2)
    G = rng.normal(size=(N,) + res + (4,)).astype(np.float32)
    tri = _t(b["tri"])
    ro, _ = oracle.rasterize(b["pos"], b["tri"], res)
    ga, gr, _ = oracle.interpolate_grad(b["attr"], ro, b["tri"], G)
    gp = oracle.rasterize_grad(b["pos"], b["tri"], ro, gr)

    def step(ctx, extra=None, tri_i=None):
        pos = _t(b["pos"]).requires_grad_(True)
        attr = _t(b["attr"]).requires_grad_(True)
        rast, _ = dr.rasterize(ctx, pos, tri, res)
        out, _ = dr.interpolate(attr, rast, tri if tri_i is None else tri_i)
        loss = (out * _t(G)).sum()
        if extra is not None:
            loss = loss + extra(rast)
        loss.backward()
        return pos.grad, attr.grad

    # (1) legal: fused
    ctx = dr.RasterizeCudaContext()
    before = _plugin.fused_backward_count()
    got = {}
    names = _kernels(lib, _capi, lambda: got.update(zip(("pos", "attr"), step(ctx))))
    assert "interp_raster_grad" in names and "raster_grad" not in names and "interp_grad" not in names, names
    assert _plugin.fused_backward_count()["used"] == before["used"] + 1
    within("fused autograd: g_pos", got["pos"].cpu().numpy(), gp, grad_tol(gp))
    within("fused autograd: g_attr", got["attr"].cpu().numpy(), ga, grad_tol(ga))

    # (2) a second contributor to rast's gradient: d(sum(u * w))/d rast on top of interpolate's
    wmask = rng.normal(size=(N,) + res).astype(np.float32)
    extra = lambda rast: (rast[..., 0] * _t(wmask)).sum()                                 # noqa: E731
    gr2 = gr.copy(); gr2[..., 0] += wmask
    gp2 = oracle.rasterize_grad(b["pos"], b["tri"], ro, gr2)
    assert np.abs(gp2 - gp).max() > 1e-3 * np.abs(gp).max()                               # the extra term matters
    before = _plugin.fused_backward_count()
    names = _kernels(lib, _capi, lambda: got.update(zip(("pos", "attr"), step(ctx, extra))))
    assert "interp_raster_grad" in names and "raster_grad" in names, names                 # prepared, discarded, recomputed
    assert _plugin.fused_backward_count()["discarded"] == before["discarded"] + 1
    assert _plugin.fused_backward_count()["materialized"] == before["materialized"] + 1   # autograd's sum looked at the unwritten g_rast
    within("discarded fused: g_pos", got["pos"].cpu().numpy(), gp2, grad_tol(gp2))
    within("discarded fused: g_attr", got["attr"].cpu().numpy(), ga, grad_tol(ga))

    # (3) the context has learnt: the separate kernels from now on, same results
    names = _kernels(lib, _capi, lambda: got.update(zip(("pos", "attr"), step(ctx, extra))))
    assert "interp_raster_grad" not in names and {"interp_grad", "raster_grad"} <= names, names
    within("after discard: g_pos", got["pos"].cpu().numpy(), gp2, grad_tol(gp2))

    # (3b) set_fused_backward("auto") re-arms the contexts that had given up, and a gradient that was prepared but never
    # collected -- the engine was asked for attr's gradient only -- neither blocks the next step nor stays alive
    _plugin.set_fused_backward("auto")
    names = _kernels(lib, _capi, lambda: got.update(zip(("pos", "attr"), step(ctx))))
    assert "interp_raster_grad" in names and "raster_grad" not in names, names
    pos_s = _t(b["pos"]).requires_grad_(True)
    attr_s = _t(b["attr"]).requires_grad_(True)
    rast_s, _ = dr.rasterize(ctx, pos_s, tri, res)
    out_s, _ = dr.interpolate(attr_s, rast_s, tri)
    (ga_only,) = torch.autograd.grad((out_s * _t(G)).sum(), [attr_s], retain_graph=True)        # the rasterize node never runs
    assert rast_s._nvdr_origin.pending is not None
    before = _plugin.fused_backward_count()
    (out_s * _t(G)).sum().backward()                                                             # a full backward over the same graph
    assert rast_s._nvdr_origin.pending is None and _plugin.fused_backward_count()["used"] == before["used"] + 1
    within("after a stale prepared gradient: g_pos", pos_s.grad.cpu().numpy(), gp, grad_tol(gp))
    within("after a stale prepared gradient: g_attr", ga_only.cpu().numpy(), ga, grad_tol(ga))

    # (3c) the fused kernel does not write g_rast: autograd carries a stand-in that computes it when somebody looks (ops._LazyGrad).
    # Nobody looked in (1); the summation in (2) did; so do retain_grad(), a hook, and autograd.grad(..., inputs=[rast]) -- and what
    # they see is the reference's g_rast, while the position gradient still comes from the fused kernel where that is legal.
    def graph():
        pos = _t(b["pos"]).requires_grad_(True)
        attr = _t(b["attr"]).requires_grad_(True)
        rast, _ = dr.rasterize(ctx, pos, tri, res)
        out, _ = dr.interpolate(attr, rast, tri)
        return pos, attr, rast, (out * _t(G)).sum()
    before = _plugin.fused_backward_count()
    pos_r, attr_r, rast_r, loss = graph()
    rast_r.retain_grad()
    loss.backward()
    after = _plugin.fused_backward_count()
    assert after["used"] == before["used"] + 1 and after["materialized"] == before["materialized"] + 1
    within("retain_grad on rast: g_rast", rast_r.grad.cpu().numpy(), gr, grad_tol(gr))
    within("retain_grad on rast: g_pos", pos_r.grad.cpu().numpy(), gp, grad_tol(gp))
    pos_r, attr_r, rast_r, loss = graph()
    g_rast_user, g_pos_user = torch.autograd.grad(loss, [rast_r, pos_r])
    assert _plugin.fused_backward_count()["used"] == after["used"] + 1
    within("autograd.grad w.r.t. rast: g_rast", (g_rast_user * 1.0).cpu().numpy(), gr, grad_tol(gr))
    within("autograd.grad w.r.t. rast: g_rast (numpy)", g_rast_user.cpu().numpy(), gr, grad_tol(gr))
    within("autograd.grad w.r.t. rast: g_pos", g_pos_user.cpu().numpy(), gp, grad_tol(gp))
    pos_r, attr_r, rast_r, loss = graph()
    seen = []
    rast_r.register_hook(lambda g: seen.append(g.detach().clone()) or g * 2.0)                 # a hook that looks AND replaces
    before = _plugin.fused_backward_count()
    loss.backward()
    assert _plugin.fused_backward_count()["discarded"] == before["discarded"] + 1              # twice the gradient: not the prepared one
    within("hook on rast: g_rast seen", seen[0].cpu().numpy(), gr, grad_tol(gr))
    within("hook on rast: g_pos", pos_r.grad.cpu().numpy(), 2.0 * gp, grad_tol(2.0 * gp))
    _plugin.set_fused_backward("auto")                                                          # (the hook step made the context give up)
    # (3d) ADVICE r4: a hook that edits the gradient IN PLACE leaves the stand-in object what it was -- the prepared position
    # gradient must not be used then; and a deferred gradient first looked at after one of its inputs changed must raise
    pos_r, attr_r, rast_r, loss = graph()
    def edit_in_place(g):
        g.mul_(2.0)                                                                              # (returns None: the object autograd carries on is the same stand-in)
    rast_r.register_hook(edit_in_place)
    before = _plugin.fused_backward_count()
    loss.backward()
    assert _plugin.fused_backward_count()["discarded"] == before["discarded"] + 1
    within("in-place hook on rast: g_pos", pos_r.grad.cpu().numpy(), 2.0 * gp, grad_tol(2.0 * gp))
    _plugin.set_fused_backward("auto")
    pos_r, attr_r, rast_r, loss = graph()
    (g_late,) = torch.autograd.grad(loss, [rast_r])
    with torch.no_grad():
        attr_r.add_(1.0)                                                                         # (an optimizer step)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        g_late.cpu()
    _plugin.set_fused_backward("auto")

    # (4) another index buffer for the attributes: not this path's graph
    ctx2 = dr.RasterizeCudaContext()
    tri_b = _t(b["tri"].copy())
    names = _kernels(lib, _capi, lambda: got.update(zip(("pos", "attr"), step(ctx2, None, tri_b))))
    assert "interp_raster_grad" not in names and {"interp_grad", "raster_grad"} <= names, names
    within("other index buffer: g_pos", got["pos"].cpu().numpy(), gp, grad_tol(gp))

    # (5) switched off
    _plugin.set_fused_backward("off")
    try:
        names = _kernels(lib, _capi, lambda: got.update(zip(("pos", "attr"), step(dr.RasterizeCudaContext()))))
        assert "interp_raster_grad" not in names and {"interp_grad", "raster_grad"} <= names, names
    finally:
        _plugin.set_fused_backward("auto")


def test_fused_backward_inside_a_captured_graph(dr, oracle):
    """The identity test of the operator layer is host logic: it is evaluated once at capture time and the replayed
    kernels are the fused one."""
    N, res = 2, (96, 96)
    b = m10k_batch(N, seed=34, nx=24, ny=12)
    rng = np.random.default_rng(3)
    G = _t(rng.normal(size=(N,) + res + (4,)).astype(np.float32))
    tri = _t(b["tri"])
    pos = _t(b["pos"]).requires_grad_(True)
    attr = _t(b["attr"]).requires_grad_(True)
    ctx = dr.RasterizeCudaContext()

    def step():
        pos.grad = None; attr.grad = None
        rast, _ = dr.rasterize(ctx, pos, tri, res)
        out, _ = dr.interpolate(attr, rast, tri)
        torch.autograd.backward(out, G)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    eager_pos, eager_attr = pos.grad.clone(), attr.grad.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    ro, _ = oracle.rasterize(b["pos"], b["tri"], res)
    ga, gr, _ = oracle.interpolate_grad(b["attr"], ro, b["tri"], G.cpu().numpy())
    gp = oracle.rasterize_grad(b["pos"], b["tri"], ro, gr)
    within("fused in a graph: g_pos", pos.grad.cpu().numpy(), gp, grad_tol(gp))
    within("fused in a graph: g_attr", attr.grad.cpu().numpy(), ga, grad_tol(ga))
    assert torch.allclose(pos.grad, eager_pos, rtol=1e-5, atol=1e-5 * float(eager_pos.abs().max()))
    assert torch.allclose(attr.grad, eager_attr, rtol=1e-5, atol=1e-5 * float(eager_attr.abs().max()))


@pytest.mark.parametrize("A,diff,db_to_pos", [(2, "all", True), (3, [0, -1], True), (4, [2], True), (2, "all", False), (5, "all", True)])
def test_fused_kernel_with_pixel_differentials_against_the_oracle(dr, oracle, A, diff, db_to_pos):
    """interpolate_grad_da + rasterize_grad_db in one kernel (config 3's backward pair): attribute gradients through the
    values AND the differentials, g_rast, g_rast_db, and the position gradient with (or, for grad_db=False, without) the
    rast_db term."""
    from nvdiffrast_amd.torch import _plugin
    N, res = 2, (120, 168)
    b = m10k_batch(N, seed=35, nx=30, ny=15)
    rng = np.random.default_rng(10 + A)
    V = b["pos"].shape[1]
    attr = rng.uniform(-1, 1, size=(1, V, A)).astype(np.float32)
    D = A if diff == "all" else len(diff)
    dy = rng.normal(size=(N,) + res + (A,)).astype(np.float32)
    dda = rng.normal(size=(N,) + res + (2 * D,)).astype(np.float32)
    ro, rdbo = oracle.rasterize(b["pos"], b["tri"], res)
    ga, gr, grdb = oracle.interpolate_grad(attr, ro, b["tri"], dy, rdbo, dda, diff)
    gp = oracle.rasterize_grad(b["pos"], b["tri"], ro, gr, grdb if db_to_pos else None)
    g_attr, g_rast, g_rast_db, g_pos = _plugin.interpolate_rasterize_grad(
        _t(attr), _t(ro), _t(b["tri"]), _t(b["pos"]), _t(dy), rast_db=_t(rdbo), dda=_t(dda),
        diff_attrs_all=(diff == "all"), diff_attrs_vec=([] if diff == "all" else diff), db_to_pos=db_to_pos)
    within("fused da: g_attr", g_attr.cpu().numpy(), ga, grad_tol(ga))
    within("fused da: g_rast", g_rast.cpu().numpy(), gr, grad_tol(gr))
    within("fused da: g_rast_db", g_rast_db.cpu().numpy(), grdb, grad_tol(grdb))
    within("fused da: g_pos", g_pos.cpu().numpy(), gp, grad_tol(gp))


def test_operator_layer_fuses_the_differential_pair_too(dr, oracle):
    """rasterize -> interpolate(diff_attrs='all') -> (something per pixel) -> loss: one backward kernel for the pair, unless
    rast_db's gradient has another contributor."""
    from nvdiffrast_amd import _capi
    lib = _capi.load()
    N, res = 2, (128, 128)
    b = m10k_batch(N, seed=36, nx=30, ny=15, attrs=2)
    rng = np.random.default_rng(4)
    G = rng.normal(size=(N,) + res + (2,)).astype(np.float32)
    Gd = rng.normal(size=(N,) + res + (4,)).astype(np.float32)
    tri = _t(b["tri"])
    ro, rdbo = oracle.rasterize(b["pos"], b["tri"], res)
    ga, gr, grdb = oracle.interpolate_grad(b["uv"], ro, b["tri"], G, rdbo, Gd, "all")
    gp = oracle.rasterize_grad(b["pos"], b["tri"], ro, gr, grdb)
    got = {}

    def step(ctx, extra=None, grad_db=True):
        pos = _t(b["pos"]).requires_grad_(True)
        uv = _t(b["uv"]).requires_grad_(True)
        rast, rast_db = dr.rasterize(ctx, pos, tri, res, grad_db=grad_db)
        out, out_da = dr.interpolate(uv, rast, tri, rast_db=rast_db, diff_attrs="all")
        loss = (out * _t(G)).sum() + (out_da * _t(Gd)).sum()
        if extra is not None:
            loss = loss + extra(rast_db)
        loss.backward()
        got["pos"], got["uv"] = pos.grad, uv.grad

    ctx = dr.RasterizeCudaContext()
    names = _kernels(lib, _capi, lambda: step(ctx))
    assert "interp_raster_grad_da" in names and not ({"raster_grad_db", "interp_grad_da"} & names), names
    within("fused da autograd: g_pos", got["pos"].cpu().numpy(), gp, grad_tol(gp))
    within("fused da autograd: g_uv", got["uv"].cpu().numpy(), ga, grad_tol(ga))
    # grad_db=False: the differentials' gradient stops at rast_db
    gp_nodb = oracle.rasterize_grad(b["pos"], b["tri"], ro, gr)
    names = _kernels(lib, _capi, lambda: step(dr.RasterizeCudaContext(), grad_db=False))
    assert "interp_raster_grad_da" in names and "raster_grad" not in names, names
    within("fused da, grad_db=False: g_pos", got["pos"].cpu().numpy(), gp_nodb, grad_tol(gp_nodb))
    # a second contributor to rast_db's gradient: discarded, recomputed from the sum
    w = rng.normal(size=(N,) + res + (4,)).astype(np.float32)
    gp2 = oracle.rasterize_grad(b["pos"], b["tri"], ro, gr, grdb + w)
    names = _kernels(lib, _capi, lambda: step(ctx, lambda rdb: (rdb * _t(w)).sum()))
    assert "interp_raster_grad_da" in names and "raster_grad_db" in names, names
    within("discarded fused da: g_pos", got["pos"].cpu().numpy(), gp2, grad_tol(gp2))
