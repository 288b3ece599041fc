"""Host logic of the plugin-level fused backward (`_plugin.py`, section "fused backward") on CPU tensors: who gets the prepared
position gradient, who voids it.  No kernel runs here -- the launches are replaced by a recorder; the GPU counterpart with real
gradients is tests/test_gpu_plugin_fused_backward.py."""
import gc
import weakref

import pytest
import torch

from nvdiffrast_amd.torch import _plugin, ops


class _State:
    fused_disabled = -1


def _scene():
    pos = torch.zeros(1, 5, 4, requires_grad=True)
    tri = torch.zeros(3, 3, dtype=torch.int32)
    rast = torch.zeros(1, 8, 8, 4)
    rast_db = torch.zeros(1, 8, 8, 4)
    org = _plugin._FwdOrigin(pos, tri, _State(), rast, rast_db)
    return pos, tri, rast, rast_db, org


def _prepare(org, rast, rast_db=None):
    """What _fused_interpolate_grad leaves behind, without the kernel."""
    src = _plugin._LazySource(lambda: (torch.ones_like(rast), torch.full_like(rast, 2.0)), ())
    g_rast = _plugin._LazyGrad(rast, src, 0)
    g_rast._origin = org
    g_db = None
    if rast_db is not None:
        g_db = _plugin._LazyGrad(rast_db, src, 1)
        g_db._origin = org
    g_pos = torch.zeros(1, 5, 4)
    org.pending = (weakref.ref(g_rast), g_pos, None if g_db is None else weakref.ref(g_db))
    return g_rast, g_db, g_pos


def _recorder():
    calls = []
    return calls, (lambda dy, ddb, grad: calls.append((dy, ddb, grad)))


def test_lazy_classes_are_shared_with_the_operator_layer():
    assert ops._LazyGrad is _plugin._LazyGrad and ops._LazySource is _plugin._LazySource


def test_origin_accepts_only_the_untouched_rast_of_its_call():
    pos, tri, rast, rast_db, org = _scene()
    attr = torch.zeros(1, 5, 3)
    seen = rast.detach().requires_grad_(True)                # what autograd hands a backward: same storage, same version
    assert not org.usable_by(attr, seen, tri, None)          # never interpolated
    org.interpolations = 1
    assert org.usable_by(attr, seen, tri, None) and org.usable_by(attr, seen, tri, rast_db)
    assert not org.usable_by(torch.zeros(1, 6, 3), seen, tri, None)                      # another vertex set
    assert not org.usable_by(attr, seen, tri.clone(), None)                              # another index buffer
    assert not org.usable_by(attr, rast, tri, None)                                      # rast outside the graph
    assert not org.usable_by(attr, seen, tri, rast_db.clone())                           # not the rast_db of this call
    org.interpolations = 2
    assert not org.usable_by(attr, seen, tri, None)                                      # somebody else interpolates the same rast
    org.interpolations = 1
    rast.add_(1.0)                                                                       # written to since
    assert not org.usable_by(attr, rast.detach().requires_grad_(True), tri, None)
    pos2, tri2, rast2, _, org2 = _scene()
    org2.interpolations = 1
    with torch.no_grad():
        pos2.add_(1.0)                                                                   # pos changed after the forward pass
    assert not org2.usable_by(attr, rast2.detach().requires_grad_(True), tri2, None)
    _plugin.set_fused_backward("off")
    try:
        assert not org.usable_by(attr, seen, tri, None)
    finally:
        _plugin.set_fused_backward("auto")


def test_the_prepared_gradient_goes_to_its_own_unedited_stand_in():
    pos, tri, rast, _, org = _scene()
    g_rast, _, g_pos = _prepare(org, rast)
    calls, call = _recorder()
    before = _plugin.fused_backward_count()
    assert _plugin._take_prepared("t", pos, tri, rast, g_rast, None, call) is g_pos and not calls and org.pending is None
    # a materialised (or real) ddb adds its share to the SAME buffer, with dy absent
    g_rast, _, g_pos = _prepare(org, rast)
    ddb = torch.zeros(1, 8, 8, 4)
    assert _plugin._take_prepared("t", pos, tri, rast, g_rast, ddb, call) is g_pos
    assert len(calls) == 1 and calls[0][0] is None and calls[0][1] is ddb and calls[0][2] is g_pos
    after = _plugin.fused_backward_count()
    assert after["used"] == before["used"] + 2 and after["discarded"] == before["discarded"] and after["materialized"] == before["materialized"]
    # nothing prepared: nothing to take, nothing counted
    assert _plugin._take_prepared("t", pos, tri, rast, torch.zeros(1, 8, 8, 4), None, call) is None
    assert _plugin.fused_backward_count() == after


def test_the_stand_in_carries_the_exchange_when_the_record_is_gone():
    pos, tri, rast, _, org = _scene()
    g_rast, _, g_pos = _prepare(org, rast)
    out_copy = rast.detach()                                  # the saved OUTPUT autograd returns: another object, no record attached
    calls, call = _recorder()
    assert _plugin._record_of(out_copy, "rast") is None
    assert _plugin._take_prepared("t", pos, tri, out_copy, g_rast, None, call) is g_pos


@pytest.mark.parametrize("how", ["edited", "summed", "other rast", "other pos", "stale stand-in"])
def test_what_voids_the_prepared_gradient(how):
    pos, tri, rast, _, org = _scene()
    _plugin._attach_tiles(rast, None, "rast")
    rast._nvdr_tiles.origin = org
    g_rast, _, g_pos = _prepare(org, rast)
    calls, call = _recorder()
    out, dy, p = rast, g_rast, pos
    if how == "edited":
        g_rast.mul_(2.0)                                      # a hook working in place: same object, other values
    elif how == "summed":
        dy = torch.ones(1, 8, 8, 4)                           # autograd summed rast's gradient with somebody else's: an ordinary tensor arrives
    elif how == "other rast":
        out = torch.zeros(1, 8, 8, 4)
    elif how == "other pos":
        p = torch.zeros(1, 5, 4)
    else:
        older = g_rast
        g_rast, _, g_pos = _prepare(org, rast)                # a second backward pass prepared again; the first pass's stand-in arrives
        dy = older
    before = _plugin.fused_backward_count()
    org.state.fused_disabled = -1
    assert _plugin._take_prepared("t", p, tri, out, dy, None, call) is None and not calls
    assert org.pending is None
    assert _plugin.fused_backward_count()["discarded"] == before["discarded"] + 1
    assert org.state.fused_disabled == _plugin.fused_backward_epoch()                    # the context stops preparing ...
    attr = torch.zeros(1, 5, 3)
    org.interpolations = 1
    seen = rast.detach().requires_grad_(True)
    if how != "edited":                                        # (the in-place edit also moved the shared zero's version; not rast's)
        assert not org.usable_by(attr, seen, tri, None)
        _plugin.set_fused_backward("auto")                     # ... until re-armed
        assert org.usable_by(attr, seen, tri, None)


def test_the_differential_pair_needs_both_stand_ins():
    pos, tri, rast, rast_db, org = _scene()
    calls, call = _recorder()
    g_rast, g_db, g_pos = _prepare(org, rast, rast_db)
    assert _plugin._take_prepared("t", pos, tri, rast, g_rast, g_db, call) is g_pos and not calls
    # rasterize ran with grad_db=False: the caller's rasterize_grad passes no ddb, but rast_db's share is inside the prepared gradient
    g_rast, g_db, g_pos = _prepare(org, rast, rast_db)
    org.state.fused_disabled = -1
    assert _plugin._take_prepared("t", pos, tri, rast, g_rast, None, call) is None
    # rast_db's gradient has another contributor: an ordinary tensor arrives for ddb
    g_rast, g_db, g_pos = _prepare(org, rast, rast_db)
    assert _plugin._take_prepared("t", pos, tri, rast, g_rast, torch.ones(1, 8, 8, 4), call) is None and not calls
    _plugin.set_fused_backward("auto")


def test_a_stand_in_nobody_collects_pins_nothing():
    pos, tri, rast, _, org = _scene()
    g_rast, _, g_pos = _prepare(org, rast)
    ref = weakref.ref(g_rast)
    del g_rast
    gc.collect()
    assert ref() is None                                       # `pending` refers to the stand-in weakly
    calls, call = _recorder()
    _plugin._attach_tiles(rast, None, "rast")
    rast._nvdr_tiles.origin = org
    org.state.fused_disabled = -1
    assert _plugin._take_prepared("t", pos, tri, rast, torch.ones(1, 8, 8, 4), None, call) is None      # ... and what arrives instead voids it
    _plugin.set_fused_backward("auto")


def test_looking_at_the_stand_in_computes_the_values_once():
    pos, tri, rast, _, org = _scene()
    n = []
    src = _plugin._LazySource(lambda: (n.append(1) or torch.full_like(rast, 3.0),), (rast,))
    g = _plugin._LazyGrad(rast, src, 0)
    assert g.shape == rast.shape and g.dtype == rast.dtype and g.unedited() and not n
    assert float((g + 1).sum()) == 4.0 * rast.numel() and float(g.sum()) == 3.0 * rast.numel() and len(n) == 1
    # a tensor the thunk reads changed before the first look: autograd's message
    src2 = _plugin._LazySource(lambda: (torch.zeros_like(rast),), (rast,))
    g2 = _plugin._LazyGrad(rast, src2, 0)
    rast.add_(1.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        g2 + 0
