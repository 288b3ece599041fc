#!/usr/bin/env python3
"""Records WHAT THE REFERENCE'S OWN ops.py ASKS OF ITS EXTENSION MODULE -- tests/golden/reference_ops_transcript.{json,npz}.

The reference's Python layer (/root/reference/nvdiffrast/torch/ops.py, unmodified) is executed on the reference's own
C++/CUDA sources compiled for the host (oracle/_ref) with a tracing stub between the two: every call that ops.py makes
into `_nvdiffrast_c` -- function name, argument order, shapes / dtypes of tensor arguments, values of the scalar ones,
which earlier result each tensor is, the opaque wrappers handed around, and the shapes / dtypes / VALUES of everything
that came back -- is written down.  tests/test_gpu_reference_ops.py replays the transcript call by call against
`nvdiffrast_amd.torch._plugin` on the GPU box, where the reference checkout does not exist: same calls, same argument
structure, results within the parity bars of what the reference returned.  (Where the checkout exists the same file also
runs the reference's ops.py itself on the plugin.)

Scenes: those of make_reference_fixture.py (config-3 chain, headline chain, cube map with two slices, three peeling
layers, range mode) plus the entry points they do not reach: plain / nearest / mipmap-nearest texture filters with
their gradient functions, a prebuilt mip wrapper, a custom mip stack, a prebuilt topology hash, diff_attrs lists,
set_log_level / get_log_level.

    python tests/golden/make_ops_transcript.py          (needs /root/reference; run in the build container)
"""
import hashlib
import importlib.util
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

CPU_ARGS = {("rasterize_fwd_cuda", 4)}            # `ranges` stays a CPU tensor (torch_rasterize.cpp:49)
SKIP_VALUES = {("antialias_fwd", 1)}              # the work buffer: item order depends on atomics, private to fwd -> grad


def _key(a):
    a = np.ascontiguousarray(a)
    return (a.shape, str(a.dtype), hashlib.sha1(a.tobytes()).hexdigest())


class Tracer:
    """Stands between the reference's ops.py and a plugin module, writing down the conversation."""

    def __init__(self, inner):
        self.inner = inner
        self.calls = []
        self.arrays = {}              # npz name -> ndarray (literal inputs and returned values)
        self.by_content = {}          # content key -> tensor reference ("lit:3" / "h:12")
        self.obj_ids = {}             # id(python wrapper object) -> small integer
        self.keep = []                # keeps wrapper objects alive so that id() stays unique
        self.n_lit = self.n_h = 0
        self.TextureMipWrapper = inner.TextureMipWrapper
        self.TopologyHashWrapper = inner.TopologyHashWrapper

    # ---- argument / result description ---------------------------------------------------------------------------
    def _tensor_arg(self, t, fn, idx):
        if t.numel() == 0 and t.dim() == 1:
            return {"k": "empty"}                                   # ops.py's placeholder torch.tensor([]) (ops.py:301-305)
        a = t.detach().cpu().numpy()
        key = _key(a)
        ref = self.by_content.get(key)
        if ref is None:
            ref = "lit:%d" % self.n_lit
            self.arrays["lit_%d" % self.n_lit] = np.ascontiguousarray(a)
            self.n_lit += 1
            self.by_content[key] = ref
        return {"k": "tensor", "ref": ref, "shape": list(a.shape), "dtype": str(a.dtype),
                "device": "cpu" if (fn, idx) in CPU_ARGS else "dev"}

    def _obj(self, o, create):
        if id(o) not in self.obj_ids:
            if not create:
                # an object ops.py made itself, e.g. the empty TextureMipWrapper() placeholder (ops.py:306-307)
                self.obj_ids[id(o)] = len(self.obj_ids)
                self.keep.append(o)
                return {"k": "new", "cls": type(o).__name__, "id": self.obj_ids[id(o)]}
            self.obj_ids[id(o)] = len(self.obj_ids)
            self.keep.append(o)
        return {"k": "obj", "cls": type(o).__name__, "id": self.obj_ids[id(o)]}

    def _arg(self, v, fn, idx):
        if isinstance(v, torch.Tensor):
            return self._tensor_arg(v, fn, idx)
        if v is None or isinstance(v, (bool, int, float, str)):
            return {"k": "val", "v": v}
        if isinstance(v, (list, tuple)):
            if all(isinstance(x, (bool, int, float)) for x in v):
                return {"k": "val", "v": list(v)}
            return {"k": "list", "items": [self._arg(x, fn, idx) for x in v]}
        return self._obj(v, create=False)

    def _ret(self, v, fn, idx):
        if isinstance(v, torch.Tensor):
            a = np.ascontiguousarray(v.detach().cpu().numpy())
            ref = "h:%d" % self.n_h
            self.n_h += 1
            check = (fn, idx) not in SKIP_VALUES
            if check:
                self.arrays["h_%d" % (self.n_h - 1)] = a
            self.by_content.setdefault(_key(a), ref)
            return {"k": "tensor", "ref": ref, "shape": list(a.shape), "dtype": str(a.dtype), "check": check}
        if isinstance(v, (list, tuple)):
            return {"k": "list", "items": [self._ret(x, fn, idx) for x in v]}
        if v is None or isinstance(v, (bool, int, float, str)):
            return {"k": "val", "v": v}
        return self._obj(v, create=True)

    # ---- the module surface ops.py sees --------------------------------------------------------------------------
    def __getattr__(self, name):
        target = getattr(self.inner, name)

        def call(*args):
            rec = {"fn": name, "args": [self._arg(a, name, i) for i, a in enumerate(args)]}
            out = target(*args)
            outs = out if isinstance(out, tuple) else (out,)
            rec["ret"] = [self._ret(o, name, i) for i, o in enumerate(outs)]
            rec["tuple"] = isinstance(out, tuple)
            self.calls.append(rec)
            return out
        return call


def extra_scenes(dr, i, dev="cpu"):
    """Entry points that make_reference_fixture.run() does not reach."""
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)      # noqa: E731
    rng = np.random.default_rng(81)
    ctx = dr.RasterizeCudaContext()
    tri = T(i["tri"])
    pos = T(i["pos"]).requires_grad_(True)
    rast, rast_db = dr.rasterize(ctx, pos, tri, (40, 56), grad_db=False)
    uvattr = T(i["uv"]).requires_grad_(True)
    uv, uv_da = dr.interpolate(uvattr, rast, tri, rast_db=rast_db, diff_attrs=[1, -2])      # list form, negative index
    (uv.sum() + uv_da.sum()).backward()
    uv = uv.detach(); uv_da_all = dr.interpolate(T(i["uv"]), rast.detach(), tri, rast_db=rast_db.detach(), diff_attrs="all")[1].detach()
    tex_np = i["tex"]
    g = T(rng.normal(size=(2, 40, 56, 3)).astype(np.float32))
    for fm, bm in (("nearest", "clamp"), ("linear", "zero"), ("linear-mipmap-nearest", "wrap")):
        tex = T(tex_np).requires_grad_(True)
        u = uv.clone().requires_grad_(True)
        kw = dict(filter_mode=fm, boundary_mode=bm)
        if "mipmap" in fm:
            out = dr.texture(tex, u, uv_da_all, **kw)
        else:
            out = dr.texture(tex, u, **kw)
        (out * g).sum().backward()
    # prebuilt mip wrapper with a level limit, mip_level_bias only
    tex = T(tex_np).requires_grad_(True)
    mipw = dr.texture_construct_mip(tex.detach(), max_mip_level=3)
    bias = T(rng.uniform(0.0, 3.0, size=(2, 40, 56)).astype(np.float32)).requires_grad_(True)
    out = dr.texture(tex, uv, mip_level_bias=bias, mip=mipw, filter_mode="linear-mipmap-linear", max_mip_level=3)
    (out * g).sum().backward()
    # custom mip stack: the levels receive their own gradients
    tex = T(tex_np).requires_grad_(True)
    levels = [T(rng.uniform(size=(1, 32 >> k, 32 >> k, 3)).astype(np.float32)).requires_grad_(True) for k in (1, 2)]
    out = dr.texture(tex, uv, uv_da_all, mip=levels, filter_mode="linear-mipmap-linear")
    (out * g).sum().backward()
    # prebuilt topology hash, gradient boost (the boost is applied by ops.py, not by the plugin)
    topo = dr.antialias_construct_topology_hash(tri)
    col = T(rng.uniform(size=(2, 40, 56, 3)).astype(np.float32)).requires_grad_(True)
    pos2 = T(i["pos"]).requires_grad_(True)
    aa = dr.antialias(col, rast.detach(), pos2, tri, topology_hash=topo, pos_gradient_boost=2.0)
    (aa * g).sum().backward()
    lvl = dr.get_log_level()
    dr.set_log_level(2)
    assert dr.get_log_level() == 2
    dr.set_log_level(lvl)


def main():
    from oracle import ref, ref_torch
    ref.build()
    spec = importlib.util.spec_from_file_location("make_reference_fixture", os.path.join(HERE, "make_reference_fixture.py"))
    fixture = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fixture)
    tracer = Tracer(ref_torch.cpu_plugin("fma"))
    dr = ref_torch.reference_on_cpu("fma")
    dr._nvdiffrast_c = tracer                                          # the stub goes between ops.py and the plugin
    i = fixture.inputs()
    o = fixture.run(dr, i)
    n_fixture_calls = len(tracer.calls)
    extra_scenes(dr, i)
    # the fixture's named results as handles (or sums of two handles: autograd adds the position gradients of
    # antialias and rasterize) -- informative, the replay checks every handle anyway
    named = {}
    handles = {k[2:]: v for k, v in tracer.arrays.items() if k.startswith("h_")}
    for name, val in o.items():
        hit = [h for h, a in handles.items() if a.shape == val.shape and np.array_equal(a, val)]
        if hit:
            named[name] = ["h:" + hit[0]]
            continue
        same = [(h, a) for h, a in handles.items() if a.shape == val.shape and a.dtype == val.dtype]
        for x in range(len(same)):
            for y in range(x + 1, len(same)):
                if np.array_equal(same[x][1] + same[y][1], val) or np.array_equal(same[y][1] + same[x][1], val):
                    named[name] = ["h:" + same[x][0], "h:" + same[y][0]]
    doc = {"what": "calls of the reference's nvdiffrast/torch/ops.py into _nvdiffrast_c, recorded by tests/golden/make_ops_transcript.py",
           "reference_root": ref.lib().nvdr_ref_reference_root().decode(),
           "calls_from_fixture_scenes": n_fixture_calls, "calls": tracer.calls, "fixture_outputs": named}
    jpath = os.path.join(HERE, "reference_ops_transcript.json")
    with open(jpath, "w") as f:
        json.dump(doc, f, indent=0, separators=(",", ":"))
    npath = os.path.join(HERE, "reference_ops_transcript.npz")
    np.savez_compressed(npath, **tracer.arrays)
    fns = sorted({c["fn"] for c in tracer.calls})
    print("%d calls (%d from the fixture scenes), %d distinct entry points: %s" % (len(tracer.calls), n_fixture_calls, len(fns), ", ".join(fns)))
    print("fixture outputs located:", {k: v for k, v in named.items()}, "missing:", sorted(set(o) - set(named)))
    print(jpath, os.path.getsize(jpath), npath, os.path.getsize(npath))


if __name__ == "__main__":
    main()
