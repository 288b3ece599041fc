"""Replays tests/golden/reference_ops_transcript.{json,npz} -- the calls the reference's own ops.py makes into its
extension module `_nvdiffrast_c`, recorded by tests/golden/make_ops_transcript.py -- against a plugin module.

`replay(plugin, device, on_tensor)` issues every recorded call with the recorded argument structure (tensors resolved to
the literal inputs or to the REPLAY'S OWN earlier results, wrappers to the objects the replay got back), checks that what
comes back has the recorded arity, shapes and dtypes, and hands every returned tensor to `on_tensor(call, index, got,
want, chained)` together with the value the reference returned."""
import json
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    with open(os.path.join(HERE, "golden", "reference_ops_transcript.json")) as f:
        doc = json.load(f)
    return doc, np.load(os.path.join(HERE, "golden", "reference_ops_transcript.npz"))


def replay(plugin, device, on_tensor):
    doc, arrays = load()
    handles, objs, lits = {}, {}, {}

    def arg(a):
        k = a["k"]
        if k == "val":
            return a["v"], False
        if k == "empty":
            return torch.tensor([]), False
        if k == "list":
            items = [arg(x) for x in a["items"]]
            return [x for x, _ in items], any(c for _, c in items)
        if k == "new":
            objs[a["id"]] = getattr(plugin, a["cls"])()
            return objs[a["id"]], False
        if k == "obj":
            return objs[a["id"]], True
        ref = a["ref"]
        if ref.startswith("lit:"):
            if ref not in lits:
                lits[ref] = torch.from_numpy(arrays["lit_" + ref[4:]])
            t = lits[ref]
            chained = False
        else:
            t, chained = handles[ref], True
        assert list(t.shape) == a["shape"] and str(t.dtype).replace("torch.", "") == a["dtype"], (ref, t.shape, t.dtype, a)
        return (t.cpu() if a["device"] == "cpu" else t.to(device)), chained

    def ret(r, got, call, idx, chained):
        k = r["k"]
        if k == "val":
            assert got == r["v"], (call["fn"], got, r["v"])
        elif k == "list":
            assert isinstance(got, (list, tuple)) and len(got) == len(r["items"]), (call["fn"], "list result")
            for x, g in zip(r["items"], got):
                ret(x, g, call, idx, chained)
        elif k == "obj":
            assert type(got).__name__ == r["cls"], (call["fn"], type(got).__name__, r["cls"])
            objs[r["id"]] = got
        else:
            assert isinstance(got, torch.Tensor), (call["fn"], idx, type(got))
            assert list(got.shape) == r["shape"], (call["fn"], idx, tuple(got.shape), r["shape"])
            assert str(got.dtype).replace("torch.", "") == r["dtype"], (call["fn"], idx, got.dtype, r["dtype"])
            handles[r["ref"]] = got
            if r["check"]:
                on_tensor(call, idx, got.detach().cpu().numpy(), arrays["h_" + r["ref"][2:]], chained)

    for call in doc["calls"]:
        pairs = [arg(a) for a in call["args"]]
        chained = any(c for _, c in pairs)
        out = getattr(plugin, call["fn"])(*[v for v, _ in pairs])
        outs = out if call["tuple"] else (out,)
        assert isinstance(out, tuple) == call["tuple"] and len(outs) == len(call["ret"]), (call["fn"], "return arity")
        for idx, (r, g) in enumerate(zip(call["ret"], outs)):
            ret(r, g, call, idx, chained)
    return doc, handles
