"""Development tool: run a sample with every uninitialised allocation of the host glue (`torch.empty`, `empty_like` in
nvdiffrast_amd/torch/_plugin.py) POISONED (NaN for floats, 0x7f7f7f7f for ints, 0xff for the byte scratch), to find
kernels that read memory nobody wrote or outputs that are not fully written.  Usage: tools/poison_check.py [sample]"""
import sys, os, types, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "samples"))
import torch
from nvdiffrast_amd.torch import _plugin


class Poisoned:
    def __getattr__(self, name):
        return getattr(torch, name)

    @staticmethod
    def _poison(t):
        if t.is_floating_point():
            t.fill_(float("nan"))
        elif t.dtype == torch.uint8:
            t.fill_(255)
        else:
            t.fill_(0x7f7f7f7f)
        return t

    def empty(self, *a, **k):
        return self._poison(torch.empty(*a, **k))

    def empty_like(self, *a, **k):
        return self._poison(torch.empty_like(*a, **k))


_plugin.torch = Poisoned()
which = sys.argv[1] if len(sys.argv) > 1 else "cube"
if which == "cube":
    import fit_cube_synth as f
    for i in range(2):
        r = f.fit(iters=300, res=32, batch=8, seed=2)
        print(json.dumps({k: r[k] for k in ("pos_err_after", "col_err_after", "loss_first", "loss_last")}))
