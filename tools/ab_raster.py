#!/usr/bin/env python3
"""Interleaved A/B of several builds of the library on the rasterizer's forward pass (tools/build_ab.sh makes the builds):
every variant is loaded into THIS process and called in turn on the same buffers, so box-to-box and run-to-run drift cancel.
    python tools/ab_raster.py [--rounds R] [--scenes ch,c2,s10k,dense] name[=path] ...     ("main" = the in-tree library)
Prints, per scene and variant, the library's own hipEvent times of raster_fine / raster_setup (median over the rounds) and
whether the outputs equal the first variant's bit for bit."""
import ctypes, json, os, statistics, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nvdiffrast_amd import _capi
from tools.bench_regimes import scene


def load(path):
    lib = ctypes.CDLL(path)
    for name, (res, args) in _capi.SIGNATURES.items():
        fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
    return lib


def prof(lib, cap=64):
    names = (ctypes.c_char_p * cap)(); total = (ctypes.c_double * cap)(); cnt = (ctypes.c_int * cap)()
    n = lib.nvdr_profile_read(names, total, cnt, cap)
    return {names[i].decode(): total[i] / max(cnt[i], 1) for i in range(n)}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 30
    scenes = sys.argv[sys.argv.index("--scenes") + 1].split(",") if "--scenes" in sys.argv else ["ch", "c2", "s10k", "dense"]
    args = [a for a in args if a not in (str(rounds), ",".join(scenes))]
    libs = []
    for a in args or ["main"]:
        name, _, path = a.partition("=")
        path = path or (os.path.join(ROOT, "nvdiffrast_amd", "libnvdr_hip.so") if name == "main" else os.path.join(ROOT, "nvdiffrast_amd", f"libnvdr_hip_{name}.so"))
        libs.append((name, load(path)))
    dev = torch.device("cuda", 0)
    for sc in scenes:
        b, R = scene(sc)
        pos = torch.from_numpy(b["pos"]).to(dev); tri = torch.from_numpy(b["tri"]).to(dev)
        N, V, T = pos.shape[0], pos.shape[1], tri.shape[0]
        out = torch.empty(N, R, R, 4, device=dev); out_db = torch.empty_like(out)
        l0 = libs[0][1]
        flags = torch.empty(int(l0.nvdr_tile_flags_bytes(N, R, R)), dtype=torch.uint8, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        res = {}
        ref = None
        per = {name: {"raster_fine": [], "raster_setup": []} for name, _ in libs}
        scratch = {}
        for name, lib in libs:
            nb = int(lib.nvdr_rasterize_scratch_bytes(N, T, R, R))
            scratch[name] = torch.zeros(nb + 256, dtype=torch.uint8, device=dev)
            lib.nvdr_profile_enable(1)
        for r in range(rounds + 2):
            for name, lib in libs:
                s = scratch[name]
                lib.nvdr_profile_reset()
                rc = lib.nvdr_rasterize_fwd(pos.data_ptr(), tri.data_ptr(), None, 1, N, V, T, T, R, R, None, None,
                                            (s.data_ptr() + 255) // 256 * 256, s.numel() - 256, 1 if r > 0 else 0, -1,
                                            out.data_ptr(), out_db.data_ptr(), flags.data_ptr(), stream)
                assert rc == 0, lib.nvdr_last_error()
                torch.cuda.synchronize()
                if r == 0:
                    o = out.clone()
                    if ref is None: ref = o
                    res[name] = bool(torch.equal(o.view(torch.int32), ref.view(torch.int32)))
                elif r >= 2:
                    p = prof(lib)
                    for k in per[name]: per[name][k].append(p.get(k, float("nan")))
        for name, _ in libs:
            print(json.dumps({"scene": sc, "variant": name, "fine_us": round(statistics.median(per[name]["raster_fine"]) * 1e3, 2),
                              "fine_min_us": round(min(per[name]["raster_fine"]) * 1e3, 2),
                              "setup_us": round(statistics.median(per[name]["raster_setup"]) * 1e3, 2), "same_as_first": res[name]}), flush=True)


if __name__ == "__main__":
    main()
