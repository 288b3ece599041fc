#!/usr/bin/env python3
"""profiles/traffic.json from two tools/profile_gpu.sh summaries (headline workload, config 3): PMC bytes per launch of the
kernels, under the names bench.py's per-kernel timing uses.   usage: make_traffic.py <tag> <ch_summary.json> <c3_summary.json>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# kernel name prefix (tools/summarize_profile.py) -> profile name (csrc ProfileScope); first match wins
NAMES = [("k_setup", "raster_setup"), ("k_order", "raster_order"), ("k_flag_order", "raster_flag_order"),
         ("k_fine<false, false, false, false, false, false>", "raster_fine"),
         ("k_interp_raster_grad<", None),        # _da by the instantiation's last argument, below
         ("k_interp_fwd_cols<", "interp_fwd"), ("k_interp_fwd<", None), ("k_interp_grad<", "interp_grad_da"), ("k_raster_grad<true>", "raster_grad_db"), ("k_raster_grad<false>", "raster_grad"),
         ("k_tex_fwd<", "tex_fwd"), ("k_tex_grad_light_w<", "tex_grad_light"), ("k_tex_grad_light<", "tex_grad_light"), ("k_tex_grad_fold<", "tex_grad_fold"), ("k_tex_grad_lean<", "tex_grad"), ("k_tex_grad<", "tex_grad"),
         ("k_mip_grad", "tex_mip_grad"), ("k_aa_discontinuity", "aa_discontinuity"), ("k_aa_analysis", "aa_analysis"), ("k_aa_grad", "aa_grad")]


def section(path):
    out = {}
    for k, r in json.load(open(path)).items():
        if "hbm_bytes" not in r:
            continue
        for pre, name in NAMES:
            if k.startswith(pre):
                if pre == "k_interp_raster_grad<":
                    name = "interp_raster_grad_da" if k.rstrip(">").endswith("true") and k.count(",") == 2 else "interp_raster_grad"
                if pre == "k_interp_fwd<":
                    name = "interp_fwd_da" if k.rstrip(">").endswith("true") else "interp_fwd"
                out[name] = max(out.get(name, 0), int(r["hbm_bytes"]))
                break
    return out


def main():
    tag, ch, c3 = sys.argv[1:4]
    doc = {"_source": "%s: profiles/%s and %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `bench.py --no-extra-configs` and "
                      "`bench.py --workload c3`; read bytes = 2 x FETCH_SIZE KiB per MI355X_MICROARCH.md, median launch)"
                      % (tag, os.path.basename(ch), os.path.basename(c3)),
           "ch": section(ch), "c3": section(c3)}
    json.dump(doc, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
