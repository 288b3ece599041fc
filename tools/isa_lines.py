#!/usr/bin/env python3
"""Static instruction attribution: which source lines a kernel's VALU / SALU / memory instructions come from.
    tools/isa_lines.py nvdiffrast_amd/csrc/raster.hip k_raster_gradILb0 [top]
Compiles the file for gfx950 with line tables (no GPU needed), cuts out the first kernel whose mangled name
contains the pattern and prints the source lines with the most instructions.  Static counts: loops and
skipped branches are not weighted -- read it next to the SQ_INSTS_* counters of tools/pmc_step.sh."""
import collections, os, re, subprocess, sys, tempfile

src, pat = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
out = os.path.join(tempfile.gettempdir(), "isa_lines.s")
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fhip-fp32-correctly-rounded-divide-sqrt", "-w",
                "-gline-tables-only", "--cuda-device-only", "-S", src, "-o", out], check=True)
files, cur, inside = {}, None, False
cnt = {k: collections.Counter() for k in ("v", "s", "m")}
for line in open(out):
    m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', line)
    if m:
        files[int(m.group(1))] = m.group(2)
    if not inside:
        if re.match(r'^_Z\w*' + re.escape(pat) + r'\w*:', line):
            inside = True
            print("kernel:", line.strip().rstrip(":"))
        continue
    if "s_endpgm" in line:
        break
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', line)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        continue
    t = line.strip()
    if t.startswith("v_"):
        cnt["v"][cur] += 1
    elif t.startswith("s_"):
        cnt["s"][cur] += 1
    elif t.startswith(("ds_", "global_", "scratch_", "buffer_", "flat_")):
        cnt["m"][cur] += 1
print("total VALU %d  SALU %d  memory %d" % tuple(sum(cnt[k].values()) for k in "vsm"))
text = {}
for key in sorted(set(cnt["v"]) | set(cnt["s"]), key=lambda k: -(cnt["v"][k] + cnt["s"][k]))[:top]:
    if key is None:
        continue
    f, l = key
    name = files.get(f, "?")
    if name not in text:
        for d in (os.path.dirname(src), "."):
            p = os.path.join(d, name)
            if os.path.exists(p):
                text[name] = open(p).read().splitlines()
                break
        else:
            text[name] = []
    code = text[name][l - 1].strip()[:100] if 0 < l <= len(text[name]) else ""
    print("%5d V %5d S %4d M  %s:%d  %s" % (cnt["v"][key], cnt["s"][key], cnt["m"][key], os.path.basename(name), l, code))
