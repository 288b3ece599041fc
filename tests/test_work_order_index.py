"""The index arithmetic of ordered launches (csrc/nvdr_device.hpp ordered_list_index / tile_flags_ordered_grid), on the host:
for every number of bins and of covered bins, the launch that tile_flags_ordered_grid sizes visits every entry of the work
order exactly once, each XCD its covered share before its empty share, and the shares differ by at most one eighth's rounding."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

SRC = r'''
#include "nvdr_device.hpp"
#include <cstdio>
#include <vector>
using namespace nvdr;
int main()
{
    long long checked = 0;
    for (int nBins = 1; nBins <= 700; nBins += (nBins < 260 ? 1 : 37))
        for (int nCov = 0; nCov <= nBins; nCov++) {
            TileFlags t{}; t.nBins = nBins;
            const long long grid = tile_flags_ordered_grid(t, 1);            // one workgroup per bin
            if (grid % 8) { printf("grid not a multiple of 8\n"); return 1; }
            std::vector<int> seen(nBins, 0);
            int share[8][2] = {};
            for (int xcd = 0; xcd < 8; xcd++) {
                bool inEmpty = false, done = false;
                for (int slot = 0; slot < grid / 8; slot++) {
                    const int idx = ordered_list_index(nBins, nCov, xcd, slot);
                    if (idx < 0) { done = true; continue; }
                    if (done) { printf("hole in the share of xcd %d (nBins %d nCov %d)\n", xcd, nBins, nCov); return 1; }
                    if (idx >= nBins) { printf("index out of range\n"); return 1; }
                    const bool empty = idx >= nCov;
                    if (inEmpty && !empty) { printf("covered after empty\n"); return 1; }
                    inEmpty = empty;
                    seen[idx]++; share[xcd][empty]++;
                }
            }
            for (int b = 0; b < nBins; b++) if (seen[b] != 1) { printf("entry %d visited %d times (nBins %d nCov %d)\n", b, seen[b], nBins, nCov); return 1; }
            // the paired form (k_interp_fwd_cols): every entry is handled exactly once -- by its own workgroup, or, an empty bin, by
            // the covered bin's workgroup that takes it on (whose own workgroup then leaves); partners are always empty bins
            std::vector<int> handled(nBins, 0);
            for (int xcd = 0; xcd < 8; xcd++)
                for (int slot = 0; slot < grid / 8; slot++) {
                    int own, partner; bool skip;
                    ordered_list_pair(nBins, nCov, xcd, slot, own, partner, skip);
                    if (own != ordered_list_index(nBins, nCov, xcd, slot)) { printf("pair: own differs from the plain index\n"); return 1; }
                    if (own >= 0 && !skip) handled[own]++;
                    if (skip && (own < nCov || partner >= 0)) { printf("pair: a covered bin is skipped\n"); return 1; }
                    if (partner >= 0) { if (partner < nCov || partner >= nBins || own < 0 || own >= nCov) { printf("pair: bad partner\n"); return 1; } handled[partner]++; }
                }
            for (int b = 0; b < nBins; b++) if (handled[b] != 1) { printf("pair: entry %d handled %d times (nBins %d nCov %d)\n", b, handled[b], nBins, nCov); return 1; }
            for (int e = 0; e < 2; e++) {
                int lo = 1 << 30, hi = 0, n = e ? nBins - nCov : nCov;
                for (int x = 0; x < 8; x++) { lo = share[x][e] < lo ? share[x][e] : lo; hi = share[x][e] > hi ? share[x][e] : hi; }
                if (hi > (n + 7) / 8) { printf("share too large\n"); return 1; }
            }
            checked++;
        }
    printf("ok %lld\n", checked);
    return 0;
}
'''


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_ordered_launch_visits_every_bin_once(tmp_path):
    src = tmp_path / "order_index.hip"
    src.write_text(SRC)
    exe = tmp_path / "order_index"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "--cuda-host-only", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "nvdiffrast_amd", "csrc"),
                        "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok "), (r.stdout, r.stderr[-2000:])
    assert int(r.stdout.split()[1]) > 30000
