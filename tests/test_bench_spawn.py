"""`bench.py --gpus N` must really run N ranks (the flag used to be dead code).  Dry run on CPU: gloo
instead of RCCL, a stand-in instead of the kernels, everything else -- self-spawn under torch.distributed.run,
item sharding, chunked image all-gather inside the step, shared-gradient all-reduce, barrier-fenced timing, max over
ranks, one JSON line from rank 0 -- is the code the GPU run executes."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run-cpu", "--steps", "2", "--warmup", "1", *flags],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # rank 0 only
    assert r.stdout.strip() == lines[0], r.stdout          # and nothing else on stdout (everything else goes to stderr)
    return json.loads(lines[0])


def test_single_rank_line():
    j = _run()
    assert j["n_gpus"] == 1 and j["rccl_ranks"] == 1 and j["dry_run"] is True and j["scaling"] == "weak"
    assert j["config"]["gather_images"] is False and j["config"]["chunks"] == 1


@pytest.mark.parametrize("workload,scaling,total", [("ch", "weak", 8), ("c4", "strong", 256)])
def test_two_ranks_are_spawned_and_take_part(workload, scaling, total):
    flags = ["--gpus", "2", "--chunks", "4", "--workload", workload] + (["--batch", "4"] if workload == "ch" else [])
    j = _run(*flags)
    assert j["n_gpus"] == 2 and j["rccl_ranks"] == 2 and j["backend"] == "gloo"
    assert j["scaling"] == scaling and j["config"]["total_items"] == total
    assert j["config"]["batch_per_gpu"] == total // 2
    assert j["config"]["gather_images"] is True and j["config"]["chunks"] == 4
    assert j["gathered_rows_chunk0"] == 2 * (total // 2 // 4)          # every rank's chunk arrived
    assert j["value"] > 0 and j["ms_per_step"] > 0


def test_two_ranks_default_is_the_packed_pipelined_gather_with_the_other_figures_beside_it():
    """The default N > 1 line: the output images all-gathered EVERY step as rgb8 (three channels, one byte each), collected one
    step later; next to it the same job without the gather and with the f32 gather inside the step (rounds 2-5's default)."""
    j = _run("--gpus", "2", "--batch", "4")
    c = j["collective"]
    assert j["config"]["gather_format"] == "rgb8" and j["config"]["gather_pipelined"] is True and j["config"]["chunks"] == 1
    assert c["gather_format"] == "rgb8" and c["gather_pipelined"] is True and c["gather_payload"].startswith("3 of 4 channels, 1 byte")
    assert c["image_bytes_sent_per_rank_per_step"] == 4 * 32 * 32 * 3 and c["xgmi_links_per_gpu_used"] == 1      # items x res^2 x 3 bytes (dry runs render 32x32)
    assert j["gathered_rows_total"] == 8 and j["gathered_payload"] == ["uint8", 3]
    assert c["value_without_image_gather"] > 0 and c["ms_per_step_without_image_gather"] > 0
    assert c["value_with_f32_gather_in_step"] > 0 and c["ms_per_step_with_f32_gather_in_step"] > 0
    ps = c["predicted_scaling"]["pipelined_per_step_gather"]
    assert set(ps) == {"f32", "f16", "rgba8", "rgb8"} and all(ps[f]["8"] > 0 for f in ps)       # (tiny dry-run items: all formats ~ 8 x; the real arithmetic is asserted below)


@pytest.mark.parametrize("fmt,dtype,ch", [("f32", "float32", 4), ("f16", "float16", 4), ("rgba8", "uint8", 4), ("rgb8", "uint8", 3)])
@pytest.mark.parametrize("pipelined", [True, False])
def test_gather_formats_and_pipelining_at_world_2(fmt, dtype, ch, pipelined):
    if (fmt, pipelined) not in (("f32", True), ("f16", False), ("rgba8", True), ("rgb8", False)):
        pytest.skip("four cells of the matrix run here (the default line covers rgb8 pipelined); tests/test_parallel_gloo.py runs all of it on the API")
    j = _run("--gpus", "2", "--batch", "4", "--gather-format", fmt, *([] if pipelined else ["--no-gather-pipeline"]))
    assert j["config"]["gather_format"] == fmt and j["config"]["gather_pipelined"] is pipelined
    assert j["collective"]["image_bytes_sent_per_rank_per_step"] == 4 * 32 * 32 * ch * {"float32": 4, "float16": 2, "uint8": 1}[dtype]
    assert j["gathered_rows_total"] == 8
    if pipelined:
        assert j["gathered_payload"] == [dtype, ch]


def test_collectives_can_be_forced_on_a_single_rank():
    """`--force-collectives`: the N > 1 code path with a process group of one (what tests/test_gpu_bench.py runs over RCCL)."""
    j = _run("--force-collectives", "--batch", "4", "--chunks", "2")
    assert j["n_gpus"] == 1 and j["rccl_ranks"] == 1 and j["backend"] == "gloo"
    assert j["config"]["gather_images"] is True and j["config"]["chunks"] == 2 and j["collective"] is not None


def test_chunk_policy_follows_the_link_arithmetic():
    """the chunk count comes from the bytes one link carries against the rank's compute and the host's
    launch cost, not from items // 64 (which gave C4 at N = 8 one chunk: nothing overlapped the forward)."""
    sys.path.insert(0, ROOT)
    import bench
    item = 512 * 512 * 4 * 4
    # C4 at N = 8: 32 items per GPU, 0.26 ms of kernels against 0.88 ms of gather: start the gather early (3 chunks with the
    # compiled host layer's 0.07 ms per chunk; 2 with the Python layer's 0.20 of rounds 2-5), but never so many chunks that the
    # host's launch work exceeds what the earlier start gains
    c, plan = bench.plan_chunks(32, item, 32 * 0.008, 7)
    assert c == 3 and abs(plan["gather_floor_ms"] - 0.877) < 0.01
    assert bench.plan_chunks(64, item, 64 * 0.008, 7)[0] == 6                 # weak-scaled headline: gather-bound as well
    assert bench.plan_chunks(64, item, 2.0, 7)[0] >= 3                        # kernels about as long as the gather: overlap pays, more chunks
    assert bench.plan_chunks(64, item, 64 * 0.2, 7)[0] == 1                   # kernels far longer than the gather: it hides behind the backward pass
    assert bench.plan_chunks(4, 32 * 32 * 16, 0.001, 1)[0] == 1               # tiny dry-run items: the host bounds everything
    assert bench.plan_chunks(64, item, 0.5, 0) == (1, None)                   # nobody to send to
    ps = bench.predicted_scaling(0.008, "strong", None, 256, item)
    assert ps["with_image_gather"]["8"] < 3.0 < 6.0 < ps["without_image_gather"]["8"] <= 8.0
    pw = bench.predicted_scaling(0.008, "weak", 64, None, item)
    assert pw["with_image_gather"]["8"] < 3.0 and pw["without_image_gather"]["8"] > 7.0


def test_default_chunking_in_a_spawned_job():
    assert _run("--gpus", "2", "--batch", "4")["config"]["chunks"] == 1               # dry-run items are tiny: one chunk
    j = _run("--gpus", "2", "--workload", "c4")
    assert j["collective"]["chunk_plan"]["predicted_step_ms_by_chunks"] and j["collective"]["predicted_scaling"]["with_image_gather"]["8"] > 0


def test_eight_ranks_c4():
    """BASELINE configs[3] at its own world size (gloo stand-in): 256 items -> 32 per rank, every rank's chunks gathered,
    all eight ranks inside the timed region, timing windows reported."""
    j = _run("--gpus", "8", "--workload", "c4", "--windows", "2")
    assert j["n_gpus"] == 8 and j["rccl_ranks"] == 8 and j["scaling"] == "strong"
    assert j["config"]["total_items"] == 256 and j["config"]["batch_per_gpu"] == 32 and j["config"]["gather_images"] is True
    assert j["config"]["gather_format"] == "rgb8" and j["config"]["gather_pipelined"] is True
    assert j["gathered_rows_total"] == 256                                            # 8 ranks x 32 items arrived on rank 0
    assert j["timing"]["windows"] == 2 and j["ms_per_step_min"] <= j["ms_per_step"] <= j["ms_per_step_max"]
    assert j["collective"]["xgmi_links_per_gpu_used"] == 7


def test_gather_every_kth_step_at_world_8():
    """the one schedule with an image gather that the link arithmetic lets scale: the complete batch of
    images every k-th step only, its all-gather running behind the following steps' kernels (double-buffered receive)."""
    j = _run("--gpus", "8", "--workload", "c4", "--windows", "2", "--gather-every", "4", "--steps", "5")
    assert j["n_gpus"] == 8 and j["rccl_ranks"] == 8 and j["config"]["gather_every"] == 4 and j["config"]["gather_images"] is True
    assert j["gathered_rows_total"] == 256                                   # the gathered batch is complete when it is there
    assert set(j["collective"]["predicted_scaling"]) >= {"with_image_gather", "without_image_gather", "gather_every_4", "gather_every_8"}
    # (the dry run's items are 32x32 stand-ins; the arithmetic with this tree's measured 6.6 us per item at 512^2:)
    sys.path.insert(0, ROOT)
    import bench
    pw = bench.predicted_scaling(0.0066, "weak", 64, None, 512 * 512 * 4 * 4)
    assert pw["gather_every_4"]["8"] >= 6.0 and pw["gather_every_8"]["8"] >= 7.0 and pw["with_image_gather"]["8"] < 3.0
    # ... and with round 6's 5.0 us per item: the per-step gather that the links can carry is the 8-bit RGB image, one step deep
    pw = bench.predicted_scaling(0.00503, "weak", 64, None, 512 * 512 * 4 * 4)["pipelined_per_step_gather"]
    assert pw["rgb8"]["8"] >= 6.0 and 5.0 <= pw["rgba8"]["8"] < 6.0 and pw["f16"]["8"] < 3.0 and pw["f32"]["8"] < 1.5
