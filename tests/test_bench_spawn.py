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


def test_two_ranks_report_the_job_without_the_image_gather_too():
    j = _run("--gpus", "2", "--batch", "4")
    c = j["collective"]
    img = 4 * 32 * 32 * 4 * 4                                         # items x res^2 x A x f32 of one rank (dry runs render 32x32)
    assert c["image_bytes_sent_per_rank_per_step"] == img and c["xgmi_links_per_gpu_used"] == 1
    assert c["value_without_image_gather"] > 0 and c["ms_per_step_without_image_gather"] > 0


def test_collectives_can_be_forced_on_a_single_rank():
    """`--force-collectives`: the N > 1 code path with a process group of one (what tests/test_gpu_bench.py runs over RCCL)."""
    j = _run("--force-collectives", "--batch", "4", "--chunks", "2")
    assert j["n_gpus"] == 1 and j["rccl_ranks"] == 1 and j["backend"] == "gloo"
    assert j["config"]["gather_images"] is True and j["config"]["chunks"] == 2 and j["collective"] is not None


def test_default_chunking_keeps_chunks_large():
    """Chunks below ~64 items leave the GPU waiting for the host (measured on the MI355X): the default never makes them."""
    assert _run("--gpus", "2", "--batch", "4")["config"]["chunks"] == 1
    assert _run("--gpus", "2", "--workload", "c4")["config"]["chunks"] == 2          # 128 items per rank
