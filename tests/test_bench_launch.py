"""bench.py's multi-rank control flow, end to end on CPU (first-lease insurance: the 8-GPU runs are the driver's and cannot be
debugged).  F3DGS_BENCH_STUB=1 replaces the op by a few torch operations and RCCL by gloo; everything else is the real file: the
re-launch under torch.distributed.run, the refusal rules, the exchange through dp.py, every diagnostic leg, the ONE JSON line
relayed from rank 0.  No number in these lines is a measurement ("data": "stub")."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, timeout=420):
    env = dict(os.environ, F3DGS_BENCH_STUB="1", OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout,
                       env=env, cwd=ROOT)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1, f"expected ONE JSON line from rank 0, got {len(lines)}:\n{p.stdout[-2000:]}"
    return json.loads(lines[0])


def test_two_ranks_relaunch_and_relay_one_line():
    d = _run("--gpus", "2", "--steps", "2", "--warmup", "1")
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["data"] == "stub"
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["unit"] == "Mpix/s" and d["value"] > 0
    assert "dp2" in d["config"]["parallelism"] and "workload" in d["config"]
    b = d["dp_breakdown"]
    for k in ("compute_only_ms", "comm_only_ms", "comm_only_direct_ms", "bytes_per_rank", "comm_algbw_GBps", "train_step_allreduce_full_adam_ms",
              "train_step_sharded_optimizer_ms", "two_views_per_gpu_ms_per_step", "two_views_per_gpu_mpix_s"):
        assert k in b and b[k] is not None, (k, b)
    assert b["bytes_per_rank"] == 1500 * (59 + 8) * 4
    # the labels come from the library's options and the line carries the exact-fp32 figure beside the default
    assert d["options"]["bwd_bf16"] == -1 and "bf16" in d["dtype"] and d["blend_backward_contraction"] == "bf16 two-term" and d["ms_per_step_fp32_exact"] > 0
    assert d["roofline"]["kernel"] in ("render_bwd", "render_fwd") and d["roofline_whole_step"]["views_per_gpu_and_step"] == 1


def test_views_per_iter_line_is_strong_scaling_and_counts_views_per_gpu():
    d = _run("--gpus", "2", "--steps", "2", "--warmup", "1", "--views-per-iter", "4")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["views_breakdown"]["views_per_rank"] == 2
    w = d["roofline_whole_step"]
    # one GPU moves the bytes of the TWO views it renders per step
    assert w["views_per_gpu_and_step"] == 2 and w["algorithmic_bytes"] % 2 == 0
    assert abs(w["frac"] - w["algorithmic_bytes"] / (d["ms_per_step"] * 1e-3) / 1e9 / 8000.0) < 1e-12


def test_comm_only_sweeps_the_collective_settings():
    d = _run("--gpus", "2", "--steps", "2", "--warmup", "1", "--comm-only")
    assert d["n_gpus"] == 2 and d["unit"] == "ms" and d["bytes_per_rank"] == 1500 * (59 + 8) * 4
    assert set(d["by_setting"]) >= {"NCCL_ALGO=Ring", "NCCL_ALGO=Tree", "NCCL_ALGO=Ring,NCCL_PROTO=Simple"}
    assert all(("ms" in v) or ("error" in v) for v in d["by_setting"].values())
    assert d["direct_all_to_all_plus_all_gather"]["ms"] > 0


def test_asking_for_more_gpus_than_ranks_is_refused():
    env = dict(os.environ, F3DGS_BENCH_STUB="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)
