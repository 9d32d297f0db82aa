"""`python bench.py --gpus N` starts its own N ranks (the driver's N = 1 command form with another N): the launcher, the
rendezvous on 127.0.0.1 and the rank count the process group really saw — without a GPU (control flow only, gloo) and, on the GPU box, the
whole bench line of two ranks sharing the one device (FA_BENCH_BACKEND=gloo: a rehearsal of every N > 1 leg incl. the sharded-VBx all-gather)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bare_command_with_gpus_2_starts_two_ranks():
    r, line = _run(["--gpus", "2", "--launch-check"], timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert line == {"launch_check": True, "n_gpus": 2, "world_size_env": 2, "backend": "gloo"}


def test_single_rank_needs_no_launcher():
    r, line = _run(["--launch-check"], timeout=120)
    assert r.returncode == 0 and line["n_gpus"] == 1


def test_rank_count_that_differs_from_gpus_is_refused():
    r, line = _run(["--gpus", "2", "--launch-check"], {"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"}, timeout=120)
    assert r.returncode == 2 and line is None and "WORLD_SIZE 3" in r.stderr


@pytest.mark.gpu
def test_two_ranks_on_the_one_gpu_print_a_whole_line():
    """Two ranks, gloo, both on cuda:0, a 1 h recording each: n_gpus is what the process group saw, every N > 1 leg ran."""
    r, line = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--hours", "1", "--mel-steps", "5", "--mel-warmup", "2", "--clock-warm-s", "0",
                    "--chunks", "64", "--ctc-matrices", "64"], {"FA_BENCH_BACKEND": "gloo"}, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 2
    assert line["e2e_8h"]["all_ranks_ok"] is True
    assert line["vbx_sharded"]["all_ranks_same_elbos"] is True and line["vbx_sharded"]["elbos_equal_single_device"] is True
    assert line["config"]["vbx_sharded_all_ranks_same_elbos"] is True
    assert line["ctc"]["ids_exact"] is True and line["ctc"]["matrices_per_rank"] == 32
    assert abs(line["value"] - 2 * 1.0 * 2 / (line["ms_per_step"] * 2 / 1e3)) < 1e-6 * line["value"]


@pytest.mark.gpu
def test_nccl_with_more_ranks_than_gpus_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer GPUs than ranks")
    r, line = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], timeout=300)
    assert r.returncode != 0 and line is None and "needs 2 visible GPUs" in r.stderr
