"""bench.py's N > 1 orchestration on CPU: `python bench.py --gpus 2 --stub-device` from a bare environment must start its
own two ranks (torch.distributed.run as a child process), run the timed loop with its barriers and max-over-ranks, count
the ranks, gather the masks in item order and print ONE line -- everything around the device, which a host stand-in
replaces.  What a driver on a multi-GPU node runs is this code with the stand-in taken out."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, env=env, timeout=timeout)


def test_bare_gpus_2_spawns_its_ranks_and_prints_one_line():
    r = _run(["--gpus", "2", "--stub-device", "--steps", "3", "--warmup", "1", "--repeats", "3"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["repeats"] == 3
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["unit"] == "images/s"
    assert "stub_device" in d                                   # never mistaken for a measurement
    assert d["rccl"]["rccl_ranks"] == 2 and "REHEARSAL" in d["rccl"]["backend"]
    g = d["rccl"]["gather"]
    assert g["masks"] == 2 and g["own_share_intact"] is True and g["bytes"] == 2 * 1024 * 1024
    assert d["value"] > 0 and abs(d["ms_per_step"] * d["steps"] * d["value"] / 1e3 - 2 * d["steps"]) < 1e-6
    # every rank's own rate of the median repeat (a straggler is visible there; `value` uses the slowest rank's clock)
    assert len(d["per_rank_value"]) == 2 and all(v > 0 for v in d["per_rank_value"])
    assert min(d["per_rank_value"]) * 2 <= d["value"] * (1 + 1e-9) <= sum(d["per_rank_value"]) * (1 + 1e-9)


def test_mismatched_world_size_is_refused():
    r = _run(["--gpus", "2", "--stub-device"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "does not match WORLD_SIZE" in (r.stderr + r.stdout)


def test_single_rank_stub_prints_the_contract_fields():
    r = _run(["--stub-device", "--steps", "2", "--warmup", "0", "--repeats", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config"):
        assert key in d
    assert d["n_gpus"] == 1 and d["rccl"]["rccl_ranks"] == 1 and len(d["per_rank_value"]) == 1


def test_gemm_shape_table_matches_the_survey_figures():
    """bench.py's algorithmic FLOPs / bytes per launch (roofline.per_kernel): the linear layers add up to SURVEY.md section
    8(d)'s 695.8 GFLOP per ViT-B image, and proj's arithmetic intensity is the 152 FLOP per byte VERDICT r05 computed --
    under the ridge of 312, while fc2, qkv and fc1 are above it."""
    sys.path.insert(0, str(ROOT))
    import bench
    from dlimgedit_amd.sam_config import get_config
    cfg = get_config("vit_b")
    one = bench.gemm_shapes(cfg, 1.0)
    linear = 12 * sum(one[k]["flops"] for k in ("gemm_proj", "gemm_fc2", "gemm_norm", "gemm_norm_gelu"))
    assert abs(linear / 1e9 - 695.8) < 0.1 and abs(one["gemm_patch"]["flops"] / 1e9 - 4.83) < 0.01
    four = bench.gemm_shapes(cfg, 4.0)
    intensity = {k: v["flops"] / v["bytes"] for k, v in four.items()}
    assert 150 < intensity["gemm_proj"] < 155 and intensity["gemm_patch"] < 312.5
    assert all(intensity[k] > 312.5 for k in ("gemm_fc2", "gemm_norm", "gemm_norm_gelu"))
    assert abs(four["gemm_proj"]["bytes"] / 1e6 - 127.5) < 1.0
