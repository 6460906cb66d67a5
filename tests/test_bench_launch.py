"""`python bench.py --gpus N` from a bare shell must spawn its own N ranks (the reference does the same with
mp.spawn + init_process_group, ACT/main_dist.py:59,79-80).  The GPU step needs an MI355X; `--dry-run` swaps in a
CPU stand-in and gloo so the launcher, the barrier / max-over-ranks timing, the logits all-gather and the
one-JSON-line contract are exercised here."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "3", "--warmup", "1",
                          "--sustained-steps", "4", "--batch", "4", "--frames", "2"] + extra,
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks_from_a_bare_shell():
    r = _run(["--gpus", "2"])
    assert r["n_gpus"] == 2 and r["ranks"] == 2 and r["backend"] == "gloo" and r["dry_run"] is True
    assert r["config"]["global_batch"] == 8 and r["config"]["parallelism"] == "dp2"
    assert len(r["per_rank_clips_per_s"]) == 2 and all(v > 0 for v in r["per_rank_clips_per_s"])
    # value = clips of ALL ranks / max-over-ranks time
    assert abs(r["value"] - 8 * 3 / (r["ms_per_step"] * 3e-3)) / r["value"] < 0.01
    assert r["value"] <= 2 * min(r["per_rank_clips_per_s"]) * 1.01
    assert r["sustained"]["steps"] == 4 and r["sustained"]["value"] > 0


def test_bench_single_rank_dry_run_contract():
    r = _run([])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["ranks"] == 1 and r["steps"] == 3 and r["warmup"] == 1


def test_bench_eight_ranks_dry_run():
    """The launch the driver makes on an 8-GPU node (`--gpus 8`: self-launch, 8 ranks, barrier, max-over-ranks, one JSON line,
    logits of all 8 shards gathered), on CPU ranks over gloo."""
    r = _run(["--gpus", "8"])
    assert r["n_gpus"] == 8 and r["ranks"] == 8 and r["backend"] == "gloo" and r["rccl_ranks"] == 0
    assert r["config"]["global_batch"] == 32 and r["config"]["parallelism"] == "dp8" and r["scaling"] == "weak"
    assert len(r["per_rank_clips_per_s"]) == 8 and all(v > 0 for v in r["per_rank_clips_per_s"])
    assert abs(r["value"] - 32 * 3 / (r["ms_per_step"] * 3e-3)) / r["value"] < 0.01



def test_bench_summary_is_the_tail_of_the_line():
    """The driver keeps only the last few KB of bench.py's stdout: the compact `summary` object must be the LAST key of the line and parse
    back from its last 2 KB on its own -- from a dry run, and from a full line of the previous round (every row filled)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "3", "--warmup", "1", "--sustained-steps", "4",
                          "--batch", "4", "--frames", "2"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]

    def summary_from_tail(text):
        tail = text.rstrip()[-2048:]
        i = tail.rfind('"summary": ')
        assert i >= 0, tail[-400:]
        body = tail[i + len('"summary": '):]
        assert body.endswith("}}")
        return json.loads(body[:-1])       # (the line's own closing brace follows the object)

    s = summary_from_tail(out.stdout)
    assert s["value"] > 0 and s["sustained"] > 0 and "glancer" in s and "split_bf16" in s
    assert list(json.loads(out.stdout.strip().splitlines()[-1]).keys())[-1] == "summary"

    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    full = json.loads(open(os.path.join(ROOT, "profiles", "r5_bench_lines", "bench_s2u.json")).read().strip().splitlines()[-1])
    full.pop("summary", None)
    full["summary"] = bench.bench_summary(full)
    text = json.dumps(full)
    assert len(json.dumps(full["summary"])) <= 1536
    s = summary_from_tail(text)
    assert s["serial_value"] == full["serial_value"]["value"] and s["back_to_back_frac"] == full["roofline"]["back_to_back"]["frac"]
    assert s["split_bf16"]["clips_per_s"] == full["also"]["split_bf16"]["clips_per_s"]
    assert s["glancer"]["ms"] == full["next_rows"]["f2_glancer_mobilenetv2"]["ms"]
    assert s["full_forward"] == full["next_rows"]["full_forward_from_uint8"]["value"]
    assert s["evaluate_loop"] == full["next_rows"]["evaluate_loop"]["value"]
    assert s["sth_shipped"]["full_forward"] == full["also"]["sth_shipped_T8_12_P144"]["full_forward_from_uint8"]["value"]
    assert s["config3"]["clips_per_s"] == full["also"]["config3_T16_P128_act"]["clips_per_s"]


def test_bench_distributed_init_failure_leaves_one_diagnosable_line():
    """A failure inside the distributed-init guard (rendezvous or the first collective -- where an RCCL problem shows up on the first
    unattended 8-GPU run) must produce ONE JSON line on stdout with `error`, `rccl_ranks: 0`, the exception text and the HSA_* / NCCL_* /
    rendezvous environment, and a non-zero exit -- not eight tracebacks."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--gpus", "8", "--steps", "3", "--warmup", "1",
                          "--batch", "4", "--frames", "2", "--inject-init-failure"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode != 0
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 8 and r["rccl_ranks"] == 0 and r["value"] is None and "error" in r and "injected failure" in r["exception"]
    assert r["env"]["WORLD_SIZE"] == "8" and r["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "MASTER_ADDR" in r["env"]
    assert 1 <= out.stderr.count("bench.py rank") <= 7       # the other ranks report on stderr (those the launcher had not torn down yet)


def _bare(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "4", "--frames", "2"] + extra,
                          capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)


def test_bench_more_gpus_asked_than_visible_leaves_one_json_line():
    """`bench.py --gpus 2` WITHOUT --dry-run on a box that exposes fewer than 2 devices (this container has none; a driver box with a
    narrower HIP_VISIBLE_DEVICES is the likeliest first multi-GPU failure): exactly one parseable line in the init-failure form from the
    parent -- no torchrun error report, no per-rank traceback -- and exit code 3."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this box has >= 2 GPUs: the launch would be real")
    out = _bare(["--gpus", "2"])
    assert out.returncode == 3, (out.returncode, out.stderr[-2000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["value"] is None and r["rccl_ranks"] == 0 and "error" in r
    assert "--gpus 2" in r["exception"] and "HIP_VISIBLE_DEVICES" in r["exception"]
    assert "visible_gpus" in r and r["visible_gpus"] < 2 and "HIP_VISIBLE_DEVICES" in r["env"] and "ROCR_VISIBLE_DEVICES" in r["env"]
    assert "Traceback" not in out.stderr


def test_bench_rank_without_its_device_reports_through_the_guard():
    """The same condition met INSIDE a rank (launched by the driver's own torchrun, so the parent check never ran): rank 1 of 2 asks for
    HIP device 1 on a box without it -> the init-failure line (rank 0 on stdout, others on stderr), exit code 3, no bare traceback."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this box has >= 2 GPUs")
    out = _bare(["--gpus", "2"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert out.returncode == 3, (out.returncode, out.stderr[-2000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["value"] is None and r["rccl_ranks"] == 0 and "wants HIP device 1" in r["exception"]
    assert "Traceback" not in out.stderr
