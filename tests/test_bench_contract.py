"""bench.py's output contract (one JSON line on stdout with the driver's keys plus `roofline` and `cpu_baseline`),
checked by running it for two short steps on the GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_emits_one_json_line_with_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-clips", "2",
                          "--cpu-threads", "8", "--batch", "8", "--profile-steps", "1"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["unit"] == "clips/s" and r["n_gpus"] == 1 and r["steps"] == 2 and r["warmup"] == 1
    assert r["higher_is_better"] is True and r["scaling"] == "weak" and r["vs_baseline"] is None
    assert r["dtype"] == "f32" and r["data"] == "synthetic" and "workload" in r["config"] and "model" not in r["config"]
    assert r["value"] > 0 and abs(r["value"] - 8 * 2 / (r["ms_per_step"] * 2e-3)) / r["value"] < 0.01
    rf = r["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = r["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["value"] > 0
    # round 3: everything the DESIGN claims is in the line the driver records
    assert rf["algorithmic_bytes_per_launch"] > 0 and r["serial_value"]["value"] > 0 and r["serial_value"]["streams"] == 1
    also = r["also"]
    for k in ("config2_T8_P96_act", "config3_T16_P128_act", "config4_T8_P128_sth_tsm"):
        assert also[k]["clips_per_s"] > 0 and also[k]["steps"] >= 30, (k, also[k])
    assert also["config2_T8_P96_act"]["T"] == 8 and also["config3_T16_P128_act"]["P"] == 128
    lat = also["latency_small_batch"]
    assert lat["B2_T8_P96"]["graph_bit_identical_to_eager"] is True and lat["B1_T8_P96"]["eager_ms"] > 0
    c5 = also["config5_T16_P144_efficientnet_b3"]
    assert c5["f16_storage"]["clips_per_s"] > 0 and c5["f32_storage"]["local_cnn"]["bound"] == "hbm"
    assert c5["f16_storage"]["local_cnn"]["frac"] == pytest.approx(c5["f16_storage"]["local_cnn"]["achieved"] / 8000.0, abs=1e-3)
    assert also["validate_sth_loop_T8_P128"]["with_baseline_branch"]["value"] > 0
    gr = r["gather_resize"]
    assert gr["S128_to_P96"]["achieved"] > 0 and gr["mixed_S96_128_160_192_to_P96"]["unit"] == "GB/s"
    nr = r["next_rows"]
    assert nr["evaluate_loop"]["value"] > 0 and nr["f1_ingest_u8"]["frac"] > 0 and nr["full_forward_from_uint8"]["value"] > 0
    assert cb["all_cores"] is None or "error" in cb["all_cores"] or cb["all_cores"]["clips_per_s"] > 0


@pytest.mark.gpu
def test_bench_two_ranks_sharing_one_gpu():
    """`python bench.py --gpus 2 --share-gpu` from a bare shell: the launcher spawns two ranks, both run the REAL hot path on
    GPU 0 (their persistent GRU scans and trunks co-run on one device) and all-gather their logits (over gloo: RCCL refuses
    two ranks on one device).  Proves the N > 1 code path on hardware; the throughput is not a scaling figure."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "6", "--warmup", "2",
                          "--batch", "16", "--skip-extras", "--sustained-steps", "0", "--profile-steps", "1"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["ranks"] == 2 and r["config"]["global_batch"] == 32
    assert len(r["per_rank_clips_per_s"]) == 2 and min(r["per_rank_clips_per_s"]) > 0
    assert r["value"] > 0 and "roofline" in r


def test_workload_byte_and_flop_figures():
    """The per-unit figures bench.py prices rooflines with (adafocus_amd/workload.py): known values and the orderings that
    must hold between the denominators reported side by side."""
    from adafocus_amd import workload as w
    assert w.crop_bytes_per_patch(96) == 221184 and w.crop_bytes_per_patch(128) == 393216          # 2 C P^2 4
    assert abs(w.hot_path_flops_per_clip(16, 96) / 1e9 - 24.46) < 0.05 and abs(w.hot_path_flops_per_clip(8, 96) / 1e9 - 12.23) < 0.05
    assert abs(w.mobilenetv2_macs_per_frame(224) / 1e9 - 0.2995) < 0.002
    # glancer: block-level floor <= the plan that runs (strips: round 6) <= the wave-private plan <= the three-launch plan
    block = w.mobilenetv2_block_bytes_per_frame(224)
    strips = w.mobilenetv2_bytes_per_frame(224, strips=True)
    tiles = w.mobilenetv2_bytes_per_frame(224, strips=False)
    unfused = w.mobilenetv2_bytes_per_frame(224, fused=False)
    assert block < strips < tiles < unfused and abs(block / 1e6 - 9.51) < 0.01 and abs(tiles / 1e6 - 22.18) < 0.01 and abs(strips / 1e6 - 13.75) < 0.01
    # frames whose maps are not multiples of 14 take no strip kernel: same bytes either way
    assert w.mobilenetv2_bytes_per_frame(200, strips=True) == w.mobilenetv2_bytes_per_frame(200, strips=False)
    # config 5: the structural floor of a squeeze-and-excite network sits between the block-level bytes and the launch plan's
    blk = w.effnet_block_bytes_per_frame("efficientnet-b3", 144, 2)
    struct = w.effnet_structural_bytes_per_frame("efficientnet-b3", 144, 2)
    plan = w.effnet_bytes_per_frame("efficientnet-b3", 144, 2)
    assert blk < struct < plan and abs(blk / 1e6 - 4.06) < 0.01 and abs(struct / 1e6 - 8.87) < 0.01
