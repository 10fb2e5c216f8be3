"""bench.py's output contract, on the GPU: ONE JSON line with the driver's keys, the roofline object of the SMVP kernel, the CPU baseline
timed in the same run, and this round's additions (value_cold_protocol, pre_timed_msms, roofline.kernel_ms_rocprof).  A small size keeps it short;
the numbers themselves are not judged here."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_keys(built):
    d = _bench(["--steps", "6", "--warmup", "2", "--logn", "14", "--cpu-sample-logn", "12"], BENCH_STEADY_MSMS="8", BENCH_CONFIGS="c1:10,c3:12,c4:13,c5:6x11")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline", "value_steady_state", "pre_timed_msms", "protocol", "configs"):
        assert key in d, key
    # `value` is the literal protocol: exactly the W warm-up steps in front of its timed region
    assert d["untimed_steps_before_timed_region"] == 2 and d["pre_timed_msms"] == 2 and d["value_steady_state"] > 0
    assert d["steady_state"]["untimed_steps_of_this_workload_directly_before"] == 8 + 2
    assert d["unit"] == "MSM/s" and d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and rf["kernel"] == "k_smvp_chunks"
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0 < rf["frac"] < 1 and "traffic" in rf and "kernel_ms_rocprof" in rf
    # grouped small MSMs run 14-bit windows: the algorithmic bytes follow the engine's window size, not the 16-bit constants
    assert d["config"]["window_bits"] == 14 and d["config"]["msms_per_launch"] > 1
    wl = d["fixed_base_tables_wide"]  # the opt-in mode's line carries its own SMVP roofline (its stage times explain a difference to the headline)
    assert wl["same_result_as_headline_mode"] is True and wl["value"] > 0 and 0 < wl["roofline"]["frac"] < 1 and wl["roofline"]["kernel_ms"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["unit"] == "MSM/s" and cb["value"] > 0 and "sample" in cb
    # BASELINE.json's other configs on the same line, each verified in the run (round 5)
    cf = d["configs"]
    assert set(cf) == {"c1", "c3", "c4", "c5"}
    assert cf["c1"]["value"] > 0 and cf["c1"]["verified_bit_exact_vs_cpu"] is True and cf["c1"]["cpu_path"]["cores"] == 1 and 0 < cf["c1"]["roofline"]["frac"] < 1
    for mode in ("plain", "tables_wide"):
        assert cf["c3"][mode]["all_8_shares_combined_equal_whole_msm"] is True and cf["c3"][mode]["ms_per_msm_one_rank_share"] > 0, mode
        assert 0 < cf["c3"][mode]["roofline"]["frac"] < 1
    assert cf["c3"]["whole_msm_verified_bit_exact_vs_cpu"] is True
    assert cf["c4"]["timed_msms"] >= 5 and cf["c4"]["whole_equals_8_window_range_shares_combined"] is True and 0 < cf["c4"]["roofline"]["frac"] < 1
    assert any(k.startswith("slice_") and v is True for k, v in cf["c4"].items())
    for mode in ("endomorphism", "tables_wide"):
        assert cf["c5"][mode]["value"] > 0 and 0 < cf["c5"][mode]["roofline"]["frac"] < 1, mode
    assert cf["c5"]["both_modes_same_results"] is True and cf["c5"]["all_results_differ"] is True and cf["c5"]["vector_5_verified_bit_exact_vs_cpu"] is True


def test_emulated_share_line_and_native_multi_gpu_mode(built):
    d = _bench(["--steps", "8", "--warmup", "2", "--logn", "14", "--no-cpu-baseline"], BENCH_EMULATE_WORLD="8", BENCH_STEADY_MSMS="8")
    assert d["emulated_world"] == 8 and d["config"]["msms_per_launch"] == 8 and d["config"]["windows_per_gpu"] == 2 and "cpu_baseline" not in d
    n = _bench(["--gpus", "2", "--steps", "6", "--warmup", "2", "--logn", "14"], BENCH_MGPU_NATIVE="1", BENCH_MGPU_IDS="0,0", BENCH_STEADY_MSMS="4")
    assert n["native_mgpu"] is True and n["n_gpus"] == 2 and n["sharded_result_equals_single_gpu"] is True


def test_multi_rank_line_carries_the_native_c_abi_figure(built):
    """`bench.py --gpus N` (N > 1): rank 0's ONE line has the torch.distributed figure (`value`) and, from one fresh child process, the same
    workload through the in-process multi-GPU C ABI (`value_native_mgpu`, `native_rccl_ranks`).  Rehearsed on the one GPU of this box: two
    ranks over gloo on GPU 0, the child with two contexts on GPU 0 (pinned-buffer gather, hence 0 RCCL ranks); every MSM is combined once
    across the ranks and each rank checks its own."""
    d = _bench(["--gpus", "2", "--steps", "6", "--warmup", "2", "--logn", "14"], BENCH_ALL_ON_GPU0="1", BENCH_DIST_BACKEND="gloo", BENCH_STEADY_MSMS="4")
    assert d["n_gpus"] == 2 and d["dist_ranks"] == 2 and d["value"] > 0 and d["sharded_result_equals_single_gpu"] is True
    assert d["value_native_mgpu"] > 0 and abs(d["value_native_mgpu"] - 1e3 / d["ms_per_step_native_mgpu"]) < 1e-6 * d["value_native_mgpu"]
    assert d["native_rccl_ranks"] == 0 and d["native_mgpu"]["result_equals_single_gpu"] is True
    assert d["native_mgpu"]["config"]["device_ids"] == [0, 0]
    # a failing child: null fields and its error, the torch figure unaffected
    e = _bench(["--gpus", "2", "--steps", "6", "--warmup", "2", "--logn", "14"], BENCH_ALL_ON_GPU0="1", BENCH_DIST_BACKEND="gloo", BENCH_STEADY_MSMS="4",
               BENCH_NATIVE_TIMEOUT_S="0.01")
    assert e["value"] > 0 and e["value_native_mgpu"] is None and e["native_rccl_ranks"] is None and "error" in e["native_mgpu"]
