"""bench.py --gpus N: the rank launcher (CPU; no GPU is touched).  BASELINE config 3 / SURVEY 8(e): one process per GPU."""
import os
import subprocess
import sys

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_launcher_command_starts_one_rank_per_gpu():
    b = _bench_module()
    cmd = b.launcher_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], 29511, python="python3")
    assert cmd[:3] == ["python3", "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    k = cmd.index(BENCH)
    assert cmd[k + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]   # the ranks see the same arguments


def test_more_gpus_than_visible_is_an_error_not_a_smaller_run():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this host has the GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=env,
                       timeout=300)
    assert r.returncode != 0
    assert r.stdout.strip() == ""                       # no JSON line: nothing was measured
    assert "--gpus 2" in r.stderr and "visible" in r.stderr


def test_world_size_must_match_gpus():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29512")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=env,
                       timeout=300)
    assert r.returncode == 4
    assert "WORLD_SIZE=2" in r.stderr and r.stdout.strip() == ""


def test_kernel_name_follows_shape_and_mode():
    b = _bench_module()
    assert b.forward_kernel_name(2048, "p1") == "sot_area_full_kernel<256, 8, 1, false, 0>"          # p = 1, one grid: merge-free
    assert b.forward_kernel_name(2048, "p1", same_grid=False) == "sot_forward_full_kernel<256, 8, 1, 1, false, false, 0, false>"
    assert b.forward_kernel_name(1025, "cutoff") == "sot_forward_full_kernel<128, 9, 2, 2, true, true, 1025, false>"
    assert b.forward_kernel_name(1025, "cutoff", backward=True) == "sot_backward_full_kernel<128, 9, 2, 2, true, true, 1025, false, true, 4, false>"
    assert b.forward_kernel_name(2048, "cutoff", backward=True) == "sot_backward_full_kernel<256, 8, 2, 2, true, true, 0, false, true, 1, false>"
    assert b.forward_kernel_name(3000, "p1") == "sot_area_full_kernel<384, 8, 1, false, -1>"            # run-time length on the 3072-point geometry
    assert b.forward_kernel_name(4000, "p1") == "sot_area_full_kernel<512, 8, 1, false, -1>"
    assert b.forward_kernel_name(1200, "cutoff") == "sot_forward_full_kernel<192, 8, 1, 2, true, true, -1, false>"
    assert b.forward_kernel_name(2000, "cutoff", backward=True) == "sot_backward_full_kernel<256, 8, 1, 2, true, true, -1, false, false, 1, false>"
    assert b.forward_kernel_name(1000, "cutoff", backward=True) == "sot_backward_full_kernel<128, 8, 2, 2, true, true, -1, false, true, 4, false>"
    assert b.forward_kernel_name(1025, "cutoff", batch=16384) == "sot_forward_full_kernel<64, 17, 4, 2, true, true, 1025, false>"     # large batches: one wave per row
    assert b.forward_kernel_name(1025, "cutoff", batch=4096) == "sot_forward_full_kernel<128, 9, 2, 2, true, true, 1025, false>"
    assert b.forward_kernel_name(1025, "p1") == "sot_area_full_kernel<64, 17, 4, false, 1025>"
    assert b.forward_kernel_name(129, "p1") == "sot_area_half_kernel<5, 8, false, 129>"
    assert b.forward_kernel_name(257, "cutoff", batch=65536) == "sot_forward_half_kernel<9, 8, 2, true, true, 257>"      # two rows per wave
    assert b.forward_kernel_name(257, "cutoff", batch=16384) == "sot_forward_full_kernel<64, 5, 4, 2, true, true, 257, false>"
    assert b.forward_kernel_name(1000, "p1") == "sot_area_full_kernel<64, 16, 4, false, -1>"
    assert b.forward_kernel_name(1000, "cutoff", batch=16384) == "sot_forward_full_kernel<64, 16, 4, 2, true, true, -1, false>"
    assert "generic" in b.forward_kernel_name(100, "p1") and "generic" in b.forward_kernel_name(9000, "p1")


def test_visible_gpus_reads_the_environment_not_the_runtime(monkeypatch):
    b = _bench_module()
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,3,5")
    assert b.visible_gpus() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert b.visible_gpus() == 0


def test_bench_uses_the_oracle_only_for_the_cpu_baseline():
    src = open(BENCH).read()
    body = src.split("def cpu_baseline", 1)
    assert "from oracle" not in body[0] and "import oracle" not in body[0]
    after = body[1].split("\ndef ", 1)[1]   # everything behind cpu_baseline()
    assert "from oracle" not in after and "import oracle" not in after


def test_global_rows_flag_means_strong_scaling():
    """--global-rows G (BASELINE config 3: 65536): rows per rank = G / N, `scaling` = "strong"; the shards are cuts of ONE global batch."""
    import argparse
    import pytest
    import torch
    b = _bench_module()
    ns = b.resolve_rows(argparse.Namespace(gpus=8, rows=8192, global_rows=65536))
    assert (ns.rows, ns.scaling) == (8192, "strong")
    ns = b.resolve_rows(argparse.Namespace(gpus=2, rows=8192, global_rows=65536))
    assert (ns.rows, ns.scaling) == (32768, "strong")
    assert b.resolve_rows(argparse.Namespace(gpus=4, rows=8192, global_rows=0)).scaling == "weak"
    with pytest.raises(SystemExit):
        b.resolve_rows(argparse.Namespace(gpus=3, rows=8192, global_rows=65536))
    # the global batch is the concatenation of the weak-scaling ranks' blocks (seed 1234 + block), whatever the cut
    from sot_amd.bench_inputs import spectrum_pairs
    x, y = b.global_batch_rows(60, 100, 16, block=64)          # spans blocks 0, 1, 2
    x0, y0 = spectrum_pairs("uniform", 64, 16, 16, 1234)
    x1, y1 = spectrum_pairs("uniform", 64, 16, 16, 1235)
    x2, y2 = spectrum_pairs("uniform", 64, 16, 16, 1236)
    assert torch.equal(x, torch.cat([x0[60:], x1, x2[:32]])) and torch.equal(y, torch.cat([y0[60:], y1, y2[:32]]))


def test_scalar_copies_of_the_nested_figures_reach_the_top_level():
    """Round-5 review: the driver's record of the bench line keeps scalars (top level, `config`, `roofline`, `cpu_baseline`) and drops nested
    objects, so the figures of the pipeline north_star names, of the paper's step at the paper's batch and of the per-row path have scalar
    copies.  flatten_for_scalar_readers adds them from the nested entries (no new measurement) and leaves None for what a run did not measure."""
    b = _bench_module()
    rec = {"roofline": {"frac": 0.5, "paper_mode": {"kernel_ms": 0.044, "frac": 0.38, "stream_ms_per_call": 0.045},
                        "merge_p1": {"kernel_ms": 0.042, "frac": 0.40}, "training_form": {"error": "x"}},
           "extras": {"paper_loss_step_64clips_graph_replay": {"ms": 0.129}, "paper_loss_step_64clips": {"ms": 0.337},
                      "mssloss_forward_backward_64clips_graph_replay": {"ms": 0.039}, "b4096n2048_per_row_positions_forward": {"ms": 0.14},
                      "b4096n2048_segmented_sort": {"ms": 0.038}, "rccl_world1_ms_per_step": 0.0335, "sot_slice_forward_backward_64clips_graph_replay": {"error": "y"}},
           "cpu_baseline": {"value": 1e4, "paper_mode": {"value": 9.2e3, "scalar": 3.7e-4}}}
    b.flatten_for_scalar_readers(rec, 8192, 2048)
    assert rec["roofline"]["paper_mode_kernel_ms"] == 0.044 and rec["roofline"]["paper_mode_frac"] == 0.38
    assert rec["roofline"]["merge_p1_frac"] == 0.40 and rec["roofline"]["training_form_kernel_ms"] is None and rec["roofline"]["training_form_frac"] is None
    assert rec["paper_step_64clips_graph_ms"] == 0.129 and rec["paper_step_64clips_eager_ms"] == 0.337 and rec["mss_64clips_graph_ms"] == 0.039
    assert rec["per_row_forward_ms"] == 0.14 and rec["segmented_sort_ms"] == 0.038 and rec["sot_slice_64clips_graph_ms"] is None
    assert rec["roofline"]["paper_step_64clips_graph_ms"] == 0.129            # the second copy, inside an object a scalars-only reader keeps
    assert rec["cpu_baseline"]["paper_mode_rows_per_s"] == 9.2e3
    for key in ("paper_step_64clips_graph_ms", "paper_step_64clips_eager_ms", "mss_64clips_graph_ms", "per_row_forward_ms", "rccl_world1_ms_per_step"):
        assert key in rec and not isinstance(rec[key], dict)


def test_the_line_says_what_rccl_saw():
    """config.rccl_world_size / config.rccl_allreduce_ones (an all-reduce of ones before the timed region == the number of ranks): in the source of
    the N > 1 path, since no multi-GPU node runs these tests."""
    src = open(BENCH).read()
    assert "rccl_world_size" in src and "rccl_allreduce_ones" in src and "dist.all_reduce(ones)" in src
    assert src.index("dist.all_reduce(ones)") < src.index("def native_step")    # before anything is timed
