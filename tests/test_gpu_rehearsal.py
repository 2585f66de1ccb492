"""Rehearsal of the multi-rank benchmark paths on the one GPU of the test box: `bench.py --gpus 2 --one-device
--backend gloo` for every configuration that shards (c3 and c4 by image, c5 and c6 by bands of one image).  Two
ranks share GPU 0 and talk over gloo, so nothing here measures scaling; what is checked is that the planner, the
partition, the gather and the self-verification of every configuration still work with more than one rank, and
that the JSON line names the configuration and the rank count it ran with."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

COMMON = ["--gpus", "2", "--one-device", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]


def run_bench(*args):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):  # bench.py starts its own ranks
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *COMMON, *args], capture_output=True, text=True,
                         timeout=420, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 prints the one line
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["warmup"] == 1 and r["value"] > 0
    return r


def test_c3_two_ranks_batch_by_image():
    r = run_bench("--config", "c3", "--images", "48", "--distinct", "6", "--no-sensitivity", "--h2h-sample", "0")
    assert r["config"]["workload"].startswith("c3:") and r["scaling"] == "weak"
    assert r["config"]["images_per_gpu"] == 48
    v = r["verified_vs_oracle"]
    assert v["mismatches"] == 0 and v["copies_equal_their_source"] and v["distinct_images_checked"] == 6
    assert not any(r["fast_path"].values())


def test_c4_two_ranks_mixed_job_in_sub_batches():
    r = run_bench("--config", "c4", "--images", "16", "--sub-batch", "8", "--distinct", "2")
    assert r["config"]["workload"].startswith("c4:") and r["scaling"] == "weak"
    assert r["config"]["sub_batch_images"] == 8 and r["config"]["sub_batches"] >= 2
    v = r["verified_vs_oracle"]
    assert v["mismatches_on_rank0"] == 0 and v["all_ranks_ok"]


def test_c5_two_ranks_bands_of_a_restart_marked_image():
    r = run_bench("--config", "c5", "--c5-size", "2048")
    assert r["config"]["workload"].startswith("c5:") and r["scaling"] == "strong"
    assert len(r["config"]["bands_mcu_rows"]) == 2
    assert r["verified"]["equal"] and r["verified"]["equals_oracle"]


def test_c6_two_ranks_image_without_restart_markers():
    """The stream split by bytes: each rank unstuffs and walks its own part plus the lead-in only."""
    r = run_bench("--config", "c6")
    assert r["config"]["workload"].startswith("c6:") and r["scaling"] == "strong"
    assert "split by bytes" in r["config"]["workload"] and "FELL BACK" not in r["config"]["workload"]
    assert r["config"]["scan_bytes_read_by_rank0"] < 0.6 * r["config"]["scan_bytes"]
    wins = r["config"]["mcus_first_count_per_rank"]
    assert len(wins) == 2 and wins[0][0] == 0 and wins[1][0] == wins[0][1] and wins[0][1] + wins[1][1] == (7680 // 8) * (4320 // 8)
    assert r["verified"]["equal"] and r["verified"]["equals_oracle"]


def test_c6_two_ranks_falls_back_when_the_parts_do_not_join_up():
    """A lead-in of 16 bytes is shorter than any stream needs to synchronise: every rank learns from the same reports
    that the chain does not validate, and the run repeats the entropy front on every rank instead."""
    r = run_bench("--config", "c6", "--lead-bytes", "16")
    assert "FELL BACK" in r["config"]["workload"]
    assert r["verified"]["equal"] and r["verified"]["equals_oracle"]


def test_c6_two_ranks_row_windows_on_request():
    r = run_bench("--config", "c6", "--c6-rows")
    assert "entropy front repeated" in r["config"]["workload"]
    assert r["verified"]["equal"] and r["verified"]["equals_oracle"]
