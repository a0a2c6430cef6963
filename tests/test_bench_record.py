"""bench.py's contract line: the driver keeps the last 8 KB of stdout and parses the LAST line.  Round 4's line had grown to
20 KB and was never parsed; these tests pin its size and keys (CPU: a canned measure() result; GPU: the real command)."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

CONTRACT_KEYS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]
ROOFLINE_KEYS = ["bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_rev", "traffic_pose", "peak_note", "algorithmic_tflops",
                 "executed_mfma_tflops", "executed_frac_of_instruction_peak"]
CPU_KEYS = ["value", "unit", "cores", "kind", "sample", "chunks"]


def _canned_leg(precision, steps, warmup, l=3, ms_stage=413.5):
    """What bench.measure() returns for a 1080p, 2-performer view: numbers of the size the MI355X produces, at full double
    precision (the worst case for the line's length)."""
    launches = 2 * 4 * steps
    evals = 4.123456789e8 * steps
    ksum = {
        "mlp_stage": dict(launches=launches, ms=ms_stage * launches, evals=evals, flop=evals * 1.0123456789e6),
        "sample_coarse": dict(launches=4 * steps, ms=1.23456789 * steps, bytes=6.123456789e9 * steps, bytes_dense=6.5e9 * steps),
        "composite": dict(launches=8 * steps, ms=6.43210987 * steps, bytes=1.7054321e10 * steps, bytes_dense=2.2e10 * steps),
        "resample": dict(launches=4 * steps, ms=2.96117281 * steps, bytes=1.0123456e10 * steps, bytes_dense=1.3e10 * steps),
        "generate_rays": dict(launches=steps, ms=0.0312345 * steps, bytes=7.4649600e7 * steps, bytes_dense=7.4649600e7 * steps),
    }
    elapsed = 3.3212345678 * steps
    return dict(elapsed=elapsed, ksum=ksum, evals=evals, evals_all=evals, mask_fraction=[1.0] + [0.1612345678] * (l - 1),
                per_rank_compute_s=[elapsed], steps=steps, warmup=warmup, precision=precision,
                dims=(1080, 1920, l - 1, 64, 64, True, True), rays=1080 * 1920 * steps)


# what tools/summarise.py writes beside the counters: the build they were measured on and the pose
CANNED_PMC = {"workload": "taekwondo-1080p-64+64", "rev": "abc1234", "fatbin_sha256": "0" * 64, "pose_short": "pose 0 (orbit 10 deg), 1 step",
              "kernels_bf16x3": {"mlp_stage": {"hbm_bytes_per_launch": 3.2021812345e9, "hbm_bytes_per_step": 2.56e10}},
              "kernels": {"mlp_stage": {"hbm_bytes_per_launch": 3.1e9, "hbm_bytes_per_step": 2.48e10}}}


def _ctx(bench, **over):
    cpu = dict(value=526.123456789, unit="rays/s", cores=32, kind="port",
               sample="5 reference chunks of 3584 rays spread evenly over the rows of the 1920x1080 view (the performer boxes span the "
                      "image height: every chunk crosses them, see chunks[].performer_hit_fraction), oracle/stnerf_oracle.py on torch "
                      "2.10.0+rocm7.0 CPU fp32, 34.1 s",
               sample_short="5 reference chunks of 3584 rays spread over the 1920x1080 view, oracle/stnerf_oracle.py, torch 2.10.0+rocm7.0 CPU fp32",
               seconds=34.0612345, ray_samples_per_s=1.23456789e5, extrapolated_frame_seconds=3941.123,
               host=dict(nproc=256, torch_threads=32, cpu="AMD EPYC 9575F 64-Core Processor"),
               chunks=[dict(first_row=i * 130, seconds=6.8, rays_per_s=527.1, performer_hit_fraction=0.16) for i in range(8)])
    ctx = dict(workload="taekwondo-1080p-64+64", precision="bf16x3", steps=20, warmup=5, world=1, partition="stripes",
               stripe_rows=1, rays_per_launch=1 << 19, dims=(1080, 1920, 2, 64, 64, True, True),
               device={"name": "gfx950:sramecc+:xnack-", "cus": 256, "note": "x" * 300},
               psnr={"reference_seed_b_vs_seed_a_dB": 41.123456, "hip_device_rng_bf16x3_vs_reference_seed_a_dB": 41.2345678,
                     "hip_device_rng_fp32_vs_reference_seed_a_dB": 41.2345679, "view": "128x128", "fixture": "tests/golden/psnr_view.npz"},
               pmc=CANNED_PMC, pmc_same_build=True,
               hbm_microbench={"read_GBps": 6498.2, "copy_GBps": 5100.0, "write_GBps": 5200.0},
               eager={"value": 1.1e5, "unit": "rays/s", "kind": "eager", "sample": "3584 rays"}, cpu=cpu,
               share={"note": "emulation", "t1_ms": 3321.0, "steps": 3, "stripe_rows": 1,
                      "shares": [{"ranks": n, "t_share_ms": {"0": 3321.0 / n}, "t_share_max_ms": 3400.0 / n, "n_times_t_share_over_t1": 1.02,
                                  "predicted_compute_efficiency": 0.9812345678, "rays_per_rank": 2073600 // n,
                                  "gather_payload_bytes_per_rank": {"all": 1, "fine": 1, "final": 1}} for n in (2, 4, 8)]})
    ctx.update(over)
    return ctx


def test_final_line_is_small_and_carries_the_contract(capsys, tmp_path, monkeypatch):
    import bench
    monkeypatch.setattr(bench, "DETAIL_PATH", str(tmp_path / "bench_detail.json"))
    head, second = _canned_leg("bf16x3", 20, 5), _canned_leg("fp32", 5, 2, ms_stage=596.4)
    legs = []
    for wl, l in (("walking-1080p-L4-64+64", 5), ("single-512-64+64", 2)):
        leg = _canned_leg("bf16x3", 2, 1, l=l)
        leg["workload"] = wl
        legs.append(leg)
    detail, final = bench.build_records(head, second, legs, _ctx(bench))
    bench.emit(detail, final)
    lines = capsys.readouterr().out.strip().split("\n")
    assert len(lines) == 2                                     # the detail digest, then the contract record LAST
    assert len(lines[-1]) < bench.FINAL_LINE_LIMIT == 3072
    assert len(lines[0]) <= bench.DETAIL_LINE_LIMIT and len(lines[0]) + len(lines[1]) < 8000   # both inside the driver's 8 KB tail
    rec = json.loads(lines[-1])
    for k in CONTRACT_KEYS:
        assert k in rec, k
    assert all(k in rec["roofline"] for k in ROOFLINE_KEYS) and all(k in rec["cpu_baseline"] for k in CPU_KEYS)
    assert rec["n_gpus"] == 1 and rec["steps"] == 20 and rec["warmup"] == 5 and rec["higher_is_better"] is True
    assert rec["config"]["workload"] == "taekwondo-1080p-64+64" and rec["config"]["precision"] == "bf16x3"
    assert "model" not in rec["config"]
    assert rec["value"] == pytest.approx(1080 * 1920 / 3.3212345678, rel=1e-5)
    assert rec["ms_per_step"] == pytest.approx(3321.2345678, rel=1e-5)
    # achieved = ALGORITHMIC TF/s; peak = the ceiling for fp32-faithful products (2500 / 6 bf16x3 terms); frac = their ratio;
    # the executed MFMA rate stays under names that say so
    roof = rec["roofline"]
    assert roof["bound"] == "mfma" and roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], rel=1e-4)
    assert roof["achieved"] == roof["algorithmic_tflops"] and roof["peak"] == pytest.approx(2500.0 / 6, rel=1e-4) and "2500" in roof["peak_note"]
    assert roof["executed_mfma_tflops"] == pytest.approx(6 * roof["achieved"], rel=1e-4)
    assert roof["executed_frac_of_instruction_peak"] == pytest.approx(roof["executed_mfma_tflops"] / 2500.0, rel=1e-4)
    assert roof["frac"] == pytest.approx(roof["executed_frac_of_instruction_peak"], rel=1e-4)      # the two readings coincide by construction
    assert roof["traffic"] == pytest.approx(3.2021812345e9, rel=1e-5) and roof["traffic_rev"] == "abc1234" and "pose 0" in roof["traffic_pose"]
    assert rec["cpu_baseline"]["chunks"] == 8
    assert rec["other_precision"]["peak"] == pytest.approx(157.3)
    assert rec["other_precision"]["precision"] == "fp32" and rec["other_precision"]["steps"] == 5
    assert rec["vs_baseline"] is None and rec["scaling"] == "strong" and rec["unit"] == "rays/s"
    # everything else lives in the side file (and, digested, on the line before)
    side = json.load(open(tmp_path / "bench_detail.json"))
    assert side["final"] == rec
    assert set(side["detail"]["precision_legs"]) >= {"bf16x3", "fp32"} and len(side["detail"]["config_legs"]) == 2
    assert len(side["detail"]["cpu_baseline"]["chunks"]) == 8
    assert json.loads(lines[0])["record"] == "detail"


def test_traffic_is_null_when_the_counters_were_measured_on_another_build(tmp_path, monkeypatch):
    """`roofline.traffic` comes from committed PMC passes: it is printed only when the loaded library's kernels (.hip_fatbin digest) are
    the ones the passes ran on; the provenance keys stay so that the reader sees which build / pose the file holds."""
    import bench
    monkeypatch.setattr(bench, "DETAIL_PATH", str(tmp_path / "bench_detail.json"))
    detail, final = bench.build_records(_canned_leg("bf16x3", 3, 1), None, [], _ctx(bench, steps=3, warmup=1, pmc_same_build=False))
    assert final["roofline"]["traffic"] is None and final["roofline"]["traffic_rev"] == "abc1234"
    # without the override the digest of whatever library is in the tree decides: the canned file's digest matches nothing
    ctx = _ctx(bench, steps=3, warmup=1)
    ctx.pop("pmc_same_build")
    detail, final = bench.build_records(_canned_leg("bf16x3", 3, 1), None, [], ctx)
    assert final["roofline"]["traffic"] is None
    # another workload's counters are not this workload's
    detail, final = bench.build_records(_canned_leg("bf16x3", 3, 1), None, [], _ctx(bench, steps=3, warmup=1, workload="walking-1080p-L4-64+64"))
    assert final["roofline"]["traffic"] is None and final["roofline"]["traffic_rev"] is None


def test_an_empty_or_broken_profile_file_costs_nothing(tmp_path):
    """Round 6's first evidence pass died on a zero-byte profiles/r06_hbm_copy_microbench.json: side files are read forgivingly."""
    import bench
    empty, broken, fine = tmp_path / "a.json", tmp_path / "b.json", tmp_path / "c.json"
    empty.write_text("")
    broken.write_text("{not json")
    fine.write_text('{"read_GBps": 6400.0}')
    assert bench._load_profile(str(empty)) == {} and bench._load_profile(str(broken)) == {} and bench._load_profile(str(tmp_path / "none.json")) == {}
    assert bench._load_profile(str(fine)) == {"read_GBps": 6400.0}
    for f in os.listdir(os.path.join(REPO, "profiles")):           # and nothing of the kind is committed
        if f.endswith(".json"):
            assert os.path.getsize(os.path.join(REPO, "profiles", f)) > 2, f


def test_fatbin_digest_reads_the_section_of_the_built_library():
    from stnerf_amd import hip
    if not os.path.exists(hip.LIB_PATH):
        pytest.skip("library not built")
    d = hip.fatbin_sha256()
    assert isinstance(d, str) and len(d) == 64 and d == hip.fatbin_sha256(hip.LIB_PATH)
    assert hip.fatbin_sha256(__file__) is None                  # not an ELF


def test_final_line_sheds_optional_blocks_rather_than_outgrow_the_limit(capsys, tmp_path, monkeypatch):
    import bench
    monkeypatch.setattr(bench, "DETAIL_PATH", str(tmp_path / "bench_detail.json"))
    monkeypatch.setattr(bench, "FINAL_LINE_LIMIT", 1700)
    detail, final = bench.build_records(_canned_leg("bf16x3", 3, 1), _canned_leg("fp32", 3, 1), [], _ctx(bench, steps=3, warmup=1))
    bench.emit(detail, final)
    last = capsys.readouterr().out.strip().split("\n")[-1]
    rec = json.loads(last)
    assert len(last) < 1700 and all(k in rec for k in CONTRACT_KEYS)


def test_multi_gpu_record_has_no_cpu_leg_and_names_the_partition(tmp_path, monkeypatch):
    import bench
    monkeypatch.setattr(bench, "DETAIL_PATH", str(tmp_path / "bench_detail.json"))
    head = _canned_leg("bf16x3", 20, 5)
    head["per_rank_compute_s"] = [8.1, 8.2, 8.3, 8.25]
    detail, final = bench.build_records(head, None, [], _ctx(bench, world=4, cpu=None, eager=None, share=None, psnr=None))
    assert final["n_gpus"] == 4 and final["cpu_baseline"] is None and "stripes" in final["config"]["parallelism"]
    assert len(json.dumps(final)) < 3072 and "other_precision" not in final


def _bench_cli(args, env_extra=None):
    env = dict(os.environ, **(env_extra or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, capture_output=True, text=True, timeout=300, env=env, cwd=REPO)


@pytest.mark.skipif(__import__("torch").cuda.is_available() and __import__("torch").cuda.device_count() >= 4, reason="needs a box with fewer than 4 GPUs")
def test_more_ranks_than_devices_ends_with_one_json_error_line():
    """`python bench.py --gpus N` on a node with fewer than N visible devices: non-zero exit status and ONE parsable JSON line
    {"error": ...} as the last stdout line -- not a bare string, not a traceback, no ranks launched."""
    p = _bench_cli(["--gpus", "4", "--steps", "1", "--warmup", "0"])
    assert p.returncode != 0
    lines = [ln for ln in p.stdout.strip().split("\n") if ln.strip()]
    rec = json.loads(lines[-1])
    assert len(lines) == 1 and "error" in rec and "--gpus 4" in rec["error"] and rec["value"] is None
    # launched as a rank (the driver's torch.distributed.run) with a WORLD_SIZE that contradicts --gpus: rank 0 prints the record,
    # other ranks print nothing on stdout
    p0 = _bench_cli(["--gpus", "4"], dict(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    p1 = _bench_cli(["--gpus", "4"], dict(WORLD_SIZE="2", RANK="1", LOCAL_RANK="1"))
    assert p0.returncode != 0 and p1.returncode != 0
    assert "WORLD_SIZE=2" in json.loads(p0.stdout.strip().split("\n")[-1])["error"] and p1.stdout.strip() == ""


@pytest.mark.gpu
def test_bench_command_prints_a_parsable_last_line():
    """The driver's command, short: the LAST stdout line parses, is small, and carries the contract keys."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--cpu-baseline-rays", "3584", "--no-config-legs", "--emulate-share", "8"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=REPO)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.strip().split("\n") if ln.strip()]
    assert len(lines[-1]) < 3072, len(lines[-1])
    assert len(p.stdout) < 8000, len(p.stdout)
    rec = json.loads(lines[-1])
    for k in CONTRACT_KEYS:
        assert k in rec, k
    assert rec["steps"] == 2 and rec["warmup"] == 1 and rec["n_gpus"] == 1
    assert rec["value"] > 1e5 and 0.0 < rec["roofline"]["frac"] < 1.0 and rec["cpu_baseline"]["value"] > 0
    assert rec["roofline"]["achieved"] == rec["roofline"]["algorithmic_tflops"] and rec["roofline"]["peak"] == pytest.approx(416.667, rel=1e-3)
    assert rec["cpu_baseline"]["chunks"] == 1
    assert rec["value"] == pytest.approx(1080 * 1920 / (rec["ms_per_step"] * 1e-3), rel=1e-3)
    assert "8" in rec["share_emulation"]
