"""CPU: `python bench.py --gpus 2` with no WORLD_SIZE starts its two ranks itself (the driver's command form), they rendezvous over gloo
on 127.0.0.1, run the barrier + max-over-ranks timing protocol, and rank 0 prints ONE JSON line with n_gpus == 2.  The dry rehearsal
runs no hot path (there is no GPU here, and the product has no CPU path): it checks the launcher, not the benchmark."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *args):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(AID_BENCH_REHEARSAL="dry", **extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=300)


def test_bench_self_launches_two_ranks():
    r = _run({}, "--gpus", "2", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["config"]["global_batch"] == 512
    assert out["steps"] == 3 and out["scaling"] == "weak" and out["value"] is None and "not a result" in out["data"]


def test_bench_self_launch_propagates_a_rank_failure():
    r = _run({"AID_BENCH_DRY_FAIL_RANK": "1"}, "--gpus", "2", "--steps", "1")
    assert r.returncode != 0
    assert "rank 1 exited" in r.stderr


def test_bench_under_an_external_launcher_keeps_working():
    """torch.distributed.run sets WORLD_SIZE: bench.py must then NOT spawn anything and act as the rank it was given."""
    env = {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}
    r = _run(env, "--gpus", "1", "--steps", "2")
    assert r.returncode == 0, r.stderr
    assert json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])["n_gpus"] == 1
    r = _run(env, "--gpus", "2", "--steps", "2")       # inconsistent: refuse rather than silently mislabel the line
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def _line(r):
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_eight_ranks_weak_and_strong_self_launched():
    """The two curves the first 8-GPU lease must yield without a code change: weak (BASELINE configs[4]: 8 x 256 rows) and strong
    (configs[1]'s 256-row batch split 8 ways), each with the ranks the communicator really had."""
    out = _line(_run({}, "--gpus", "8", "--steps", "2", "--warmup", "1"))
    assert out["n_gpus"] == 8 and out["scaling"] == "weak" and out["config"]["global_batch"] == 2048 and out["config"]["rows_per_rank"] == 256
    assert out["data_parallel"]["rccl_ranks_seen"] == 8
    out = _line(_run({}, "--gpus", "8", "--steps", "2", "--warmup", "1", "--scaling", "strong"))
    assert out["n_gpus"] == 8 and out["scaling"] == "strong" and out["config"]["global_batch"] == 256 and out["config"]["rows_per_rank"] == 32
    assert out["data_parallel"]["rccl_ranks_seen"] == 8 and "global batch=256" in out["unit"]
    r = _run({}, "--gpus", "3", "--steps", "1", "--scaling", "strong")          # 3 does not divide 256: refuse, every rank
    assert r.returncode != 0 and "do not divide" in (r.stderr + r.stdout)


def test_bench_eight_ranks_under_torch_distributed_run():
    """The driver's launch form for N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ..."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(AID_BENCH_REHEARSAL="dry", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--scaling", "strong"],
                       env=env, capture_output=True, text=True, timeout=600)
    out = _line(r)
    assert out["n_gpus"] == 8 and out["scaling"] == "strong" and out["data_parallel"]["rccl_ranks_seen"] == 8 and out["config"]["rows_per_rank"] == 32


def test_bench_line_is_compact_strict_json():
    """The driver reads ONE stdout line through a bounded window (round 4: a 25.9 KB line came back `parsed: null`).  Build the line from a canned
    full result (round 4's, committed under profiles/) and check: < 8000 characters, strict JSON (no NaN / Infinity tokens), the contract keys,
    `roofline` for the dominant kernel only with `traffic`, `cpu_baseline`; the bulky objects stay out."""
    import importlib.util
    import math
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench.json")))
    full["data_parallel"] = {"rccl_ranks_seen": 8, "backend": "nccl", "per_rank_ms_per_step": [50.123456789] * 8, "per_rank_ms_per_step_dead_belief_elided": [44.1] * 8,
                             "rows_per_rank": 256, "scaling": "weak", "belief_loop_schedule": "chains", "belief_loop_schedule_ms": {"chains": 1.0, "merged": 2.0}}
    full["roofline"]["frac_rocprof"] = 0.27; full["roofline"]["junk"] = "x" * 5000
    full["train_step_live_flop"] = float("nan")          # a non-finite value must not reach the line as a bare NaN token
    txt = bench.compact_line(full)
    assert "\n" not in txt and len(txt) < 8000, len(txt)
    line = json.loads(txt, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "value_dead_belief_elided", "train_step_frac_of_f32_mfma_peak", "data_parallel"):
        assert k in line, k
    assert line["config"]["workload"] and "model" not in line["config"]
    r = line["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(r) and "junk" not in r and "note" not in r
    assert math.isclose(r["frac"], r["achieved"] / r["peak"], rel_tol=1e-4)
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(line["cpu_baseline"])
    for k in ("roofline_phases", "roofline_kernels", "other_configs", "small_batch", "roofline_single_stream_kernels"):
        assert k not in line
    assert set(line["other_configs_ms"]) >= {"cfg1_b32", "cfg3_hopper_b1024", "cfg4_pixel_b256"}
    assert "train_step_live_flop" not in line or line["train_step_live_flop"] is None
