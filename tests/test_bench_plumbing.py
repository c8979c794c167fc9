"""bench.py's control flow on CPU (tests/bench_cpu_shim.py stands the oracle in for the HIP library): the one JSON line, the profile legs in a
child process, and the fallbacks when that child fails or hangs -- the bench line must survive a failure of the instrumented legs."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "tests", "bench_cpu_shim.py")
ARGS = ["--pts", "6000", "--steps", "3", "--warmup", "1", "--profile-scans", "1", "--cpu-seconds", "0.3", "--map-voxels", "1000"]


def _run(script, extra=(), env=None, timeout=600):
    e = dict(os.environ)
    e["TMPDIR"] = e.get("TMPDIR", "/tmp")
    if env:
        e.update(env)
    r = subprocess.run([sys.executable, script] + ARGS + list(extra), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout, env=e, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    return json.loads(lines[0])


def _check_line(d):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["value"] > 0 and d["steps"] == 3 and d["n_gpus"] == 1 and d["unit"] == "scans/s" and d["vs_baseline"] is None
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["value"] > 0
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "source"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6


def test_profile_child_success():
    d = _run(SHIM)                                            # the child is the same (shimmed) script with --profile-child 1
    _check_line(d)
    assert "live" in d["roofline"]["source"] and "profile_leg_note" not in d
    assert d["stages_ms_serial"]["mesh"] == 0.5 and d["kernels_ms_per_scan"]["mesh_delaunay_kernel<256>"] == 0.25


def test_profile_inproc():
    d = _run(SHIM, ["--profile-inproc", "1"])
    _check_line(d)
    assert "live" in d["roofline"]["source"]


def test_profile_child_failure_falls_back(tmp_path):
    # a child that dies: the parent still prints its line, with the roofline taken from the committed rocprofv3 stats and labelled as such
    bad = tmp_path / "bench_bad_child.py"
    bad.write_text(open(SHIM).read().replace('if __name__ == "__main__":\n    bench.main()',
                   'if __name__ == "__main__":\n    if "--profile-child" in sys.argv:\n        os._exit(3)\n    bench.main()').replace(
                   'ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))', f'ROOT = {ROOT!r}'))
    d = _run(str(bad))
    _check_line(d)
    assert "committed rocprofv3" in d["roofline"]["source"] and "rc 3" in d["profile_leg_note"]


def test_only_the_profiler_child_fails(tmp_path):
    # the uninstrumented stage-timing child still delivers; only the roofline falls back
    bad = tmp_path / "bench_bad_prof.py"
    bad.write_text(open(SHIM).read().replace('if __name__ == "__main__":\n    bench.main()',
                   'if __name__ == "__main__":\n    if "--profile-child" in sys.argv and sys.argv[sys.argv.index("--profile-child") + 1] == "2":\n        os._exit(5)\n    bench.main()').replace(
                   'ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))', f'ROOT = {ROOT!r}'))
    d = _run(str(bad))
    _check_line(d)
    assert "profiler child failed (rc 5)" in d["profile_leg_note"] and "stage-timing" not in d["profile_leg_note"]
    assert d["stages_ms_serial"]["mesh"] == 0.5 and "committed rocprofv3" in d["roofline"]["source"]


def test_profile_child_hang_times_out(tmp_path):
    bad = tmp_path / "bench_hang_child.py"
    bad.write_text(open(SHIM).read().replace('if __name__ == "__main__":\n    bench.main()',
                   'if __name__ == "__main__":\n    if "--profile-child" in sys.argv:\n        import time\n        time.sleep(600)\n    bench.main()').replace(
                   'ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))', f'ROOT = {ROOT!r}'))
    d = _run(str(bad), ["--profile-timeout", "3"])
    _check_line(d)
    assert "timed out" in d["profile_leg_note"] and "committed rocprofv3" in d["roofline"]["source"]


def test_two_rank_replicas_gloo():
    """the N > 1 launch contract (torch.distributed.run, one rank per GPU, barrier + max-over-ranks timing, rank 0 prints) with gloo on CPU"""
    port = 29600 + (os.getpid() % 300)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           SHIM, "--gpus", "2", "--backend", "gloo"] + ARGS
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    d = json.loads(lines[0])
    _check_line_n = dict(d); _check_line_n["n_gpus"] = 1
    _check_line(_check_line_n)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "2 independent scan streams" in d["config"]["parallelism"]
    assert abs(d["value"] - 2 * d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)) < 1e-2 * d["value"]      # whole-job throughput = all ranks' scans / slowest rank's time
