"""bench.py's control flow on CPU (tests/bench_cpu_shim.py stands the oracle in for the HIP library): the one JSON line, the instrumented legs on the
same context, the fallbacks when they fail or hang (the bench line must survive), the N > 1 launch contract and `--gpus N` spawning its ranks."""
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


def test_instrumented_legs_in_process():
    d = _run(SHIM, ["--extra-configs", "0"])                  # the legs run on the same context, right after the timed region
    _check_line(d)
    assert "live" in d["roofline"]["source"] and "profile_leg_note" not in d
    assert d["stages_ms_serial"]["mesh"] == 0.5 and d["kernels_ms_per_scan"]["mesh_delaunay64_kernel"] == 0.25
    assert set(d["roofline"]["counters_of_the_profiled_scans"]) >= {"n_u", "t_v", "n_v", "c20", "n_refit_pts", "n_app", "c1"}   # frac is recomputable from the line
    cb = d["cpu_baseline"]
    assert cb["reference_threading"]["stages_ms_p50"]["register"] > 0 and cb["all_cores"]["scans"] >= 3


def _variant(tmp_path, name, patch):
    bad = tmp_path / name
    bad.write_text(open(SHIM).read().replace('if __name__ == "__main__":\n    bench.main()', patch + '\nif __name__ == "__main__":\n    bench.main()').replace(
                   'ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))', f'ROOT = {ROOT!r}'))
    return str(bad)


def test_profiler_leg_failure_falls_back(tmp_path):
    # the HIP-event leg raises: the line is still printed, the roofline comes from the committed rocprofv3 stats and says so
    script = _variant(tmp_path, "bench_bad_prof.py", "def _boom(self, reset=False):\n    raise RuntimeError('profiler boom')\n_OracleHotPath.profile_read = _boom\n")
    d = _run(script, ["--extra-configs", "0"])
    _check_line(d)
    assert "committed rocprofv3" in d["roofline"]["source"] and "profiler boom" in d["profile_leg_note"]
    assert d["stages_ms_serial"]["mesh"] == 0.5          # the uninstrumented stage timing had already been delivered


def test_hanging_leg_is_cut_by_the_watchdog(tmp_path):
    script = _variant(tmp_path, "bench_hang.py", "import time as _t\ndef _hang(self, reset=False):\n    _t.sleep(600)\n_OracleHotPath.profile_read = _hang\n")
    d = _run(script, ["--extra-configs", "0", "--profile-timeout", "1", "--cpu-seconds", "0.2"], timeout=300)
    for k in ("metric", "value", "roofline"):
        assert k in d
    assert "watchdog" in d["profile_leg_note"] and "committed rocprofv3" in d["roofline"]["source"] and d["value"] > 0


def test_gpus_flag_spawns_the_ranks():
    """`python bench.py --gpus 2` (not under torch.distributed.run) re-executes itself as 2 ranks; the line carries both legs"""
    r = subprocess.run([sys.executable, SHIM, "--gpus", "2", "--backend", "gloo"] + ARGS + ["--extra-configs", "0", "--profile-scans", "0", "--cpu-seconds", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    d = json.loads(lines[0])
    # N > 1: the headline is the SHARDED job (one stream, strong scaling); the independent replicas are reported beside it
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["sharded"]["scaling"] == "strong" and d["value"] == d["sharded"]["value"] > 0
    assert d["replicas"]["scaling"] == "weak" and d["replicas"]["value"] > 0 and "bricks" in d["config"]["parallelism"]
    assert set(d["sharded"]["load_balance_point_share_per_brick_size"]) == {"8", "16", "32"}


def test_two_rank_replicas_gloo():
    """the N > 1 launch contract (torch.distributed.run, one rank per GPU, barrier + max-over-ranks timing, rank 0 prints) with gloo on CPU"""
    port = 29600 + (os.getpid() % 300)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           SHIM, "--gpus", "2", "--backend", "gloo", "--extra-configs", "0", "--sharded-leg", "0"] + ARGS   # (the replica leg alone)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    d = json.loads(lines[0])
    assert d["cpu_baseline"] is None and "N = 1 only" in d["cpu_baseline_note"]      # the CPU leg belongs to the N = 1 line
    d1 = _run(SHIM, ["--extra-configs", "0"])
    _check_line_n = dict(d); _check_line_n["n_gpus"] = 1; _check_line_n["cpu_baseline"] = d1["cpu_baseline"]
    _check_line(_check_line_n)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "2 independent scan streams" in d["config"]["parallelism"]
    assert abs(d["value"] - 2 * d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)) < 1e-2 * d["value"]      # whole-job throughput = all ranks' scans / slowest rank's time


def test_kernel_source_fingerprint_ignores_comments_and_the_committed_traffic_file_is_current(tmp_path, monkeypatch):
    """roofline.traffic is only reported from a PMC file measured on THESE kernel sources (bench.kernel_sources_sha: the compiled text, comments and
    layout removed).  A comment or blank line must not move the fingerprint, a token must; and the file under profiles/ should be the current one
    (a stale file is not an error -- bench.py then reports traffic: null and says why -- but it is worth a visible skip)."""
    import shutil
    import pytest
    sys.path.insert(0, ROOT)
    import bench
    src = os.path.join(ROOT, "immesh_amd", "csrc")
    work = tmp_path / "immesh_amd" / "csrc"
    work.mkdir(parents=True)
    for name in os.listdir(src):
        if name.endswith((".hip", ".inc", ".hpp", ".cpp")):
            shutil.copy(os.path.join(src, name), work / name)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    base = bench.kernel_sources_sha()
    victim = work / "ds_kernels.hip"
    text = victim.read_text()
    victim.write_text("// a remark\n\n" + text.replace("\n", "\n   ", 3) + "\n/* another\n   one */\n")
    assert bench.kernel_sources_sha() == base
    victim.write_text(text.replace("#define DSH_CHUNK 512", "#define DSH_CHUNK 256"))
    assert bench.kernel_sources_sha() != base
    victim.write_text(text)
    host = work / "mesh_host.cpp"                       # the host layer decides which kernels run on what: part of the fingerprint
    host.write_text(host.read_text().replace("round > 4096", "round > 4097"))
    assert bench.kernel_sources_sha() != base
    victim.write_text(text.replace("#define DSH_CHUNK 512\n", "#define DSH_CHUNK 512 \\\n"))   # a line end that ends a #define is not layout
    assert bench.kernel_sources_sha() != base
    monkeypatch.setattr(bench, "ROOT", ROOT)
    assert bench.kernel_sources_sha() == base
    meta = json.load(open(os.path.join(ROOT, "profiles", bench.COMMITTED_TRAFFIC)))["_meta"]
    if meta["kernel_sources_sha16"] != base:
        pytest.skip(f"profiles/{bench.COMMITTED_TRAFFIC} was measured on {meta['kernel_sources_sha16']}, the kernels are {base}: tools/r04_traffic.sh regenerates it")
