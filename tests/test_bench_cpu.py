"""CPU tests of bench.py's launcher decision: `python bench.py --gpus N` must start its N ranks by itself, before anything
touches the GPU, and a worker whose WORLD_SIZE disagrees with --gpus must fail instead of printing a mislabelled line."""
import importlib.util
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_module", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_single_gpu_and_workers_do_not_launch(bench):
    assert bench.launcher_argv(1, {}, []) is None
    assert bench.launcher_argv(8, {"WORLD_SIZE": "8", "RANK": "3", "LOCAL_RANK": "3"}, ["--gpus", "8"]) is None
    assert bench.launcher_argv(2, {"RANK": "0"}, []) is None


def test_launcher_command_line(bench):
    argv = bench.launcher_argv(8, {"PATH": "/usr/bin"}, ["--gpus", "8", "--steps", "20", "--warmup", "2"], port=29517)
    assert argv[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in argv and "--nnodes=1" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and argv[argv.index("--master-port") + 1] == "29517"
    k = argv.index(str(ROOT / "bench.py"))
    assert argv[k + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "2"]
    port = int(bench.launcher_argv(2, {}, [])[8])
    assert 1024 < port < 65536


def test_world_size_must_equal_gpus(bench):
    bench.check_world(4, 4)
    with pytest.raises(SystemExit):
        bench.check_world(8, 1)
    with pytest.raises(SystemExit):
        bench.check_world(1, 2)


def test_launcher_process_never_imports_torch_and_propagates_failure(tmp_path):
    """Run the launcher for real with a stand-in `torch.distributed.run` first on the module path: the parent must not have
    imported torch (so it cannot have initialised the GPU) and must exit with the child's code."""
    pkg = tmp_path / "torch" / "distributed"
    pkg.mkdir(parents=True)
    (tmp_path / "torch" / "__init__.py").write_text("")
    (pkg / "__init__.py").write_text("")
    (pkg / "run.py").write_text("import sys, json\nprint(json.dumps(sys.argv[1:]))\nsys.exit(7)\n")
    probe = ("import sys, runpy\n"
             f"sys.argv = [{str(ROOT / 'bench.py')!r}, '--gpus', '2', '--steps', '1']\n"
             "try:\n    runpy.run_path(sys.argv[0], run_name='__main__')\n"
             "except SystemExit as e:\n    print('EXIT', e.code, 'torch' in sys.modules)\n")
    env = {"PATH": "/usr/bin:/bin", "PYTHONPATH": str(tmp_path)}
    out = subprocess.run([sys.executable, "-c", probe], env=env, capture_output=True, text=True, timeout=120)
    assert "EXIT 7 False" in out.stdout, out.stdout + out.stderr
    assert "--nproc-per-node=2" in out.stdout and "--gpus" in out.stdout


def test_profile_figures_are_quoted_only_for_the_running_build(bench, tmp_path):
    """The roofline block's issue_frac / useful_lane_frac / traffic come from committed PMC summaries; they must be nulled and marked
    stale when the summary was taken on another build than the library that is running."""
    import json
    pmc = {"provenance": {"build": "abc123"}, "counters_mean_per_launch": {"SQ_INSTS_VALU": 270e9, "SQ_INSTS_SALU": 74e9, "SQ_INSTS_VMEM_RD": 6e9,
           "SQ_INSTS_LDS": 7e9}, "valu": {"kernel_cycles": 9.8e8, "lanes_active_frac": 0.65, "wave_time_split": {"issuing": 0.4}}}
    (tmp_path / "r07_pmc_fused_kernel.json").write_text(json.dumps(pmc))
    (tmp_path / "r07_traverse_traffic.json").write_text(json.dumps({"provenance": {"build": "abc123"}, "fabric_bytes_per_launch": 1.4e12, "tcc_hit_rate": 0.78}))
    (tmp_path / "r06_pmc_fused_kernel.json").write_text(json.dumps({"provenance": {"build": "old"}, "valu": {}}))      # an older round: ignored
    traffic, src, issue = bench.profile_fields(tmp_path, "abc123")
    assert traffic == 1.4e12 and src["stale"] is False and src["file"] == "profiles/r07_traverse_traffic.json"
    assert issue["stale"] is False and 0.8 < issue["issue_frac"] < 0.9 and abs(issue["useful_lane_frac"] - issue["issue_frac"] * 0.65) < 1e-3
    for running in ("def456", None):
        traffic, src, issue = bench.profile_fields(tmp_path, running)
        assert traffic is None and src["stale"] is True and src["tcc_hit_rate"] is None
        assert issue["stale"] is True and "issue_frac" not in issue and issue["build"] == "abc123"
    assert bench.profile_fields(tmp_path / "missing", "abc123") == (None, None, None)
    # the committed summaries of this round belong to the committed sources
    root = ROOT / "profiles"
    newest = sorted(root.glob("r*_pmc_fused_kernel.json"))[-1]
    assert json.loads(newest.read_text())["provenance"]["build"]
