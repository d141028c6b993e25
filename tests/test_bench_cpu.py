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
