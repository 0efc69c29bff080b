"""CPU-side checks of `python bench.py --gpus N` starting its own ranks:
the torchrun command it builds, and the relay of rank 0's JSON line and of the child's exit code."""
import json
import os
import subprocess
import sys
import tempfile

from conftest import ROOT

sys.path.insert(0, ROOT)
BENCH = os.path.join(ROOT, "bench.py")


def _bench():
    import importlib
    return importlib.import_module("bench")


def test_launch_command_is_torchrun_with_the_same_arguments():
    b = _bench()
    argv = ["--gpus", "4", "--steps", "20", "--warmup", "5", "--no-cpu"]
    cmd = b.launch_command(argv, 4, 29777)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29777"
    i = cmd.index(BENCH)
    assert cmd[i + 1:] == argv
    p = b.free_port()
    assert 1024 < p < 65536


def test_self_launch_relays_one_json_line_and_the_exit_code(capfd):
    b = _bench()
    line = json.dumps({"metric": "scenes/sec", "value": 1.5, "n_gpus": 2})
    with tempfile.TemporaryDirectory() as td:
        child = os.path.join(td, "child.py")
        with open(child, "w") as f:
            f.write("import sys\n"
                    "print('rank chatter')\n"
                    "print('{\"not\": \"the result\"}')\n"
                    f"print({line!r})\n"
                    "sys.stderr.write('warn\\n')\n"
                    "sys.exit(3)\n")
        rc = b.self_launch([sys.executable, child])
        out, err = capfd.readouterr()
        assert rc == 3
        assert out.strip() == line                        # exactly one line on stdout: rank 0's result
        assert "rank chatter" in err and "not" in err
        quiet = os.path.join(td, "quiet.py")
        with open(quiet, "w") as f:
            f.write("print('no result here')\n")
        rc = b.self_launch([sys.executable, quiet])
        out, err = capfd.readouterr()
        assert rc == 1 and out == "" and "no result line" in err


def test_parent_launches_before_importing_torch():
    """`python bench.py --gpus 2` without WORLD_SIZE must start its ranks before anything imports torch (and with it
    the GPU runtime) in the parent: run it with a stub `torch` on PYTHONPATH whose import FAILS in the parent process
    and whose `torch.distributed.run` prints the arguments it was started with."""
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, "torch", "distributed"))
        with open(os.path.join(td, "torch", "__init__.py"), "w") as f:
            f.write("import os\n"
                    "if os.environ.get('SAD_TEST_PARENT_PID') == str(os.getpid()):\n"
                    "    raise ImportError('the parent imported torch')\n")
        open(os.path.join(td, "torch", "distributed", "__init__.py"), "w").close()
        with open(os.path.join(td, "torch", "distributed", "run.py"), "w") as f:
            f.write("import json, sys\n"
                    "print(json.dumps({'metric': 'm', 'value': 2.0, 'argv': sys.argv[1:]}))\n")
        parent = os.path.join(td, "parent.py")
        with open(parent, "w") as f:
            f.write("import os, runpy, sys\n"
                    "os.environ['SAD_TEST_PARENT_PID'] = str(os.getpid())\n"
                    f"sys.argv = [{BENCH!r}, '--gpus', '2', '--steps', '20', '--warmup', '5']\n"
                    f"runpy.run_path({BENCH!r}, run_name='__main__')\n")
        env = dict(os.environ, PYTHONPATH=td)
        env.pop("WORLD_SIZE", None)
        r = subprocess.run([sys.executable, parent], env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-2000:]
        rec = json.loads(r.stdout.strip())
        assert rec["argv"][:2] == ["--nnodes=1", "--nproc-per-node=2"]
        assert rec["argv"][-6:] == ["--gpus", "2", "--steps", "20", "--warmup", "5"]


def test_bench_batch_rotation_and_workload_defaults():
    """Round 4: the timed region rotates over distinct resident batches; no scene is shared between the batches of a rank or
    between ranks; the secondary legs keep their own stream plan.  Pure host logic (no torch, no GPU)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    B, world = 32, 8
    seen = set()
    for rank in range(world):
        for k in range(4):
            first = b.batch_first_scene(k, rank, world, B)
            scenes = set(range(first, first + B))
            assert not (scenes & seen)
            seen |= scenes
    assert len(seen) == world * 4 * B
    w = b.Workload("kitti", "kitti", "f32", 32)
    assert (w.fps_streams, w.main_streams, w.queue_depth, w.n_batches) == (8, 2, 10, 4)     # (eight sampling streams since the stream placement: DESIGN 5)
    b.DISTRIBUTED = True          # under a process group: a smaller stream set (the backend needs hardware queues of its own)
    w = b.Workload("kitti", "kitti", "f32", 32)
    assert (w.fps_streams, w.queue_depth) == (6, 8)
    b.DISTRIBUTED = False
    w = b.Workload("nuscenes", "kitti", "bf16", 32, n_batches=2)
    assert (w.fps_streams, w.queue_depth, w.n_batches) == (8, 10, 2) and w.peak() == b.PEAK_MFMA_BF16_TFLOPS
    assert "nuScenes" in w.describe() and "configs[4]" in w.describe()
    res = {"metric": "m", "value": 1.0, "unit": "scenes/s", "steps": 3, "warmup": 1, "ms_per_step": 2.0, "dtype": "bf16",
           "config": {"workload": "w", "fps_streams": 6, "scenes_per_gpu": 32},
           "roofline": {"bound": "mfma", "achieved": 1.0, "peak": 2.0, "unit": "TFLOP/s", "frac": 0.5, "traffic": None, "flop_per_step": 1,
                        "ms_per_step": 1.0, "ms_per_step_uncorrected": 1.1, "event_pair_ms": 0.003, "frac_source": "back_to_back",
                        "in_step": {"frac": 0.4}},
           "kernels": [{"kernel": "fps_sort_kernel + ...", "ms_per_step": 2.0, "cu_ms_per_step": 64.0, "us_per_serial_step": 0.5},
                       {"kernel": "ball query: ...", "ms_per_step": 0.15, "frac": 0.1}]}
    leg = b.leg_record(res)
    assert leg["roofline"]["frac"] == 0.5 and leg["roofline"]["in_step_frac"] == 0.4 and leg["fps_cu_ms_per_step"] == 64.0
    assert leg["ball_query_hbm_frac"] == 0.1 and leg["mlp_ms_per_step"] == 1.0
