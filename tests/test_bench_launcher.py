"""bench.py started the plain way -- `python bench.py --gpus N`, no torch.distributed.run around it -- must start its
own ranks, never hang, and end non-zero when a rank stalls.  These cases run WITHOUT a GPU:
the ranks are made to stop in the CPU-only rendezvous stage (TEST HOOK SMK_BENCH_TEST_HANG), which is where a broken
node would stop them too; the GPU legs of the same launcher are in tests/test_gpu_dist.py."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--steps", "2", "--warmup", "1", "--workload", "c1", "--no-cpu-baseline", "--no-fallback"]


def run(extra_env, *args, timeout=240):
    env = dict(os.environ, **extra_env)
    env.pop("WORLD_SIZE", None)
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", *args, *COMMON], cwd=ROOT, capture_output=True, text=True,
                       timeout=timeout, env=env)
    return r, time.monotonic() - t0


def test_hung_rank_ends_the_run_nonzero_through_the_rank_watchdog():
    r, dt = run({"SMK_BENCH_TEST_HANG": "0:rendezvous"}, "--stall-s", "8", "--watchdog-s", "200")
    assert r.returncode != 0 and dt < 120, (r.returncode, dt)
    assert "WATCHDOG: no progress" in r.stderr and "rendezvous" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]         # no JSON line from a failed run


def test_hung_rank_is_killed_by_the_parent_when_the_ranks_cannot_help_themselves():
    r, dt = run({"SMK_BENCH_TEST_HANG": "0:rendezvous", "SMK_BENCH_NO_RANK_WATCHDOG": "1"}, "--stall-s", "5", "--watchdog-s", "200")
    assert r.returncode != 0 and dt < 120, (r.returncode, dt)
    assert "[bench] WATCHDOG" in r.stderr and "no heartbeat from any rank" in r.stderr
    # the children are gone: the parent killed its own process group, nothing else
    out = subprocess.run(["ps", "-eo", "pid,args"], capture_output=True, text=True).stdout
    assert "bench.py --gpus 2 --steps 2 --warmup 1 --workload c1" not in out, out


def test_overall_limit_applies():
    r, dt = run({"SMK_BENCH_TEST_HANG": "0:rendezvous", "SMK_BENCH_NO_RANK_WATCHDOG": "1"}, "--stall-s", "300", "--watchdog-s", "10")
    assert r.returncode != 0 and dt < 90, (r.returncode, dt)
    assert "overall limit --watchdog-s" in r.stderr


def test_the_launching_parent_imports_neither_torch_nor_the_library():
    """the parent must not touch the GPU: it gets by with the standard library (a process that has initialised HIP
    must not start other GPU programs on this pool)"""
    code = ("import sys; sys.argv = ['bench.py', '--gpus', '2', '--no-fallback', '--stall-s', '5', '--watchdog-s', '60', "
            "'--workload', 'c1', '--no-cpu-baseline'];\n"
            "import runpy, os\n"
            "os.environ.pop('WORLD_SIZE', None)\n"
            "try:\n    runpy.run_path('bench.py', run_name='__main__')\nexcept SystemExit as e:\n    print('exit', e.code)\n"
            "print('torch' in sys.modules, 'smallk_amd' in sys.modules, 'numpy' in sys.modules)\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=240,
                       env=dict(os.environ, SMK_BENCH_TEST_HANG="0:rendezvous"))
    assert r.stdout.strip().splitlines()[-1] == "False False False", (r.stdout, r.stderr[-2000:])
