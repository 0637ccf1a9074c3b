"""bench_launch.py -- how `python bench.py --gpus N` comes to have N ranks, and how a stalled run ends (no GPU library is imported here).

  * heartbeat + watchdogs: every rank reports the stage it has reached (`beat`); a daemon thread inside each rank and the launching
    parent both read it and end a run whose rendezvous, communicator set-up, first collective or a timed window stalls;
  * the launching parent (`launch`): plain `python bench.py --gpus N` never touches a GPU; it starts torch.distributed.run ranks as a
    child process group, relays rank 0's single JSON line and falls back to `--single-process` (one process, a thread per device);
  * RCCL's own log: the excerpt printed after a run (rings / trees / transport) and, under SMK_BENCH_RCCL_TUNING=1, its per-size
    algorithm / protocol choices for the JSON line (`rccl_choices`);
  * `StdoutGuard`: stdout carries only the JSON line (RCCL prints banners there).
Tests: tests/test_bench_launcher.py (CPU), tests/test_gpu_dist.py."""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))

def lim(args, floor):
    """stage limit: long stages (matrix fill, solver set-up, the first collectives, the CPU baseline) get `floor` seconds
    unless --stall-s was lowered below a minute -- then the caller wants failures fast and gets exactly that"""
    return max(args.stall_s, floor) if args.stall_s >= 60.0 else args.stall_s


# ---- heartbeat + watchdog ------------------------------------------------------------------------------------------
# Every rank reports the stage it has reached (`beat`).  Two watchdogs read it:
#   * inside each rank, a daemon thread: no beat for longer than the stage's limit -> Python stacks of all threads and
#     the RCCL log excerpt on stderr, exit code 3 (torch.distributed.run then ends the other ranks);
#   * in the launching parent (plain `python bench.py --gpus N`): the ranks append their beats to one file; no new line
#     for the limit + a margin, or the overall --watchdog-s, -> the children's process group is killed.
# A C call that never returns (a collective whose peer is missing) does not hold the GIL, so the thread gets to run.
_hb = {"stage": "start", "t": time.monotonic(), "limit": 300.0, "rank": 0, "file": os.environ.get("SMK_BENCH_HEARTBEAT"), "armed": False}
_hb_lock = threading.Lock()


def beat(stage, limit=None, rank=None):
    """this rank has reached `stage`; it must reach the next one within `limit` seconds (default: --stall-s)"""
    with _hb_lock:
        _hb["stage"], _hb["t"] = stage, time.monotonic()
        if limit is not None:
            _hb["limit"] = float(limit)
        r = _hb["rank"] if rank is None else rank
        lim_now = _hb["limit"]
    if _hb["file"]:
        try:                                # rank <tab> limit of this stage <tab> stage: the parent applies the same limit (+ a margin)
            with open(_hb["file"], "a") as f:
                f.write(f"{r}\t{lim_now:.0f}\t{stage}\n")
        except OSError:
            pass
    if os.environ.get("SMK_BENCH_VERBOSE"):
        print(f"[bench rank {r}] {stage}", file=sys.stderr, flush=True)
    hang = os.environ.get("SMK_BENCH_TEST_HANG")          # TEST HOOK "rank:stage": that rank stops for good at that stage
    if hang:
        hr, _, hs = hang.partition(":")
        if int(hr) == r and stage.startswith(hs):
            print(f"[bench rank {r}] TEST HOOK: hanging at stage '{stage}'", file=sys.stderr, flush=True)
            time.sleep(1e6)


def rccl_log_excerpt(max_lines=60, out=sys.stderr):
    """RCCL's own description of what it built (rings / trees / transport per channel), from the per-rank files that
    NCCL_DEBUG_FILE names"""
    import glob
    import re
    pat = re.compile(r"Channel|Ring|Tree|Trees|XGMI|xgmi|P2P|SHM|NET|algo|Algo|proto|Connected|nChannels|comm 0x|WARN|error|fail", re.I)
    files = sorted(glob.glob(os.environ.get("SMK_BENCH_RCCL_LOG_GLOB", "/tmp/smk_rccl_*.log")), key=os.path.getmtime, reverse=True)
    for fn in files[:1]:
        try:
            lines = [l.rstrip() for l in open(fn, errors="replace") if pat.search(l)]
            print(f"[bench] RCCL log excerpt ({fn}, {len(lines)} matching lines, first {max_lines}):", file=out)
            for l in lines[:max_lines]:
                print("[rccl] " + l[:220], file=out)
        except Exception as e:      # pragma: no cover
            print(f"[bench] no RCCL log: {e}", file=out)
    if not files:
        print("[bench] no RCCL log files (/tmp/smk_rccl_*.log)", file=out)
    out.flush()


def rccl_choices(max_items=40):
    """What RCCL chose per collective and size -- its own TUNING lines ("AllReduce: 33554432 Bytes -> Algo 1 proto 2 time ...",
    NCCL_DEBUG_SUBSYS=TUNING) from this run's per-rank log files, de-duplicated; algorithm / protocol numbers are RCCL's enums
    (algo 0 tree, 1 ring, ...; proto 0 LL, 1 LL128, 2 simple).  Empty unless the run was started with SMK_BENCH_RCCL_TUNING=1 (RCCL
    then logs every collective call: a diagnostic run, not a measurement), with one rank, or with another log format."""
    import glob
    import re
    pat = re.compile(r"(\w+): (\d+) Bytes -> Algo (\d+) proto (\d+)")
    seen, out = set(), []
    try:
        files = sorted(glob.glob(os.environ.get("SMK_BENCH_RCCL_LOG_GLOB", "/tmp/smk_rccl_*.log")), key=os.path.getmtime, reverse=True)
        for fn in files[:1]:
            for l in open(fn, errors="replace"):
                m = pat.search(l)
                if m and m.groups() not in seen:
                    seen.add(m.groups())
                    out.append({"collective": m.group(1), "bytes": int(m.group(2)), "algo": int(m.group(3)), "proto": int(m.group(4))})
                    if len(out) >= max_items:
                        return out
    except Exception:
        pass
    return out


def start_rank_watchdog(stall_s):
    """daemon thread of a rank: exit(3) with diagnostics when the heartbeat stops"""
    if _hb["armed"] or os.environ.get("SMK_BENCH_NO_RANK_WATCHDOG"):      # TEST HOOK: leave a hung rank to the parent's watchdog
        return
    _hb["armed"] = True

    def watch():
        import faulthandler
        while True:
            time.sleep(1.0)
            with _hb_lock:
                idle, stage, limit, r = time.monotonic() - _hb["t"], _hb["stage"], _hb["limit"], _hb["rank"]
            if stage == "done":
                return
            if idle > limit:
                print(f"[bench rank {r}] WATCHDOG: no progress for {idle:.0f} s in stage '{stage}' (limit {limit:.0f} s); "
                      "Python stacks of all threads follow, then the RCCL log excerpt; exiting with code 3", file=sys.stderr, flush=True)
                try:
                    faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
                    rccl_log_excerpt()
                except Exception:
                    pass
                os._exit(3)
    _hb["limit"] = max(_hb["limit"], stall_s)
    threading.Thread(target=watch, name="bench-watchdog", daemon=True).start()


# ---- the launching parent: plain `python bench.py --gpus N` ------------------------------------------------------------
def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_plan(name, cmd, env, args, deadline):
    """one attempt: the command as a process group of its own, stdout captured, stderr passed through; returns
    (return code, JSON line or None, reason)"""
    import signal
    import subprocess
    import tempfile
    hb = tempfile.NamedTemporaryFile(prefix="smk_bench_hb_", suffix=".txt", delete=False)
    hb.close()
    env = dict(env, SMK_BENCH_HEARTBEAT=hb.name)
    print(f"[bench] launching plan '{name}': {' '.join(cmd)}", file=sys.stderr, flush=True)
    p = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=None, text=True, start_new_session=True)
    out_lines = []
    reader = threading.Thread(target=lambda: out_lines.extend(p.stdout.readlines()), daemon=True)
    reader.start()
    last_size, last_change, reason = 0, time.monotonic(), ""
    stages, limits = {}, {}                              # per rank: the stage it is in and the time that stage may take
    while p.poll() is None:
        time.sleep(0.25)
        try:
            size = os.path.getsize(hb.name)
        except OSError:
            size = last_size
        now = time.monotonic()
        if size != last_size:
            last_size, last_change = size, now
            try:
                for l in open(hb.name):
                    r, lm, st = l.rstrip("\n").split("\t", 2)
                    stages[r], limits[r] = st, float(lm)
            except Exception:
                pass
        # until the first beat: imports of a cold image take minutes.  Afterwards the longest limit among the stages the ranks
        # are in, plus a margin: the ranks' own watchdogs fire first -- they know more
        limit = (max(limits.values()) if limits else 300.0) + 15.0
        if now - last_change > limit:
            reason = f"no heartbeat from any rank for {now - last_change:.0f} s"
        elif now > deadline:
            reason = f"overall limit --watchdog-s {args.watchdog_s:.0f} s reached"
        if reason:
            print(f"[bench] WATCHDOG ({name}): {reason}; last stage per rank: {stages}", file=sys.stderr, flush=True)
            rccl_log_excerpt()
            for sig, wait in ((signal.SIGTERM, 5.0), (signal.SIGKILL, 5.0)):      # OUR process group only
                try:
                    os.killpg(p.pid, sig)
                except ProcessLookupError:
                    break
                t_end = time.monotonic() + wait
                while p.poll() is None and time.monotonic() < t_end:
                    time.sleep(0.1)
                if p.poll() is not None:
                    break
            break
    rc = p.wait()
    reader.join(timeout=5.0)
    try:
        os.unlink(hb.name)
    except OSError:
        pass
    line = None
    for l in out_lines:
        l = l.strip()
        if l.startswith("{"):
            try:
                if "metric" in json.loads(l):
                    line = l
            except ValueError:
                pass
    if reason and rc == 0:
        rc = 3
    return rc, line, reason


def launch(args):
    """Plain `python bench.py --gpus N` (N > 1, no WORLD_SIZE).  This process imports neither torch nor the library and
    makes no GPU call: the ranks are fresh children.  Plan A = torch.distributed.run (one process per GPU), plan B =
    one process with a thread per device (--single-process).  Rank 0's JSON line is relayed with a `launcher` object."""
    t_start = time.monotonic()
    deadline = t_start + args.watchdog_s
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # the host driver only supports dmabuf IPC (RCCL across processes)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    common = ["--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup), "--workload", args.workload,
              "--data", args.data, "--stall-s", str(args.stall_s)] + (["--single-copy"] if args.single_copy else [])
    if args.no_cpu_baseline:
        common.append("--no-cpu-baseline")
    me = os.path.join(ROOT, "bench.py")
    plans = []
    if not args.single_process:
        plans.append(("torch.distributed.run, one process per GPU",
                      [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                       "--master-addr", "127.0.0.1", "--master-port", str(free_port()), me] + common))
    if args.single_process or not args.no_fallback:
        plans.append(("single process, one host thread per device (ncclCommInitAll)",
                      [sys.executable, me, "--single-process", "--in-child"] + common))
    attempts = []
    for name, cmd in plans:
        if time.monotonic() > deadline - 30.0 and attempts:
            break
        t0 = time.monotonic()
        rc, line, reason = run_plan(name, cmd, env, args, deadline)
        attempts.append({"plan": name, "rc": rc, "seconds": round(time.monotonic() - t0, 1), "stopped_by_watchdog": reason or None})
        if rc == 0 and line:
            out = json.loads(line)
            out["launcher"] = {"mode": "self-launched children (the parent makes no GPU call)", "attempts": attempts}
            print(json.dumps(out), flush=True)
            return 0
        print(f"[bench] plan '{name}' ended with code {rc}" + (f" ({reason})" if reason else "") + (", no JSON line" if not line else ""),
              file=sys.stderr, flush=True)
    print(f"[bench] no plan produced a result: {json.dumps(attempts)}", file=sys.stderr, flush=True)
    worst = max((abs(a["rc"]) for a in attempts), default=1)
    return worst if 0 < worst < 256 else 1


def rccl_env_defaults(set_keys):
    """RCCL's own description of what it built goes to a file per rank (an excerpt is printed after the run, or by the
    watchdog); on one node its bootstrap sockets stay on the loopback interface (the container's hostname may not
    resolve) and no InfiniBand probing -- the data path is xGMI either way.  The caller's settings win."""
    if "NCCL_DEBUG" not in os.environ:
        os.environ["NCCL_DEBUG"] = "INFO"
        # TUNING makes RCCL print one line per collective CALL (algorithm / protocol for that size): wanted once, for rccl_choices,
        # but it is host work inside the timed region -- opt-in (SMK_BENCH_RCCL_TUNING=1), never in a run whose number is quoted
        os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH" + (",TUNING" if os.environ.get("SMK_BENCH_RCCL_TUNING") == "1" else ""))
        os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/smk_rccl_%h_%p.log")
    if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
        for key, val in (("NCCL_SOCKET_IFNAME", "lo"), ("NCCL_IB_DISABLE", "1")):
            if key not in os.environ:
                os.environ[key] = val
                set_keys.append(key)


class StdoutGuard:
    """RCCL prints a version banner on stdout when a communicator is created (and INFO lines at the first collectives if
    NCCL_DEBUG_FILE is not honoured): fd 1 points at stderr while it can, stdout carries ONLY the JSON line"""

    def __init__(self, active):
        self.saved = None
        if active:
            sys.stdout.flush()
            self.saved = os.dup(1)
            os.dup2(2, 1)

    def restore(self):
        if self.saved is None:
            return
        sys.stdout.flush()
        try:                                # what sits in the C library's stdio buffer goes out while fd 1 is still stderr
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(self.saved, 1)
        os.close(self.saved)
        self.saved = None


