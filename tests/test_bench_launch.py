"""`python bench.py --gpus N` must start its own ranks (the driver runs it that way; the reference goes multi-GPU in-process,
main.py:350-355).  CPU part: the parent makes no torch / HIP call before it spawns, gives every child its rank environment,
relays rank 0's JSON line as its last stdout line and fails when a child fails.  GPU part: two real ranks of the toy encoder on
one device over gloo — the whole N > 1 code path of bench.py incl. the `rccl` block."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_FAKE = r'''
import json, os, subprocess, sys
sys.argv = ["bench.py", "--gpus", "3", "--steps", "2", "--warmup", "1"]
sys.path.insert(0, %(root)r)
import bench
assert "torch" not in sys.modules, "bench.py imported torch before deciding whether it is the launcher"
seen = []
class FakeProc:
    def __init__(self, cmd, env=None, stdout=None):
        assert "torch" not in sys.modules and "torch.cuda" not in sys.modules, "the launcher touched torch before spawning"
        assert cmd[0] == sys.executable and cmd[1].endswith("bench.py") and cmd[2:] == sys.argv[1:], cmd
        self.rank = int(env["RANK"])
        seen.append({k: env[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")})
        self.stdout = None
        if self.rank == 0:
            assert stdout == subprocess.PIPE
            import io
            self.stdout = io.BytesIO(b"banner of some library\n" + json.dumps({"value": 1.5, "n_gpus": 3}).encode() + b"\n")
        else:
            assert stdout is None          # only rank 0 owns the relayed stdout
    def poll(self):
        return %(code)s if self.rank == %(bad)d else 0
    def wait(self):
        return self.poll()
    def kill(self):
        pass
subprocess.Popen = FakeProc
try:
    bench.main()
except SystemExit as e:
    print("RANKS " + json.dumps(seen), file=sys.stderr)
    assert "torch" not in sys.modules, "the launcher imported torch"
    sys.exit(e.code)
raise AssertionError("launcher returned")
'''


def _run_fake(bad, code):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    return subprocess.run([sys.executable, "-c", _FAKE % {"root": ROOT, "bad": bad, "code": code}], capture_output=True, text=True, env=env, timeout=120)


def test_launcher_spawns_ranks_without_touching_torch_and_relays_the_json_line():
    r = _run_fake(bad=-1, code=0)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert json.loads(lines[-1]) == {"value": 1.5, "n_gpus": 3} and lines[0] == "banner of some library"
    ranks = json.loads([l for l in r.stderr.splitlines() if l.startswith("RANKS ")][0][6:])
    assert [e["RANK"] for e in ranks] == ["0", "1", "2"] and [e["LOCAL_RANK"] for e in ranks] == ["0", "1", "2"]
    assert all(e["WORLD_SIZE"] == "3" and e["MASTER_ADDR"] == "127.0.0.1" for e in ranks)
    assert len({e["MASTER_PORT"] for e in ranks}) == 1 and int(ranks[0]["MASTER_PORT"]) > 0


def test_launcher_fails_when_a_rank_fails():
    r = _run_fake(bad=2, code=7)
    assert r.returncode == 1 and "(2, 7)" in r.stderr, (r.returncode, r.stderr)


_REAL = r'''
import subprocess, sys, time
sys.argv = ["bench.py", "--gpus", "2"]
sys.path.insert(0, %(root)r)
import bench
real = subprocess.Popen
def popen(cmd, env=None, stdout=None):
    # rank 0 blocks "in the rendezvous" with its stdout pipe open; rank 1 dies at once (bad GPU index, import error, ...)
    body = "import time; time.sleep(600)" if env["RANK"] == "0" else "import sys; sys.exit(3)"
    return real([sys.executable, "-c", body], env=env, stdout=stdout)
subprocess.Popen = popen
t0 = time.time()
try:
    bench.main()
except SystemExit as e:
    print("ELAPSED %%.1f" %% (time.time() - t0), file=sys.stderr)
    sys.exit(e.code)
'''


def test_launcher_does_not_hang_when_a_rank_other_than_0_dies_early():
    """Real child processes: rank 0 sits in its rendezvous holding the relayed pipe open, rank 1 exits non-zero straight away.  The launcher
    must notice (it drains rank 0's stdout on a thread, so its poll loop runs meanwhile), wait the grace period, kill rank 0 and fail."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["SCL_BENCH_GRACE_S"] = "1.5"
    r = subprocess.run([sys.executable, "-c", _REAL % {"root": ROOT}], capture_output=True, text=True, env=env, timeout=60)
    assert r.returncode == 1 and "(1, 3)" in r.stderr, (r.returncode, r.stderr)
    elapsed = float([l for l in r.stderr.splitlines() if l.startswith("ELAPSED ")][0].split()[1])
    assert elapsed < 20.0, elapsed


def test_under_a_launcher_bench_is_a_rank_not_a_launcher():
    """WORLD_SIZE in the environment (torch.distributed.run) -> no children; checked without a GPU by the failure mode: the
    process goes on to initialise the backend itself (here it has no GPU and must fail, not spawn)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'args.gpus > 1 and "WORLD_SIZE" not in os.environ' in src


@pytest.mark.gpu
def test_bench_gpus_2_self_launched_on_one_device(dev):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SCL_BENCH_BACKEND="gloo", SCL_BENCH_ONE_DEVICE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--tiny", "--steps", "3", "--warmup", "1",
                        "--batch", "4", "--samples", "16000"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["value"] > 0
    assert line["rccl"]["rccl_ranks"] == 2 and line["rccl"]["backend"] == "gloo" and line["rccl"]["grad_bytes_per_step"] > 0
    assert "cpu_baseline" not in line            # N = 1 only


@pytest.mark.gpu
def test_bench_eval_mode_times_both_scoring_precisions(dev):
    """`python bench.py --eval`: the scoring forward of 03_eval.sh (model.eval(), is_train False, no grad) on the fp32 scoring path and on
    the bf16 training kernels, same weights and input; the JSON line carries both rates and how far the two sets of log-probs are apart."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--eval", "--tiny", "--steps", "2", "--warmup", "1", "--batch", "4",
                        "--samples", "16000"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["metric"].startswith("scoring utterances/sec") and line["dtype"] == "f32" and line["config"]["samples"] == 16000
    assert line["fp32"]["utterances_per_s"] > 0 and line["bf16"]["utterances_per_s"] > 0 and line["value"] == line["fp32"]["utterances_per_s"]
    assert line["bf16_vs_fp32"]["max_rel_to_largest"] < 5e-2 and 0.0 <= line["bf16_vs_fp32"]["argmax_agree"] <= 1.0
