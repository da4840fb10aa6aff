"""bench.py's entry-point contract on CPU: `--gpus N` without a launcher starts N ranks itself as a child
torch.distributed.run BEFORE the parent imports torch or loads HIP, relays rank 0's JSON line and exits with the
child's status; a WORLD_SIZE / --gpus mismatch is refused instead of reporting a wrong n_gpus."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return e


def test_self_launch_two_ranks():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env=_env(), capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.strip().split("\n") if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2.0 and j["launched_by_bench"] is True


def test_parent_never_touches_torch_or_hip():
    # the launcher path of main() runs with torch / zigz_amd absent from sys.modules (it asserts so itself); check
    # from outside as well: import bench and walk the launcher branch with the child replaced by a stub
    code = (
        "import sys, os; sys.argv = ['bench.py', '--gpus', '3', '--dry-run'];"
        "sys.path.insert(0, %r); import bench, subprocess;"
        "class P:\n"
        "    stdout = iter(['{\"ok\": 1}\\n'])\n"
        "    def wait(self): return 0\n"
        "calls = []\n"
        "def popen(cmd, **kw):\n"
        "    calls.append(cmd); return P()\n"
        "subprocess.Popen = popen\n"
        "rc = bench.main()\n"
        "assert rc == 0 and 'torch' not in sys.modules and 'zigz_amd' not in sys.modules, list(sys.modules)[-5:]\n"
        "c = calls[0]; assert '--nproc-per-node=3' in c and 'torch.distributed.run' in c and '127.0.0.1' in c, c\n"
        "print('launcher ok')\n") % ROOT
    code = code.replace(";class P", "\nclass P").replace("import bench, subprocess;", "import bench, subprocess\n")
    out = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "launcher ok" in out.stdout, out.stderr[-2000:]


def test_world_size_mismatch_is_refused():
    e = _env()
    e.update(WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env=e, capture_output=True, text=True, timeout=120)
    assert out.returncode == 2 and "refusing" in out.stderr


def test_single_rank_dry_run():
    out = subprocess.run([sys.executable, BENCH, "--dry-run"], env=_env(), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and json.loads(out.stdout.strip())["n_gpus"] == 1


def test_default_lane_count_follows_the_cpu_share(monkeypatch):
    """What a rank sizes its threads for: its own slice of the cores when it has pinned itself (NOT divided by the world size
    again), its share of the common mask otherwise, never more than its share of a cgroup quota; without the sponge service one
    lane per CPU, two left for the helper threads, at most 14."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert 1 <= b.host_cpus() <= len(os.sched_getaffinity(0))
    for cpus, want in ((16, 14), (32, 14), (8, 6), (3, 1), (2, 1)):
        assert b.default_batch(cpus) == want, cpus
    for mask, quota, world, pinned, want in ((256, None, 8, 0, 32),    # eight unpinned ranks share 256 CPUs
                                             (32, None, 8, 32, 32),    # a pinned rank keeps its 32-core slice ...
                                             (32, 128, 8, 32, 16),     # ... unless the launch's cgroup grants less
                                             (128, 16, 1, 128, 16),    # this pool's 1-GPU boxes: one socket pinned, 16-CPU quota
                                             (4, None, 8, 0, 1)):
        monkeypatch.setattr(b.os, "sched_getaffinity", lambda _pid, m=mask: set(range(m)))
        monkeypatch.setattr(b, "cgroup_cpu_quota", lambda q=quota: q)
        assert b.rank_cpus(world, pinned) == want, (mask, quota, world, pinned)
