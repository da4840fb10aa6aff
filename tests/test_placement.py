"""Rank placement (zigz_amd/placement.py): which cores a rank pins itself to, on a fake sysfs tree of a two-socket 8-GPU node."""
import importlib.util
import os

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("placement", os.path.join(HERE, "..", "zigz_amd", "placement.py"))
placement = importlib.util.module_from_spec(spec)
spec.loader.exec_module(placement)


def _fake_node(tmp_path, gpus_per_socket=4, cores_per_socket=16, with_numa=True):
    root = tmp_path / "sys"
    top = root / "class/kfd/kfd/topology/nodes"
    n = 0
    for s in range(2):  # the CPU nodes come first, like on a real host
        d = top / str(n)
        d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count 0\ndrm_render_minor -1\n" % cores_per_socket)
        n += 1
    minor = 128
    for s in range(2):
        for g in range(gpus_per_socket):
            d = top / str(n)
            d.mkdir(parents=True)
            (d / "properties").write_text("cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor %d\n" % minor)
            dev = root / ("class/drm/renderD%d/device" % minor)
            dev.mkdir(parents=True)
            (dev / "numa_node").write_text("%d\n" % (s if with_numa else -1))
            if with_numa:  # socket s: cores [16 s, 16 s + 16) and their SMT siblings [32 + 16 s, ...)
                (dev / "local_cpulist").write_text("%d-%d,%d-%d\n" % (cores_per_socket * s, cores_per_socket * (s + 1) - 1,
                                                                      32 + cores_per_socket * s, 32 + cores_per_socket * (s + 1) - 1))
            n += 1
            minor += 1
    return str(root)


def test_cpulist_parser():
    assert placement.parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert placement.parse_cpulist("") == set()


def test_ranks_split_the_cores_of_their_numa_node(tmp_path):
    root = _fake_node(tmp_path)
    allowed = set(range(64))
    slices = [placement.cpus_for_rank(r, root, env={}, allowed=allowed) for r in range(8)]
    for r in range(8):
        socket = r // 4
        local = set(range(16 * socket, 16 * socket + 16)) | set(range(32 + 16 * socket, 48 + 16 * socket))
        assert slices[r] and slices[r] <= local, (r, slices[r])
        assert len(slices[r]) == 8
    for a in range(8):
        for b in range(a + 1, 8):
            assert not (slices[a] & slices[b]), (a, b)
    assert set().union(*slices) == allowed


def test_visible_devices_and_cgroup_masks_are_respected(tmp_path):
    root = _fake_node(tmp_path)
    # only GPUs 5 and 6 are visible: HIP device 0 is physical GPU 5 (socket 1)
    s = placement.cpus_for_rank(0, root, env={"ROCR_VISIBLE_DEVICES": "5,6"}, allowed=set(range(64)))
    assert s and s <= (set(range(16, 32)) | set(range(48, 64)))
    # a cpuset that leaves this rank 4 of its socket's cores
    s = placement.cpus_for_rank(1, root, env={}, allowed={0, 1, 2, 3, 20, 21})
    assert s == {1}  # four GPUs share {0,1,2,3}: one core each
    # a cpuset with nothing on the GPU's socket: do not pin at all
    assert placement.cpus_for_rank(7, root, env={}, allowed={0, 1}) is None
    assert placement.cpus_for_rank(9, root, env={}, allowed=set(range(64))) is None  # no such device


def test_no_numa_information_means_no_pinning(tmp_path):
    root = _fake_node(tmp_path, with_numa=False)
    called = []
    rec = placement.pin_rank(0, root, env={}, apply=called.append)
    assert rec["pinned"] == 0 and not called


def test_pin_rank_applies_the_mask(tmp_path):
    root = _fake_node(tmp_path)
    called = []
    rec = placement.pin_rank(6, root, env={}, apply=called.append)
    allowed = set(os.sched_getaffinity(0))
    want = placement.cpus_for_rank(6, root, env={}, allowed=allowed)
    if want:
        assert called == [want] and rec["pinned"] == len(want) and rec["numa_node"] == 1
    else:  # this container's own mask has no core on the fake socket 1
        assert rec["pinned"] == 0 and not called


def test_more_ranks_than_devices_split_a_devices_cores(tmp_path):
    """A rehearsal of 8 ranks on a box with ONE GPU (VERDICT r3 #3b): rank r uses device r % 1 and the eight ranks split that
    device's NUMA-local cores -- disjoint slices that cover them; with two devices, ranks 0, 2, 4, 6 split device 0's."""
    root = _fake_node(tmp_path, gpus_per_socket=1)  # two GPUs, one per socket
    allowed = set(range(64))
    env1 = {"HIP_VISIBLE_DEVICES": "0"}
    got = [placement.cpus_for_rank(r, root, env1, allowed, local_world=8) for r in range(8)]
    assert all(g for g in got) and all(len(g) == 4 for g in got)
    assert set().union(*got) == set(range(16)) | set(range(32, 48))
    assert sum(len(g) for g in got) == 32  # disjoint
    got2 = [placement.cpus_for_rank(r, root, {}, allowed, local_world=8) for r in range(8)]
    assert set().union(*[got2[r] for r in (0, 2, 4, 6)]) == set(range(16)) | set(range(32, 48))
    assert set().union(*[got2[r] for r in (1, 3, 5, 7)]) == set(range(16, 32)) | set(range(48, 64))
    applied = []
    rec = placement.pin_rank(4, root, {}, apply=applied.append, local_world=8)  # (this process's own affinity mask applies here)
    if rec["pinned"]:
        assert rec["numa_node"] == 0 and applied and applied[0] <= set(range(16)) | set(range(32, 48))
