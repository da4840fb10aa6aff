"""Rank placement on a multi-GPU node: pin a rank's host threads to the cores of its GPU's NUMA node.

A rank of the proving service is one process per GPU with tens of host threads (one per proof in flight, the sponge
servers, the serialiser helpers) that feed ONE GPU through pinned buffers.  On a two-socket node threads that wander to the
other socket cross the inter-socket link for every doorbell, every pinned-buffer write and every result read; pinning costs
nothing and removes that variable before a scaling curve is taken.

Everything here reads sysfs only -- no HIP call, so it can (and must) run before the process touches the GPU: threads
started afterwards inherit the mask.

    HIP device i  ->  i-th GPU node of /sys/class/kfd/kfd/topology/nodes/*  (those with simd_count > 0, in node order,
                      filtered by ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES when they hold plain indices)
                  ->  its drm_render_minor  ->  /sys/class/drm/renderD<minor>/device/{numa_node, local_cpulist}

Ranks whose GPUs hang off the same NUMA node split that node's cores among themselves in contiguous slices.
"""
import os


def parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            cpus.update(range(int(a), int(b) + 1))
        else:
            cpus.add(int(part))
    return cpus


def _read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def _visible(indices_env, n):
    """Plain-index device filters (\"0,2,3\") -> list of physical indices; anything else (UUIDs, empty) -> no filtering."""
    if not indices_env:
        return list(range(n))
    try:
        ids = [int(x) for x in indices_env.split(",") if x.strip() != ""]
    except ValueError:
        return list(range(n))
    return [i for i in ids if 0 <= i < n]


def gpu_nodes(sysfs_root="/sys", env=None):
    """The GPUs in HIP enumeration order: [{"render_minor", "numa_node", "cpus"}]."""
    env = os.environ if env is None else env
    top = os.path.join(sysfs_root, "class/kfd/kfd/topology/nodes")
    try:
        names = sorted((n for n in os.listdir(top) if n.isdigit()), key=int)
    except OSError:
        return []
    gpus = []
    for n in names:
        props = _read(os.path.join(top, n, "properties"))
        if not props:
            continue
        kv = {}
        for ln in props.splitlines():
            p = ln.split()
            if len(p) == 2:
                kv[p[0]] = p[1]
        if int(kv.get("simd_count", "0")) <= 0:
            continue  # a CPU node
        minor = int(kv.get("drm_render_minor", "-1"))
        dev = os.path.join(sysfs_root, "class/drm/renderD%d/device" % minor)
        numa = _read(os.path.join(dev, "numa_node"))
        cpul = _read(os.path.join(dev, "local_cpulist"))
        node = int(numa.strip()) if numa and numa.strip().lstrip("-").isdigit() else -1
        cpus = parse_cpulist(cpul) if cpul else set()
        if not cpus and node >= 0:
            nl = _read(os.path.join(sysfs_root, "devices/system/node/node%d/cpulist" % node))
            cpus = parse_cpulist(nl) if nl else set()
        gpus.append({"render_minor": minor, "numa_node": node, "cpus": cpus})
    vis = _visible(env.get("ROCR_VISIBLE_DEVICES"), len(gpus))
    gpus = [gpus[i] for i in vis]
    vis = _visible(env.get("HIP_VISIBLE_DEVICES"), len(gpus))
    return [gpus[i] for i in vis]


def cpus_for_rank(local_rank, sysfs_root="/sys", env=None, allowed=None, local_world=None):
    """The cores rank `local_rank` (= HIP device index) should run on: its GPU's NUMA-local cores that this process may
    use, split among the GPUs of the same NUMA node.  None when the topology says nothing useful.
    local_world: ranks of this node when there are MORE of them than GPUs (a rehearsal of N ranks on fewer devices): rank r
    then uses device r % n_gpus and the ranks of one device split that device's share of the cores."""
    gpus = gpu_nodes(sysfs_root, env)
    sub = (0, 1)
    if local_world and gpus and local_world > len(gpus) and 0 <= local_rank < local_world:
        ng = len(gpus)
        on_dev = [r for r in range(local_world) if r % ng == local_rank % ng]
        sub = (on_dev.index(local_rank), len(on_dev))
        local_rank = local_rank % ng
    if local_rank < 0 or local_rank >= len(gpus):
        return None
    me = gpus[local_rank]
    if not me["cpus"]:
        return None
    allowed = set(os.sched_getaffinity(0)) if allowed is None else set(allowed)
    local = sorted(me["cpus"] & allowed)
    if not local:
        return None
    same = [i for i, g in enumerate(gpus) if g["cpus"] == me["cpus"]]  # the GPUs that share these cores
    k, pos = len(same), same.index(local_rank)
    per = len(local) // k
    if per == 0:
        return set(local)
    lo = pos * per
    hi = len(local) if pos == k - 1 else lo + per
    mine = local[lo:hi]
    if sub[1] > 1:  # several ranks on this device: its cores once more
        per2 = len(mine) // sub[1]
        if per2 > 0:
            mine = mine[sub[0] * per2:(len(mine) if sub[0] == sub[1] - 1 else (sub[0] + 1) * per2)]
    return set(mine)


def pin_rank(local_rank, sysfs_root="/sys", env=None, apply=None, local_world=None):
    """Pin the calling process (threads started later inherit it).  Returns a small record for the bench line."""
    try:
        cpus = cpus_for_rank(local_rank, sysfs_root, env, local_world=local_world)
    except Exception as e:  # placement is an optimisation: never fail a run over it
        return {"pinned": 0, "numa_node": -1, "why": repr(e)[:80]}
    if not cpus:
        return {"pinned": 0, "numa_node": -1, "why": "no NUMA information for this GPU"}
    gpus = gpu_nodes(sysfs_root, env)
    (apply or (lambda c: os.sched_setaffinity(0, c)))(cpus)
    return {"pinned": len(cpus), "numa_node": gpus[local_rank % len(gpus)]["numa_node"], "first_cpu": min(cpus), "last_cpu": max(cpus)}
