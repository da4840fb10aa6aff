"""Per-thread CPU time of this process between two snapshots (Linux /proc): which threads the host cores go to."""
import os


def snapshot():
    out = {}
    tick = os.sysconf("SC_CLK_TCK")
    for tid in os.listdir("/proc/self/task"):
        try:
            with open("/proc/self/task/%s/stat" % tid) as f:
                s = f.read()
            comm = s[s.index("(") + 1:s.rindex(")")]
            rest = s[s.rindex(")") + 2:].split()
            out[int(tid)] = (comm, (int(rest[11]) + int(rest[12])) / tick, int(rest[11]) / tick, int(rest[12]) / tick)
        except (OSError, ValueError):
            pass
    return out


def diff(a, b, top=12):
    rows = []
    for tid, rec in b.items():
        comm, t = rec[0], rec[1]
        old = a.get(tid, (comm, 0.0, 0.0, 0.0))
        if t - old[1] > 0:
            rows.append((t - old[1], comm, tid, rec[2] - old[2], rec[3] - old[3]))
    rows.sort(reverse=True)
    by = {}
    for dt, comm, *_ in rows:
        n, s = by.get(comm, (0, 0.0))
        by[comm] = (n + 1, s + dt)
    return rows[:top], sorted(by.items(), key=lambda kv: -kv[1][1])
