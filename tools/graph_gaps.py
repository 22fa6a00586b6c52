"""Where is a hipGraph-replayed training step NOT filling the chip?  From a rocprofv3 kernel trace (CSV) of
tools/train_replay.py: for the last replayed step, the wall time during which (a) nothing runs -- attributed to the kernel
that starts next (its launch / dependency latency) -- and (b) the kernels in flight together have fewer than `fill`
workgroups (default 128: half a wave of workgroups on 256 CUs) -- attributed to the kernels in flight, pro rata.

    rocprofv3 --kernel-trace --output-format csv -d DIR -o p -- python tools/train_replay.py c3 14
    python tools/graph_gaps.py DIR/p_kernel_trace.csv [fill=128] [launches per replay, for traces without an optimizer]
"""
import collections
import csv
import re
import sys


def family(name):
    n = name.replace("void ", "")
    n = re.sub(r"^pwg::", "", n)
    n = n.split("(")[0]
    return n.split("<")[0]


def load(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            wgs = 1
            for k, w in (("Grid_Size_X", "Workgroup_Size_X"), ("Grid_Size_Y", "Workgroup_Size_Y"), ("Grid_Size_Z", "Workgroup_Size_Z")):
                wgs *= max(1, int(r[k]) // max(int(r.get(w, 1)), 1))
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), family(r["Kernel_Name"]), wgs))
    rows.sort()
    return rows


def main():
    rows = load(sys.argv[1])
    fill = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    # steps are cut at every second optimizer launch (generator + discriminator)
    opt = [i for i, r in enumerate(rows) if r[2].startswith("adam_multi") or r[2].startswith("radam_multi")]
    if len(opt) >= 6:
        lo, hi = opt[-5] + 1, opt[-3] + 1  # the step before the last one (complete)
        rows = rows[lo:hi]
    elif len(sys.argv) > 3:  # no optimizer in the trace (inference replays): the last-but-one block of argv[3] launches
        per = int(sys.argv[3])
        rows = rows[-2 * per:-per]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    ev = []
    for i, (s, e, _, _) in enumerate(rows):
        ev.append((s, 1, i))
        ev.append((e, 0, i))
    ev.sort()
    active = set()
    idle_to = collections.Counter()
    starved = collections.Counter()
    launches = collections.Counter(r[2] for r in rows)
    idle = starved_t = 0.0
    last = t0
    pending_idle = 0.0
    for t, kind, i in ev:
        dt = (t - last) / 1e3
        if dt > 0:
            if not active:
                idle += dt
                pending_idle += dt
            else:
                w = sum(rows[j][3] for j in active)
                if w < fill:
                    starved_t += dt
                    for j in active:
                        starved[rows[j][2]] += dt / len(active)
        last = t
        if kind == 1:
            if pending_idle:
                idle_to[rows[i][2]] += pending_idle
                pending_idle = 0.0
            active.add(i)
        else:
            active.discard(i)
    span = (t1 - t0) / 1e3
    print(f"{sys.argv[1]}: one replayed step, {len(rows)} launches, span {span / 1e3:.3f} ms; nothing in flight {idle / 1e3:.3f} ms; "
          f"fewer than {fill} workgroups in flight {starved_t / 1e3:.3f} ms")
    print(f"{'family':44s} {'launches':>8s} {'idle before it (ms)':>20s} {'starved while in flight (ms)':>30s}")
    fams = sorted(set(idle_to) | set(starved), key=lambda k: -(idle_to[k] + starved[k]))
    for k in fams[:30]:
        print(f"{k:44s} {launches[k]:8d} {idle_to[k] / 1e3:20.3f} {starved[k] / 1e3:30.3f}")


if __name__ == "__main__":
    main()
