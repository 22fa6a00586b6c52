"""Timeline analysis of a hipGraph-replayed training step from a rocprofv3 kernel trace (CSV).

    rocprofv3 --kernel-trace --output-format csv -d DIR -o p -- python tools/train_replay.py c3 12
    python tools/graph_timeline.py DIR/p_kernel_trace.csv [steps_from_the_end=6] > profiles/r04_graph_timeline_c3.txt

The per-shape profiles (tools/profile_train_shapes.py) time EAGER, serial launches; the benchmark replays a graph whose
8 sub-discriminators and 3 MRF branches overlap.  This tool answers what the eager profile cannot: per replayed step,
the wall span, the union of kernel-busy time, the idle gaps, the time-weighted kernel concurrency, and -- per kernel
family -- the EXCLUSIVE time (wall time during which it is the only kernel on the device: the serial bottlenecks) next
to its summed duration.  Steps are cut at every second ``adam_multi`` launch (generator + discriminator optimizer)."""
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
            grid = "x".join(str(int(r[k]) // max(int(r.get(w, 1)), 1)) for k, w in (("Grid_Size_X", "Workgroup_Size_X"), ("Grid_Size_Y", "Workgroup_Size_Y"), ("Grid_Size_Z", "Workgroup_Size_Z")) if k in r)
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], grid, r.get("LDS_Block_Size", "")))
    rows.sort()
    return rows


def analyse(rows, out=sys.stdout, detail=False):
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    span = (t1 - t0) / 1e3
    total = sum(r[1] - r[0] for r in rows) / 1e3
    # sweep: union busy, concurrency histogram, exclusive time per family
    ev = []
    for i, (s, e, *_) in enumerate(rows):
        ev.append((s, 1, i))
        ev.append((e, -1, i))
    ev.sort(key=lambda x: (x[0], x[1]))
    active, last = set(), t0
    busy, conc, excl, gaps = 0.0, collections.Counter(), collections.Counter(), []
    excl_inst = collections.Counter()
    nb = 40
    bins = [[0.0, 0.0, collections.Counter()] for _ in range(nb)]
    for t, kind, i in ev:
        dt = (t - last) / 1e3
        if dt > 0:
            n = len(active)
            conc[min(n, 9)] += dt
            if n:
                busy += dt
            else:
                gaps.append(dt)
            if n == 1:
                j = next(iter(active))
                excl[family(rows[j][2])] += dt
                excl_inst[j] += dt
        last = t
        if kind == 1:
            active.add(i)
        else:
            active.discard(i)
    fam_sum, fam_n = collections.Counter(), collections.Counter()
    for s, e, n, *_ in rows:
        fam_sum[family(n)] += (e - s) / 1e3
        fam_n[family(n)] += 1
    print(f"  span {span / 1e3:8.3f} ms | kernel-busy union {busy / 1e3:8.3f} ms | idle {sum(gaps) / 1e3:7.3f} ms in {len(gaps)} gaps "
          f"({sum(1 for g in gaps if g > 5)} > 5 us, {sum(1 for g in gaps if g > 20)} > 20 us) | sum of durations {total / 1e3:8.3f} ms "
          f"| {len(rows)} launches | mean concurrency while busy {total / max(busy, 1e-9):.2f}", file=out)
    print("  time by number of kernels in flight: " + "  ".join(f"{k}{'+' if k == 9 else ''}: {v / 1e3:.2f} ms" for k, v in sorted(conc.items())), file=out)
    print("  family                                   launches   sum ms   exclusive ms (only kernel in flight)", file=out)
    for k, v in sorted(fam_sum.items(), key=lambda kv: -kv[1])[:24]:
        print(f"  {k:40s} {fam_n[k]:8d} {v / 1e3:8.3f} {excl[k] / 1e3:8.3f}", file=out)
    if detail:
        # strip chart: the step in 40 slices -- busy fraction, mean concurrency, dominant family
        w = (t1 - t0) / nb
        for s, e, n, *_ in rows:
            b0, b1 = int((s - t0) / w), min(int((e - t0) / w), nb - 1)
            for b in range(b0, b1 + 1):
                lo, hi = max(s, t0 + b * w), min(e, t0 + (b + 1) * w)
                if hi > lo:
                    bins[b][1] += hi - lo
                    bins[b][2][family(n)] += hi - lo
        print("  strip chart (40 slices of the step): slice start ms | mean kernels in flight | dominant families", file=out)
        for b in range(nb):
            top = ", ".join(f"{k} {v / w:.2f}" for k, v in bins[b][2].most_common(3))
            print(f"   {b * w / 1e6:7.2f} | {bins[b][1] / w:5.2f} | {top}", file=out)
        print("  longest single launches while NOTHING else is in flight (us exclusive / us duration, grid in workgroups, LDS):", file=out)
        for j, v in excl_inst.most_common(40):
            s, e, n, grid, lds = rows[j]
            print(f"   {v:8.1f} / {(e - s) / 1e3:8.1f}  grid {grid:>12s} lds {lds:>6s}  {n[:150]}", file=out)
    return span, busy, sum(gaps), total


def main():
    path = sys.argv[1]
    last_n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    rows = load(path)
    # cut into steps at every second optimizer launch
    opt = [i for i, r in enumerate(rows) if "adam_multi" in r[2] or "radam_multi" in r[2]]
    bounds = [opt[i] for i in range(1, len(opt), 2)]
    steps, prev = [], 0
    for b in bounds:
        steps.append(rows[prev:b + 1])
        prev = b + 1
    print(f"{path}: {len(rows)} dispatches, {len(steps)} steps (cut at every second optimizer launch); the last {last_n} (graph replays):")
    agg = []
    for i, st in enumerate(steps[-last_n:]):
        print(f"step -{last_n - i}:")
        agg.append(analyse(st, detail=(i == last_n - 1)))
    n = len(agg)
    print("mean of these steps: span %.3f ms, busy %.3f ms, idle %.3f ms, sum of durations %.3f ms" % tuple(sum(a[j] for a in agg) / n / 1e3 for j in range(4)))
    # inter-step gap (host side of a replay: optimizer prepare + graph launch)
    if len(steps) >= 2:
        inter = [(steps[i + 1][0][0] - max(r[1] for r in steps[i])) / 1e3 for i in range(len(steps) - last_n, len(steps) - 1)]
        print("gap between the last kernel of a step and the first of the next (us): " + " ".join(f"{g:.0f}" for g in inter))


if __name__ == "__main__":
    main()
