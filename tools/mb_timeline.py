"""Timeline of the default step form (two micro-batch graphs on two streams + join) from a rocprofv3 --kernel-trace CSV: for the last
replayed step, the wall span, the time any kernel is running, the time two or more run together, idle time, and per kernel family the
time it runs ALONE vs OVERLAPPED.  usage: python tools/mb_timeline.py <dir with *_kernel_trace.csv>"""
import csv, glob, re, sys, collections

paths = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for p in paths:
    rows += list(csv.DictReader(open(p)))
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("Stream_Id", "")) for r in rows]
rows.sort()
adam = [i for i, r in enumerate(rows) if "adam_multi" in r[2]]
print(f"{len(rows)} dispatches, {len(adam)} optimizer steps")
lo, hi = adam[-2], adam[-1]
step = rows[lo + 1:hi + 1]
t0, t1 = rows[lo][1], rows[hi][1]
span = (t1 - t0) / 1e6


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"([A-Za-z0-9_:]+)", n)
    return m.group(1) if m else n[:40]


ev = []
for s, e, n, q, st in step:
    ev.append((s, 1, short(n)))
    ev.append((e, -1, short(n)))
ev.sort()
active = collections.Counter()
nact = 0
busy1 = busy2 = idle = 0
alone = collections.Counter()
shared = collections.Counter()
prev = t0
for t, d, n in ev:
    dt = t - prev
    if dt > 0:
        if nact == 0:
            idle += dt
        elif nact == 1:
            busy1 += dt
            for k, c in active.items():
                if c > 0:
                    alone[k] += dt
        else:
            busy2 += dt
            for k, c in active.items():
                if c > 0:
                    shared[k] += dt
    prev = t
    active[n] += d
    nact += d
queues = collections.Counter(q for _, _, _, q, _ in step)
print(f"step: {len(step)} launches, span {span:.2f} ms; one kernel running {busy1 / 1e6:.2f} ms, two or more {busy2 / 1e6:.2f} ms, none {idle / 1e6:.2f} ms; "
      f"sum of durations {sum(e - s for s, e, *_ in step) / 1e6:.2f} ms; queues {dict(queues)}")
print(f"{'kernel':34s} {'alone ms':>9s} {'overlapped ms':>14s} {'launches':>9s}")
cnt = collections.Counter(short(n) for _, _, n, _, _ in step)
for k in sorted(set(alone) | set(shared), key=lambda k: -(alone[k] + shared[k]))[:28]:
    print(f"{k:34s} {alone[k] / 1e6:9.2f} {shared[k] / 1e6:14.2f} {cnt[k]:9d}")
