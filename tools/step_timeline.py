"""Timeline of the replayed contrastive step from a rocprofv3 --kernel-trace CSV: how much of a step's wall time has a kernel
running at all, how much has only side-stream work (weight gradients / heads) running, and which kernels precede the idle gaps.
usage: python tools/step_timeline.py <kernel_trace.csv> [n_steps_timed]"""
import collections, csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")) for r in rows), key=lambda t: t[0])


def short(n):
    m = re.search(r"(?:amx::)?(\w+_kernel|\w+)(?=[<(I]|$)", re.sub(r"^void ", "", n).replace("_ZN3amx", ""))
    n2 = re.sub(r"^\d+", "", m.group(1)) if m else n[:40]
    return n2[:40]


# steps: the AdamW kernel of the network closes a step; use the LAST kernel name of the trace as the marker
mk = [i for i, k in enumerate(ks) if "adamw_kernel" in k[2]]          # FusedAdamW: the last launch of a step
if not mk:
    mk = [i for i, k in enumerate(ks) if re.search(r"multi_tensor_apply|adam", k[2])]
if mk:                                                   # a step ends with the last optimizer kernel of a cluster
    ends = [i for i, j in zip(mk, mk[1:] + [10 ** 9]) if j > i + 50]
    marker = ks[mk[-1]][2]
else:
    marker = ks[-1][2]
    ends = [i for i, k in enumerate(ks) if k[2] == marker]
print(f"{len(ks)} kernels, marker {short(marker)!r}, {len(ends)} steps, kernels per step {[b - a for a, b in zip(ends[:-1], ends[1:])][-4:]}")
step = ks[ends[-2] + 1: ends[-1] + 1]
t0, t1 = step[0][0], max(k[1] for k in step)
wall = (t1 - t0) / 1e3
print(f"last step: {len(step)} kernels, wall {wall:.1f} us, sum of kernel time {sum(k[1] - k[0] for k in step) / 1e3:.1f} us")
queues = collections.defaultdict(list)
for k in step:
    queues[k[3]].append(k)
for q, v in sorted(queues.items(), key=lambda kv: -sum(k[1] - k[0] for k in kv[1])):
    print(f"  queue {q}: {len(v):4d} kernels, busy {sum(k[1] - k[0] for k in v) / 1e3:8.1f} us, e.g. {short(v[0][2])}, {short(v[len(v) // 2][2])}")
mainq = max(queues, key=lambda q: len(queues[q]))
# union busy and gaps
ev = sorted(step, key=lambda k: k[0])
cur_end, busy, gaps = ev[0][0], 0, collections.defaultdict(lambda: [0, 0.0])
last = None
for k in ev:
    if k[0] > cur_end:
        if last is not None:
            g = gaps[(short(last[2]), short(k[2]))]
            g[0] += 1; g[1] += (k[0] - cur_end) / 1e3
        busy += 0
    if k[1] > cur_end:
        busy += k[1] - max(cur_end, k[0])
        cur_end, last = k[1], k
print(f"some kernel running: {busy / 1e3:.1f} us ({busy / 10 / wall:.1f} %), idle {wall - busy / 1e3:.1f} us")
print("idle gaps by (previous kernel -> next kernel), top 25:")
for (a, b), (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {t:7.1f} us  x{n:3d}  {a} -> {b}")
# time during which the main queue is idle but another queue runs
segs = sorted((k[0], k[1]) for k in queues[mainq])
main_busy = sum(b - a for a, b in segs)
others = sorted((k[0], k[1], k[2]) for q, v in queues.items() if q != mainq for k in v)
only_side = collections.Counter()
mi = 0
for a, b, n in others:
    # subtract overlap with main-queue kernels
    t = b - a
    for s, e in segs:
        if e <= a: continue
        if s >= b: break
        t -= min(b, e) - max(a, s)
    only_side[short(n)] += max(t, 0)
print(f"main queue {mainq}: busy {main_busy / 1e3:.1f} us; other queues running while it is idle (upper bound, overlaps between side queues not merged):")
for n, t in only_side.most_common(12):
    print(f"  {t / 1e3:8.1f} us  {n}")
# main-queue kernel time by name
agg = collections.Counter()
for k in queues[mainq]:
    agg[short(k[2])] += k[1] - k[0]
print("main queue by kernel:")
for n, t in agg.most_common(25):
    print(f"  {t / 1e3:8.1f} us  {n}")
# full ordered listing of the last step (offset us, duration us, queue, kernel) beside the summary
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as f:
        for k in ev:
            f.write(f"{(k[0] - t0) / 1e3:9.1f} {(k[1] - k[0]) / 1e3:8.1f} q{k[3]:>3s} {short(k[2])}\n")
