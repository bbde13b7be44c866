"""GPU timeline of the bench's steady state from a rocprofv3 --kernel-trace CSV.

Window: the last `steps` steps of the trace, a step = one launch of the selection kernel (select4_kernel; traces of
rounds 1-4: the plan kernel) -- run the bench with --no-kernel-timing --no-bf16 --no-cpu-baseline --weights random so that the
timed window is the last thing that launches it.  Reports the span per step, the union of kernel-busy time, the sum of
kernel durations (overlap) and the largest idle gaps."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = list(csv.DictReader(open(path)))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
marks = [s for s, e, n in ks if "select4_kernel" in n] or [s for s, e, n in ks if "_plan_kernel" in n]
steps = min(steps, len(marks) - 1)
w0, w1 = marks[-steps - 1], marks[-1]
win = [(s, e, n) for s, e, n in ks if w0 <= s < w1]
busy, cur_s, cur_e, gaps = 0, win[0][0], win[0][1], []
for s, e, n in win[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += min(cur_e, w1) - cur_s
tot = sum(e - s for s, e, _ in win)
print(f"{steps} steps: span {(w1 - w0) / steps / 1e3:.1f} us/step, busy-union {busy / steps / 1e3:.1f} us/step "
      f"({busy / (w1 - w0):.1%}), sum of kernel durations {tot / steps / 1e3:.1f} us/step")
gaps.sort(reverse=True)
print(f"idle total {sum(g for g, _ in gaps) / steps / 1e3:.1f} us/step; largest gaps (us, next kernel):")
for g, name in gaps[:6]:
    print(f"  {g / 1e3:7.1f}  {name[:80]}")
agg = defaultdict(lambda: [0, 0])
for s, e, n in win:
    agg[n[:64]][0] += 1
    agg[n[:64]][1] += e - s
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:18]:
    print(f"  {d / steps / 1e3:7.1f} us/step  ({d / c / 1e3:6.1f} us x {c / steps:.1f})  {n}")
