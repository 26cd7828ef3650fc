"""Timeline of ONE training step from a rocprofv3 kernel trace (csv): every kernel in start order with its duration and the idle gap in
front of it -- what the glue between the three big kernel classes costs (small kernels + launch gaps).
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python tools/train_bench.py --rays 4096 --steps 6 --warmup 3 --articulated
    python tools/step_timeline.py gpurun_out/tl [anchor-substring]
The step is cut at the LAST two occurrences of the anchor kernel (default: the first training-forward launch of a step)."""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 else "fwd_kernel"
    files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # a step starts at the first anchor launch after a non-anchor stretch that contains an optimiser kernel
    idx = [i for i, r in enumerate(rows) if anchor in r[2]]
    starts = [i for k, i in enumerate(idx) if k == 0 or any(("multi_tensor" in rows[j][2] or "adam_arena" in rows[j][2]) for j in range(idx[k - 1], i))]
    a, b = starts[-2], starts[-1]
    step = rows[a:b]
    t0 = step[0][0]
    busy = sum(e - s for s, e, _ in step)
    span = rows[b][0] - t0
    print(f"# step of {len(step)} kernels: span {span / 1e6:.3f} ms, kernel time {busy / 1e6:.3f} ms, idle {(span - busy) / 1e6:.3f} ms")
    prev_end = t0
    small_n = small_t = small_gap = 0
    for s, e, n in step + [rows[b]]:
        gap = s - prev_end
        dur = e - s
        if dur < 100_000:
            small_n += 1; small_t += dur; small_gap += max(gap, 0)
        print(f"{(s - t0) / 1e3:10.1f} us  +{gap / 1e3:7.1f} gap  {dur / 1e3:9.1f} us  {n[:120]}")
        prev_end = max(prev_end, e)
    print(f"# kernels under 100 us: {small_n}, {small_t / 1e3:.1f} us of kernel time, {small_gap / 1e3:.1f} us of gaps in front of them")


if __name__ == "__main__":
    main()
