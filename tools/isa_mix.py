"""Instruction mix of the kernels in a hipcc -S listing (no GPU needed):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only csrc/x.hip -o /tmp/x.s; python tools/isa_mix.py /tmp/x.s [filter]
Per kernel: total instructions and the most frequent opcodes; MFMA count and the ratio of everything else to it."""
import collections
import re
import subprocess
import sys


def main():
    txt = open(sys.argv[1]).read().splitlines()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    cur, counts = None, {}
    for l in txt:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = m.group(1)
            counts[cur] = collections.Counter()
            continue
        if cur is None:
            continue
        s = l.strip()
        if s.startswith(".Lfunc_end"):
            cur = None
            continue
        if not s or s[0] in ".;/" or s.endswith(":"):
            continue
        counts[cur][s.split()[0]] += 1
    for name, c in counts.items():
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if flt not in dem or not c:
            continue
        tot, mf = sum(c.values()), sum(v for k, v in c.items() if k.startswith("v_mfma"))
        print(f"{dem[:110]}: {tot} instructions, {mf} MFMA, {(tot - mf) / max(mf, 1):.3f} others per MFMA")
        groups = collections.Counter()
        for k, v in c.items():
            g = ("mfma" if k.startswith("v_mfma") else "accvgpr" if "accvgpr" in k else "ds" if k.startswith("ds_") else "vmem" if k.startswith(("global_", "buffer_", "flat_", "scratch_")) else
                 "salu" if k.startswith("s_") and not k.startswith(("s_waitcnt", "s_nop", "s_barrier")) else k if k.startswith(("s_waitcnt", "s_nop", "s_barrier")) else "valu")
            groups[g] += v
        print("   groups:", dict(groups.most_common()))
        print("   top:", c.most_common(24))


if __name__ == "__main__":
    main()
