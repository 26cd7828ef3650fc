"""Register / scratch / occupancy table of every kernel of the library, from hipcc's own resource remarks:

    python tools/kernel_resources.py > profiles/r02_kernel_resources.txt

(compiles each translation unit with -Rpass-analysis=kernel-resource-usage through tools/resusage.sh; no GPU needed)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["aon_mlp.hip", "aon_mlp_art.hip", "aon_train.hip", "aon_train_art.hip", "aon_render.hip", "aon_gmlp.hip", "aon_fold.hip"]
PAT = re.compile(r"Name: (\S+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+)")


def main():
    print("# hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage (tools/kernel_resources.py)")
    print(f"{'kernel':<96} {'VGPR':>5} {'AGPR':>5} {'scratch B/lane':>15} {'waves/SIMD':>11}")
    seen = set()
    for f in FILES:
        txt = subprocess.run([os.path.join(ROOT, "tools", "resusage.sh"), f], capture_output=True, text=True).stdout
        for line in txt.splitlines():
            m = PAT.search(line)
            if not m or m.group(1) in seen:
                continue
            seen.add(m.group(1))
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(aon::\w+\)$", "", name).replace("void ", "")
            print(f"{name[:96]:<96} {m.group(2):>5} {m.group(3):>5} {m.group(4):>15} {m.group(5):>11}")


if __name__ == "__main__":
    sys.exit(main())
