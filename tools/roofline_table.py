"""Per-kernel roofline table from a rocprofv3 kernel trace (rocpd database) of `bench.py` (the two-level 640x480 render):

    python tools/roofline_table.py <x_results.db> [rays_per_launch=307200]

For every kernel of the path: launches, average duration, ALGORITHMIC bytes (HBM-bound kernels, SURVEY 8(d) per-ray figures)
or FLOPs (the fused MLP, reference-literal 1,186,816 FLOP per network evaluation) per launch, the achieved rate and the
fraction of the peak that bounds it (8 TB/s HBM, 157.3 TFLOP/s fp32 matrix; MI355X_MICROARCH.md).  DESIGN.md cites this file.
Launches of the MLP kernel alternate coarse (65 samples per ray) / fine (193) in a two-level render; rays per launch of the
per-ray kernels are read from the dispatch grid (one 64-lane wavefront per ray); the coarse level's compositing is fused with the
inverse CDF (composite_kernel<true, true>), composite_kernel<true, false> is the fine level."""
import sqlite3
import sys

import os

PEAK_HBM, PEAK_MFMA = 8.0e12, 157.3e12
FLOP_LITERAL = 1_186_816
# round 5: bottleneck_layer folded into views_linear[0] (default) -- the kernel EXECUTES 65,536 MACs per evaluation fewer; both fractions are printed
FLOP_PER_EVAL = FLOP_LITERAL - (0 if os.environ.get("AON_BOTTLENECK_FOLD", "") == "0" else 2 * 65_536)
B_COMP = {65: 65 * 20 + 12 + 20 + 65 * 4, 193: 193 * 20 + 12 + 20}
B_PDF = 65 * 4 + 63 * 4 + 193 * 4
B_SAR = 65 * 4          # sample_along_rays: writes 65 t per ray (reads nothing per ray when deterministic)
B_FUSED = 65 * 20 + 12 + 20 + 193 * 4   # fused coarse level (composite_kernel<true, true>): weights stay in registers, t_fine written


def main():
    db = sys.argv[1]
    rays_default = int(sys.argv[2]) if len(sys.argv) > 2 else 307200
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, duration, grid_x, workgroup_x, start from kernels order by start"))
    by = {}
    for name, dur, gx, wx, _ in rows:
        by.setdefault(name, []).append((dur, gx, wx))
    print(f"{'kernel':<58} {'launches':>8} {'avg_us':>10} {'work/launch':>16} {'rate':>14} {'bound':>6} {'frac':>7}")
    for name, ds in sorted(by.items(), key=lambda kv: -sum(d for d, _, _ in kv[1])):
        if "aon::" not in name:
            continue
        short = name.replace("void ", "").split("(")[0][:58]
        groups = {}
        if "mlp_fwd" in name:
            # the per-ray-view-bias instance (4th template argument true) does not execute the view-encoding chunk: 128 x 28 MACs fewer
            flop = FLOP_PER_EVAL - (2 * 128 * 28 if name.replace(" ", "").split("(")[0].endswith(",true,true>") else 0)
            for i, (dur, gx, wx) in enumerate(ds):
                S = 65 if i % 2 == 0 else 193
                groups.setdefault(f"S={S}", []).append((dur, rays_default * S * flop, "mfma"))
        elif "composite_kernel<true, true" in name:   # coarse level: compositing + inverse CDF + merge
            for dur, gx, wx in ds:
                groups.setdefault("S=65 fused", []).append((dur, gx // 64 * B_FUSED, "hbm"))
        elif "composite_kernel" in name:   # two-level render with the fused coarse level: only the fine level launches this one
            for dur, gx, wx in ds:
                groups.setdefault("S=193", []).append((dur, gx // 64 * B_COMP[193], "hbm"))
        elif "sample_pdf" in name:
            for dur, gx, wx in ds:
                groups.setdefault("", []).append((dur, gx // 64 * B_PDF, "hbm"))
        elif "sample_along_rays" in name:
            for dur, gx, wx in ds:
                groups.setdefault("", []).append((dur, gx // 65 * B_SAR, "hbm"))
        elif "sample_t4_kernel" in name:   # four t values per thread: grid_x threads x 16 bytes written
            for dur, gx, wx in ds:
                groups.setdefault("", []).append((dur, gx * 16, "hbm"))
        else:
            groups[""] = [(dur, 0, "-") for dur, _, _ in ds]
        for tag, items in groups.items():
            n = len(items)
            avg = sum(d for d, _, _ in items) / n
            work = sum(w for _, w, _ in items) / n
            bound = items[0][2]
            if bound == "mfma":
                rate = work / (avg * 1e-9)
                print(f"{short + ' ' + tag:<58} {n:>8} {avg / 1e3:>10.2f} {work / 1e12:>12.3f} TFLOP {rate / 1e12:>9.1f} TF/s {bound:>6} {rate / PEAK_MFMA:>7.3f}"
                      f"   (executed; reference-literal {rate / PEAK_MFMA * FLOP_LITERAL * rays_default * int(tag[2:]) / work:.3f})")
            elif bound == "hbm":
                rate = work / (avg * 1e-9)
                print(f"{short + ' ' + tag:<58} {n:>8} {avg / 1e3:>10.2f} {work / 1e6:>13.1f} MB {rate / 1e12:>9.2f} TB/s {bound:>6} {rate / PEAK_HBM:>7.3f}")
            else:
                print(f"{short:<58} {n:>8} {avg / 1e3:>10.2f} {'-':>16} {'-':>14} {'-':>6} {'-':>7}")


if __name__ == "__main__":
    main()
