"""CPU: the C-ABI library loads and exports every symbol include/aon_hip.h declares; argument validation
that needs no GPU; the package fails loudly when the library is absent."""
import ctypes
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "aon_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(aon_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from aon_amd import _lib

    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(_lib.lib, n), f"{n} declared in aon_hip.h but not exported"
    assert sorted(_lib.exported_symbols()) == names, "ctypes binding and header disagree"
    assert _lib.lib.aon_abi_version() == 5


def test_argument_validation_without_gpu():
    from aon_amd import _lib

    lib = _lib.lib
    # null pointers / bad sizes are rejected before any HIP call
    assert lib.aon_pos_enc(None, 4, 0, 10, None, None) == -1
    assert b"null pointer" in lib.aon_last_error()
    assert lib.aon_mlp_fwd(None, None, None, None, None, -1, 65, None, None) == -1
    assert lib.aon_composite(None, 2, None, 1, None, None, 1, 65, 1, 0, None, None, None, None, None) == -1
    assert lib.aon_sample_pdf(None, None, 10, None, None, 0, 1, None, None, None) == -1
    assert lib.aon_render_fwd(None, None, None, None, None, 5, 2.0, 6.0, 1, 3, None, None, 0, None, None, None, None, None,
                              None, None, 0, None) == -1
    # empty problems are a successful no-op
    assert lib.aon_pos_enc(None, 0, 0, 10, None, None) == 0
    assert lib.aon_mlp_fwd(None, None, None, None, None, 0, 65, None, None) == 0
    assert lib.aon_mlp_packed_bytes() == 2_375_680 + 3076 * 4
    per_ray = (65 + 65 + 193 + 4 * 193 + 128) * 4     # t_c, w_c, t_f, raw, the per-ray view bias
    assert lib.aon_render_workspace_bytes(1000) >= 1000 * per_ray
    assert lib.aon_render_workspace_bytes(1000) < 1000 * per_ray + 5 * 256 + 1


def test_ops_reject_cpu_tensors():
    import pytest
    import torch

    from aon_amd import ops

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.pos_enc(torch.zeros(2, 3), 0, 10)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.volumetric_rendering(torch.zeros(1, 65, 3), torch.zeros(1, 65, 1), torch.zeros(1, 65), torch.zeros(1, 3), True)


def test_missing_library_fails_loudly(tmp_path):
    """A copy of the package without libaon_hip.so must refuse to import its binding (no silent fallback)."""
    import shutil

    pkg = tmp_path / "pkgcopy"
    shutil.copytree(os.path.join(ROOT, "articulated-object-nerf_amd"), pkg,
                    ignore=shutil.ignore_patterns("*.so", "build", "__pycache__"))
    code = (
        "import importlib.util, sys\n"
        f"spec = importlib.util.spec_from_file_location('pk', r'{pkg}/__init__.py', submodule_search_locations=[r'{pkg}'])\n"
        "m = importlib.util.module_from_spec(spec); sys.modules['pk'] = m; spec.loader.exec_module(m)\n"
        "try:\n"
        "    import pk.ops\n"
        "except ImportError as e:\n"
        "    print('LOUD:', e); sys.exit(0)\n"
        "sys.exit(1)\n"
    )
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "LOUD:" in r.stdout and "no CPU/eager fallback" in r.stdout, r.stdout + r.stderr


def test_two_call_training_entries_validate_without_gpu():
    """aon_render_fwd_train / aon_render_bwd and the articulated twins: argument checking and workspace
    sizing happen before any HIP call; aon_profile_class knows its classes."""
    import ctypes as C

    from aon_amd import _lib

    lib = _lib.lib
    van, art = lib.aon_train_workspace_bytes(4096, 0, 2), lib.aon_train_workspace_bytes(4096, 1, 2)
    planes_van = 2528 * (128 * ((4096 * 65 + 127) // 128) + 128 * ((4096 * 193 + 127) // 128)) * 4
    assert van > planes_van and art > van and lib.aon_train_workspace_bytes(2048, 0, 2) < van
    # sized by the levels in use: one level = the 65-sample share; the backward's temporaries are a separate scratch
    assert lib.aon_train_workspace_bytes(4096, 1, 1) < 0.3 * art and lib.aon_train_scratch_bytes(4096, 1, 1) < 0.45 * lib.aon_train_scratch_bytes(4096, 1, 2)
    assert 10e9 < lib.aon_train_scratch_bytes(4096, 1, 2) < 16e9 and 13e9 < art < 17e9
    assert lib.aon_render_fwd_train(None, None, None, None, None, 0, 2.0, 6.0, 1, 2, None, None, 0, None, None, None, None, None, None,
                                    None, 0, None) != 0 and b"bad size" in lib.aon_last_error()
    assert lib.aon_render_fwd_train(None, None, None, None, None, 8, 2.0, 6.0, 1, 3, None, None, 0, None, None, None, None, None, None,
                                    None, 0, None) != 0
    assert lib.aon_render_fwd_train(None, None, None, None, None, 8, 2.0, 6.0, 1, 2, None, None, 0, None, None, None, None, None, None,
                                    None, 0, None) != 0 and b"null" in lib.aon_last_error()
    assert lib.aon_render_bwd(None, None, None, None, None, 8, 1, 2, None, None, None, None, None, None, 0, None, 0, None) != 0 and b"null" in lib.aon_last_error()
    assert lib.aon_art_render_bwd(None, None, None, None, None, 0, 1, 2, None, None, None, None, None, None, None, None, None, None, None, None,
                                  None, None, 0, None, 0, None) != 0
    assert lib.aon_set_bwd_overlap(0) == 0 and lib.aon_set_bwd_overlap(1) == 0
    ms, n, u = C.c_double(-1), C.c_int64(-1), C.c_int64(-1)
    for cls in range(8):
        assert lib.aon_profile_class(cls, C.byref(ms), C.byref(n), C.byref(u)) == 0 and ms.value == 0.0 and n.value == 0
    assert lib.aon_profile_class(8, C.byref(ms), C.byref(n), C.byref(u)) != 0
    # fused coarse level (aon_composite_pdf): validation before any HIP call; the fusion switch of the whole-path entry points
    assert lib.aon_composite_pdf(None, None, None, 4, 1, 1, None, 0, None, None, None, None, None, None) != 0 and b"null" in lib.aon_last_error()
    assert lib.aon_composite_pdf(None, None, None, 4, 1, 3, None, 0, None, None, None, None, None, None) != 0 and b"bad size" in lib.aon_last_error()
    assert lib.aon_composite_pdf(None, None, None, 4, 1, 1, None, 64, None, None, None, None, None, None) != 0
    assert lib.aon_composite_pdf(None, None, None, 0, 1, 1, None, 0, None, None, None, None, None, None) == 0   # empty batch
    assert lib.aon_set_coarse_fusion(0) == 0 and lib.aon_set_coarse_fusion(1) == 0
    assert lib.aon_ray_radii(None, None, 8, 8, None, None) != 0 and b"null" in lib.aon_last_error()


def test_wgrad_plan_invariants_without_gpu():
    """The grouped weight-gradient launch's plan (csrc/aon_wgrad.h:wg_make_plan, host-only).  Round 4: the steps of all jobs form one
    work line priced in cost units and workgroup i of G owns the steps that START in its 1/G of the line.  For both networks, sample
    counts from one pass to a 4096-ray fine level, and compute-unit counts from 1 to 304: the launch never exceeds the compute units
    (ONE co-resident round) nor the steps on the line; the workgroups that write partials for a job are exactly a contiguous range;
    their step ranges tile [0, steps) of the job without gap or overlap (the kernel's own arithmetic, restated on the host by
    aon_wgrad_plan_segment); partial regions do not overlap and fit aon_wgrad_workspace_bytes(); every workgroup's share of the
    cost is the same to within one step of the widest job."""
    import ctypes as C

    from aon_amd import _lib

    lib = _lib.lib
    blk = {0: 256 * 256, 1: 128 * 128, 2: 256 * 64, 3: 128 * 256, 4: 128 * 32}
    cost = {0: 16384, 1: 4280, 2: 4620, 3: 8300, 4: 1510}
    # both forms of the networks (round 5): with bottleneck_layer folded into views_linear[0] a level has one job fewer (the
    # 256x256 bottleneck job and the 128x256 view job on the bottleneck output become ONE 128x256 job on the layer-7 output)
    cases = []
    for fold in (1, 0):
        cases += [(fold, 0, 12 - fold), (fold, 1, 18 - fold)]
    for fold, art, njobs in cases:
        lib.aon_set_bottleneck_fold(fold)
        for Np in (128, 640, 4096 * 65 + 0, 128 * ((4096 * 193 + 127) // 128)):
            if Np % 128:
                Np += 128 - Np % 128
            for cus in (1, 5, 64, 255, 256, 304, 1000):
                out = (C.c_int32 * (6 * 20))()
                ws = C.c_int64(0)
                n = lib.aon_wgrad_plan(art, Np, cus, out, 20, C.byref(ws))
                assert n == njobs, (art, Np, cus, n, lib.aon_last_error())
                jobs = [tuple(out[6 * j: 6 * j + 6]) for j in range(n)]
                nsteps, regions = Np // 32, []
                G = max(j[1] + j[2] for j in jobs)
                assert G <= min(cus, 304, nsteps * njobs)
                load = [0] * G
                be = (C.c_int32 * 2)()
                full = Np <= 640 or cus in (5, 256)     # every (job, workgroup) pair where that is cheap
                for j, (kind, first, count, steps, part_off, nparts) in enumerate(jobs):
                    assert steps == nsteps and count >= 1 and first >= 0 and first + count <= G
                    regions.append((part_off, part_off + nparts * blk[kind]))
                    if not full:
                        continue
                    nxt = 0
                    for wg in range(G):
                        r = lib.aon_wgrad_plan_segment(art, Np, cus, j, wg, be)
                        assert r == (1 if first <= wg < first + count else 0), (j, wg, r)
                        if r:
                            assert be[0] == nxt and be[1] >= be[0]            # contiguous tiling, in workgroup order
                            nxt = be[1]
                            load[wg] += (be[1] - be[0]) * cost[kind]
                        else:
                            assert be[0] == be[1]                             # owns nothing of a job it writes no partial for
                    assert nxt == nsteps
                regions.sort()
                assert all(a[1] <= b[0] for a, b in zip(regions, regions[1:])) and regions[0][0] >= 0
                assert ws.value <= lib.aon_wgrad_workspace_bytes() and ws.value >= regions[-1][1] * 4
                if full:
                    assert max(load) - min(load) <= 2 * 16384, (art, Np, cus, max(load), min(load))
                    if cus == 256 and Np > 100_000:
                        assert G == 256 and max(load) / min(load) < 1.01
    lib.aon_set_bottleneck_fold(1)
    out = (C.c_int32 * 120)()
    assert lib.aon_wgrad_plan(1, 100, 256, out, 20, None) < 0 and b"multiple of 32" in lib.aon_last_error()
    assert lib.aon_wgrad_plan(1, 1024, 256, out, 3, None) < 0


def test_constructor_option_entry_points_validate_without_gpu():
    """aon_render_opts / aon_mlp_geometry (round 3): defaults, size queries and argument checks that need no GPU."""
    import ctypes as C

    from aon_amd import _lib

    lib = _lib.lib
    st = _lib.RenderOptsC()
    lib.aon_render_opts_init(C.byref(st))
    assert (st.num_coarse_samples, st.num_fine_samples, st.lindisp, st.noise_std) == (64, 128, 0, 0.0)
    assert abs(st.rgb_scale - 1.002) < 1e-6 and abs(st.rgb_shift - 0.001) < 1e-7 and st.sigma_bias == -1.0
    assert lib.aon_render_workspace_bytes_ex(1000, C.byref(st)) == lib.aon_render_workspace_bytes(1000)
    assert lib.aon_train_workspace_bytes_ex(512, 1, 2, C.byref(st)) == lib.aon_train_workspace_bytes(512, 1, 2)
    st.num_coarse_samples, st.num_fine_samples = 32, 48
    per_ray = (33 + 33 + 81 + 4 * 81 + 128) * 4
    assert 1000 * per_ray <= lib.aon_render_workspace_bytes_ex(1000, C.byref(st)) <= 1000 * per_ray + 5 * 256
    assert lib.aon_train_workspace_bytes_ex(512, 0, 2, C.byref(st)) < lib.aon_train_workspace_bytes(512, 0, 2)
    st.num_coarse_samples = 1
    assert lib.aon_render_workspace_bytes_ex(1000, C.byref(st)) == -1 and b"num_coarse_samples" in lib.aon_last_error()
    st.num_coarse_samples, st.num_fine_samples = 900, 20000
    assert lib.aon_render_workspace_bytes_ex(1000, C.byref(st)) == -1 and b"too large" in lib.aon_last_error()
    # general-size inverse CDF: size rules
    assert lib.aon_sample_pdf_n(None, None, 5, None, None, 0, 4, 1, 8, 2, None, None, None) == -1          # < 2 bins
    assert lib.aon_sample_pdf_n(None, None, 5, None, None, 0, 4, 10, 8, 9, None, None, None) == -1         # bins NULL needs num_t = num_bins + 1
    assert lib.aon_sample_pdf_n(None, None, 9, None, None, 0, 0, 10, 8, 11, None, None, None) == 0         # empty problem
    # NeRFMLP geometry
    g = _lib.MlpGeometryC()
    lib.aon_mlp_geometry_init(C.byref(g))
    assert tuple(getattr(g, n) for n, _ in g._fields_) == (0, 10, 4, 8, 256, 1, 128, 4, 3, 3, 3, 1)
    assert lib.aon_gmlp_param_count(C.byref(g)) == 24
    g.netdepth, g.netdepth_condition = 6, 3
    assert lib.aon_gmlp_param_count(C.byref(g)) == 2 * (6 + 3 + 3)
    g.netdepth = 5                                       # the skip would land on the last trunk layer: the reference fails there too
    assert lib.aon_gmlp_param_count(C.byref(g)) == -1 and b"last trunk layer" in lib.aon_last_error()
    g.netdepth, g.netwidth = 6, 0
    assert lib.aon_gmlp_param_count(C.byref(g)) == -1
    lib.aon_mlp_geometry_init(C.byref(g))
    lib.aon_render_opts_init(C.byref(st))
    n = 256
    acts = 258                                           # what the training forward keeps: E, H x 8, bott, V (floats per sample) ...
    per_sample = (64 + 8 * 256 + 256 + 128 + 1 + 4 + 3) * 4   # the engine's own encoding rows are padded to whole 8-float groups (63 -> 64)
    ws = lib.aon_grender_train_workspace_bytes(C.byref(g), n, 2, C.byref(st))
    assert n * acts * per_sample <= ws <= n * acts * per_sample + n * (32 * 2 + 65) * 4 + 64 * 256
    assert lib.aon_grender_train_scratch_bytes(C.byref(g), n, 2, C.byref(st)) > 0
    assert lib.aon_grender_workspace_bytes(C.byref(g), n, C.byref(st)) > 0
    # the general engine takes its degrees from the geometry: opts' degree fields filled to match a (1, 12, 5) network are accepted
    # there and still refused by the fused entry points
    g.min_deg_point, g.max_deg_point, g.deg_view = 1, 12, 5
    st.min_deg_point, st.max_deg_point, st.deg_view = 1, 12, 5
    assert lib.aon_grender_workspace_bytes(C.byref(g), n, C.byref(st)) > 0
    assert lib.aon_grender_train_workspace_bytes(C.byref(g), n, 2, C.byref(st)) > 0
    assert lib.aon_render_workspace_bytes_ex(n, C.byref(st)) == -1 and b"frequency levels" in lib.aon_last_error()
    lib.aon_mlp_geometry_init(C.byref(g))
    lib.aon_render_opts_init(C.byref(st))
    g.input_ch = 2                                       # NeRF.forward encodes 3-vectors
    assert lib.aon_grender_fwd(C.byref(g), None, None, None, None, None, 8, 2.0, 6.0, 1, 2, None, None, 0, None, None, None, None, None, None,
                               None, 0, None, None) == -1
    assert b"input_ch" in lib.aon_last_error()


def test_bottleneck_fold_switch_without_gpu():
    """Round 5: the form switch of the pack calls (include/aon_hip.h, aon_set_bottleneck_fold) is host state: default folded.  Round 6
    (ADVICE r5): a pointer this process never packed or declared has NO form (-1) whatever the switch says -- the launchers refuse it --
    until its owner declares one (the form of a copy of a packed buffer); the switch never changes a form that was declared."""
    import ctypes as C

    from aon_amd import _lib

    lib = _lib.lib
    before = lib.aon_get_bottleneck_fold()
    try:
        lib.aon_set_bottleneck_fold(1)
        assert lib.aon_get_bottleneck_fold() == 1 and lib.aon_stream_form(C.c_void_p(0x1000)) == -1 and lib.aon_stream_is_folded(C.c_void_p(0x1000)) == 0
        lib.aon_set_bottleneck_fold(0)
        assert lib.aon_get_bottleneck_fold() == 0 and lib.aon_stream_form(C.c_void_p(0x1000)) == -1
        assert lib.aon_declare_stream_form(C.c_void_p(0x1000), 1) == 0 and lib.aon_stream_form(C.c_void_p(0x1000)) == 1
        lib.aon_set_bottleneck_fold(1)
        lib.aon_set_bottleneck_fold(0)
        assert lib.aon_stream_form(C.c_void_p(0x1000)) == 1 and lib.aon_stream_is_folded(C.c_void_p(0x1000)) == 1     # declared: the switch does not touch it
        assert lib.aon_declare_stream_form(C.c_void_p(0x1000), 0) == 0 and lib.aon_stream_form(C.c_void_p(0x1000)) == 0
        assert lib.aon_declare_stream_form(C.c_void_p(0x1000), 7) == -1 and lib.aon_declare_stream_form(None, 1) == -1
        assert lib.aon_stream_form(C.c_void_p(0x2000)) == -1
        # either form fits the buffers whose sizes the library reports (the sizes do not depend on the switch)
        sizes = (lib.aon_mlp_packed_bytes(), lib.aon_bwd_packed_bytes(), lib.aon_art_packed_bytes(), lib.aon_art_bwd_packed_bytes())
        lib.aon_set_bottleneck_fold(1)
        assert sizes == (lib.aon_mlp_packed_bytes(), lib.aon_bwd_packed_bytes(), lib.aon_art_packed_bytes(), lib.aon_art_bwd_packed_bytes())
        assert lib.aon_bwd_packed_bytes() >= 60 * 32768 + (128 * 256 + 256 * 256 + 256 + 128 * 256 + 128) * 4   # folded stream + raw copies + W', b'
    finally:
        lib.aon_set_bottleneck_fold(before)


def test_round6_entry_points_validate_without_gpu():
    """aon_adam_step / aon_code_library_fwd / _bwd (ABI 5): argument validation happens on the host, before any launch."""
    import ctypes as C

    from aon_amd import _lib

    lib = _lib.lib
    p = C.c_void_p(0x1000)
    assert lib.aon_adam_step(p, p, p, p, 0, 5e-4, 0.9, 0.999, 1e-8, 1, None) == 0                      # nothing to do
    assert lib.aon_adam_step(p, p, p, p, -1, 5e-4, 0.9, 0.999, 1e-8, 1, None) == -1
    assert lib.aon_adam_step(p, p, p, p, 16, 5e-4, 0.9, 0.999, 1e-8, 0, None) == -1 and b"step" in lib.aon_last_error()   # torch's state["step"] AFTER the update
    assert lib.aon_adam_step(None, p, p, p, 16, 5e-4, 0.9, 0.999, 1e-8, 1, None) == -1 and b"null" in lib.aon_last_error()
    for bad in ((-1.0, 0.9, 0.999, 1e-8), (5e-4, 1.0, 0.999, 1e-8), (5e-4, 0.9, -0.1, 1e-8), (5e-4, 0.9, 0.999, -1.0), (float("nan"), 0.9, 0.999, 1e-8)):
        assert lib.aon_adam_step(p, p, p, p, 16, *bad, 1, None) == -1 and b"hyper-parameter" in lib.aon_last_error()
    three = (C.c_void_p * 3)(0x1000, 0x2000, 0x3000)
    rows, dims = (C.c_int * 3)(2, 2, 10), (C.c_int * 3)(128, 128, 32)
    for fn in (lib.aon_code_library_fwd, lib.aon_code_library_bwd):
        assert fn(None, three, rows, dims, three, None) == -1
        assert fn(three, three, (C.c_int * 3)(2, 0, 10), dims, three, None) == -1 and b"table size" in lib.aon_last_error()
        assert fn((C.c_void_p * 3)(0x1000, 0, 0x3000), three, rows, dims, three, None) == -1
    # aon_art_pack_step: degrees, null pointers and alignment are judged before anything is launched
    forty = (C.c_void_p * 40)(*[0x1000 + 64 * i for i in range(40)])
    a = C.c_void_p(0x4000)
    ok = (forty, forty, p, p, p, 0, 10, 4, a, a, a, a, a, a, None)
    swap = lambda i, v: ok[:i] + (v,) + ok[i + 1:]     # noqa: E731
    assert lib.aon_art_pack_step(*swap(6, 11)) == -1                                          # more than ten position levels
    assert lib.aon_art_pack_step(*swap(0, None)) == -1 and b"null" in lib.aon_last_error()
    assert lib.aon_art_pack_step(*swap(11, None)) == -1 and b"null" in lib.aon_last_error()   # the fine forward stream is not optional
    holed = (C.c_void_p * 40)(*[0 if i == 17 else 0x1000 + 64 * i for i in range(40)])
    assert lib.aon_art_pack_step(*swap(1, holed)) == -1 and b"null parameter" in lib.aon_last_error()
    assert lib.aon_art_pack_step(*swap(13, C.c_void_p(0x4004))) == -1 and b"16-byte" in lib.aon_last_error()
    # aon_vanilla_pack_step likewise
    t24 = (C.c_void_p * 24)(*[0x1000 + 64 * i for i in range(24)])
    okv = (t24, t24, 0, 10, 4, a, a, a, a, None)
    swv = lambda i, v: okv[:i] + (v,) + okv[i + 1:]     # noqa: E731
    assert lib.aon_vanilla_pack_step(*swv(3, 11)) == -1 and b"frequency levels" in lib.aon_last_error()
    assert lib.aon_vanilla_pack_step(*swv(1, None)) == -1 and b"null" in lib.aon_last_error()
    assert lib.aon_vanilla_pack_step(*swv(7, None)) == -1 and b"null" in lib.aon_last_error()      # the fine forward stream is not optional
    assert lib.aon_vanilla_pack_step(*swv(0, (C.c_void_p * 24)(*[0 if i == 5 else 0x1000 for i in range(24)]))) == -1 and b"null parameter" in lib.aon_last_error()
    assert lib.aon_vanilla_pack_step(*swv(6, C.c_void_p(0x4008))) == -1 and b"16-byte" in lib.aon_last_error()


def test_param_arena_mechanics_on_cpu():
    """aon_amd/arena.py without a GPU: the arena re-homes parameters (values, names, shapes kept), gradient slots, the claim rule for two
    live graphs, and ArenaAdam's refusal to run on CPU tensors (no fallback)."""
    import pytest
    import torch

    from aon_amd.arena import ALIGN, ArenaAdam, ParamArena, arena_of, grad_views

    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    before = {k: v.clone() for k, v in net.state_dict().items()}
    arena = ParamArena(net)
    assert arena.intact() and all(o % ALIGN == 0 for o in arena.offsets) and arena.total == 4 * ALIGN
    assert all(torch.equal(v, before[k]) for k, v in net.state_dict().items())
    net.load_state_dict({k: v + 1 for k, v in before.items()})       # load_state_dict copies INTO the views
    assert arena.intact() and torch.equal(arena.flat[:35].view(7, 5), before["0.weight"] + 1)
    params = list(net.parameters())
    ao = arena_of(params)
    assert ao is not None and ao[0] is arena and ao[1] == arena.offsets
    views = grad_views(ao, [tuple(p.shape) for p in params])
    assert all(v.data_ptr() == arena.grad.data_ptr() + 4 * o for v, o in zip(views, arena.offsets))
    tok = arena.claim(arena.offsets)
    assert tok is not None and arena.claim(arena.offsets[:1]) is None          # a second live graph over a held slot gets none
    tok.done = True
    tok2 = arena.claim(arena.offsets)
    assert tok2 is not None
    del tok2                                                                   # a graph dropped without a backward frees its claim
    assert arena.claim(arena.offsets) is not None
    with pytest.raises(ValueError):
        ParamArena(net)                                                        # already lives in an arena
    assert ParamArena.for_modules(net) is arena                                # ... which `configure_optimizers()` called again reuses
    with pytest.raises(ValueError):
        ParamArena.for_modules([net[0]])                                       # part of another arena: refused, not silently re-homed
    net[0].weight.data = torch.zeros(7, 5)                                     # re-homed behind the arena's back
    assert not arena.intact() and arena_of(params) is None
    opt = ArenaAdam(ParamArena(torch.nn.Linear(3, 2)))
    opt.arena.params[0].grad = torch.ones(2, 3)
    with pytest.raises(RuntimeError, match="cuda"):
        opt.step()                                                             # the product path has no CPU form
