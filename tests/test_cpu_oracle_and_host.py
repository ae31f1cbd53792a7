"""CPU suite (-m "not gpu"): oracle vs golden fixtures, host logic, C-ABI exports.  No kernel runs."""
import ctypes
import hashlib
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import GOLD, ROOT, rel_l2


# ---- oracle pinned to the reference's golden outputs ---------------------------------------------
def _small_sd():
    from meshdiffusion_amd import synth
    from meshdiffusion_amd.lib.diffusion.models import ddpm_res64, utils as mutils  # noqa: F401
    cfg = synth.small_config(); cfg.device = torch.device("cpu")
    tmpl = mutils.create_model(cfg, use_parallel=False).state_dict()
    R = cfg.data.image_size
    return cfg, synth.sensitised_state_dict(tmpl, seed=1234, grid_mask=synth.synthetic_grid_mask(R))


def test_oracle_unet_matches_reference_golden():
    from meshdiffusion_amd import synth
    from oracle import unet_oracle as uo
    cfg, sd = _small_sd()
    gold = np.load(os.path.join(GOLD, "unet_small.npz"))
    x = synth.synthetic_inputs(2, 4, cfg.data.image_size, seed=int(gold["x_seed"]))
    with torch.no_grad():
        y = uo.unet_res64_forward(sd, synth.oracle_cfg(cfg), x, torch.tensor(gold["labels"]))
    assert rel_l2(y, gold["y"]) < 1e-5


def test_oracle_sampler_matches_reference_golden():
    from meshdiffusion_amd import synth
    from oracle import unet_oracle as uo
    cfg, sd = _small_sd()
    gold = np.load(os.path.join(GOLD, "sampler_small.npz"))
    R, K = cfg.data.image_size, int(gold["K"])
    mask = synth.synthetic_grid_mask(R).view(1, R, R, R)
    torch.manual_seed(int(gold["uncond_seed"]))
    x = torch.randn(2, 4, R, R, R) * mask
    ts = torch.linspace(1.0, 1e-3, 1000)
    with torch.no_grad():
        xm = x
        for i in range(K):
            e = uo.unet_res64_forward(sd, synth.oracle_cfg(cfg), x, torch.ones(2) * ts[i] * 999)
            x, xm = uo.ancestral_step(x, e, torch.randn_like(x), ts[i], mask)
    assert rel_l2(xm, gold["uncond"]) < 1e-5


def test_oracle_dmtet_matches_reference_golden():
    from oracle import dmtet_oracle
    from oracle.gen_golden import dmtet_cases
    tet = np.load(os.path.join(GOLD, "64_tets_cropped.npz"))
    gold = np.load(os.path.join(GOLD, "dmtet.npz"))
    pos, cases = dmtet_cases(tet["vertices"])
    for name in ("sphere", "smooth", "box_zeros"):
        v, f, _ = dmtet_oracle.marching_tets(pos.numpy(), cases[name].numpy(), tet["indices"])
        assert hashlib.sha256(np.ascontiguousarray(f).tobytes()).hexdigest() == str(gold[f"{name}_faces_sha"])
        assert hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest() == str(gold[f"{name}_verts_sha"])


def test_static_edge_table_formulation_equals_oracle_on_cpu():
    """The product's formulation (static sorted edge table + prefix sums), emulated with numpy on the
    tables built by TetTables, gives the reference's faces: checks the host tables without a GPU."""
    from meshdiffusion_amd.dmtet import TetTables
    from oracle import dmtet_oracle
    from oracle.gen_golden import dmtet_cases
    tet = np.load(os.path.join(GOLD, "64_tets_cropped.npz"))
    tb = TetTables(torch.as_tensor(tet["indices"]), "cpu")
    assert tb.n_edges == 195331 and tb.n_tets == 159330
    pos, cases = dmtet_cases(tet["vertices"])
    sdf = cases["sinus"].numpy()
    occ = sdf > 0
    e = tb.edges.numpy()
    cross = occ[e[:, 0]] != occ[e[:, 1]]
    vid = np.where(cross, np.cumsum(cross) - 1, -1)
    t = tb.tets.numpy()
    idx = (occ[t] * (2 ** np.arange(4))[None]).sum(-1)
    ntri = dmtet_oracle.NUM_TRIANGLES[idx]
    ev = vid[tb.tet_edges.numpy()]
    f1 = np.take_along_axis(ev[ntri == 1], dmtet_oracle.TRIANGLE_TABLE[idx[ntri == 1]][:, :3], 1).reshape(-1, 3)
    f2 = np.take_along_axis(ev[ntri == 2], dmtet_oracle.TRIANGLE_TABLE[idx[ntri == 2]][:, :6], 1).reshape(-1, 3)
    _, fo, _ = dmtet_oracle.marching_tets(pos.numpy(), sdf, tet["indices"])
    assert np.array_equal(np.concatenate([f1, f2]), fo)


def test_grid_mask_from_tets():
    from meshdiffusion_amd.dmtet import grid_mask_from_tets
    tet = np.load(os.path.join(GOLD, "64_tets_cropped.npz"))
    m = grid_mask_from_tets(tet["vertices"], 64)
    assert int(m.sum()) == 30512 and float(m[63].sum()) == 0.0


# ---- C ABI ------------------------------------------------------------------------------------------
def test_c_abi_exports_every_declared_symbol(hip_lib):
    from meshdiffusion_amd import _lib
    header = open(os.path.join(ROOT, "include", "meshdiffusion_hip.h")).read()
    declared = set(re.findall(r"\b(md_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    assert hip_lib.md_abi_version() == _lib.ABI_VERSION
    m = re.search(r"#define MD_ABI_VERSION (\d+)", header)
    assert m and int(m.group(1)) == _lib.ABI_VERSION
    info = _lib.cfg_info(_lib.CFG_C3_128)
    assert info["taps"] == 27 and info["lds_bytes"] <= 160 * 1024 and info["threads"] == 512
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert " T md_gemm_conv" in nm


def test_struct_layout_matches_header():
    from meshdiffusion_amd._lib import MdGemmConvArgs
    # 5 pointers, 1 float + 12 int32 (+pad), 4 int64, 1 pointer, 2 int32, 1 pointer, 2 int32, 2 pointers, 1 int64, 2 int32
    # -> natural alignment, no surprises
    size = 5 * 8 + 13 * 4 + 4 + 4 * 8 + 8 + 2 * 4 + 8 + 2 * 4 + 2 * 8 + 8 + 2 * 4
    assert ctypes.sizeof(MdGemmConvArgs) == size
    assert MdGemmConvArgs.stats.offset == size - 48 and MdGemmConvArgs.stagger.offset == size - 40
    assert MdGemmConvArgs.b2.offset == size - 32 and MdGemmConvArgs.b_silu.offset == size - 4
    assert MdGemmConvArgs.a_bstride.offset % 8 == 0


def test_bad_arguments_are_rejected_without_a_gpu(hip_lib):
    from meshdiffusion_amd._lib import MdGemmConvArgs
    assert hip_lib.md_gemm_conv(None, None) == -1
    a = MdGemmConvArgs()
    assert hip_lib.md_gemm_conv(ctypes.byref(a), None) == -1
    assert hip_lib.md_packed_weight_bytes(128, 128, 27, 128, 32) == 128 * 128 * 27 * 4
    assert hip_lib.md_marching_tets_workspace_bytes(0, 5, 5) < 0
    # edge -> vertex table + one counter per 1024-edge chunk + two per 1024-tet chunk, per mesh
    assert hip_lib.md_marching_tets_workspace_bytes(32, 195331, 159330) == 32 * (195331 + 191 + 2 * 156) * 4
    # Winograd path: operand T = 8 bytes per input element, weight tiles = 36/27 of the fp32 weight; shapes it does not take
    assert hip_lib.md_wino_operand_bytes(8, 128, 64, 64, 64) == 8 * 128 * 64 ** 3 * 8
    assert hip_lib.md_wino_weight_bytes(128, 256) == 128 * 256 * 36 * 4
    assert hip_lib.md_wino_operand_bytes(1, 12, 8, 8, 8) < 0 and hip_lib.md_wino_operand_bytes(1, 16, 8, 8, 7) < 0
    assert hip_lib.md_wino_weight_bytes(96, 64) < 0 and hip_lib.md_wino_weight_bytes(128, 48) < 0
    assert hip_lib.md_wino_prep(None, None, 8, 0, None, 0, 0, None, 1, 8, 8, 8, 0.0, 0, None) == -1
    assert hip_lib.md_conv3_wino(None, None, None, None, 0, None, 0, None, 1, 32, 128, 8, 8, 8, 0, None) == -1
    # SiLU is applied together with the folded GroupNorm affine only: silu = 1 without `ac` is an argument error
    buf = ctypes.create_string_buffer(64)      # any non-null pointers: the check comes before any launch
    p = ctypes.cast(buf, ctypes.c_void_p)
    for fn in (hip_lib.md_wino_prep, hip_lib.md_wino_prep_v2):
        assert fn(p, None, 8, 0, None, 1, 0, p, 1, 8, 8, 8, 0.0, 0, None) == -1
    assert hip_lib.md_wino_prep_dual(p, None, 8, 0, None, 0, 0, p, None, None, 1, 8, 8, 8, 0.0, 0, None) == -1     # no second output
    # Winograd weight gradient: workspace size, shapes it does not take (channels % 128, W not in {32, 64}), K-range bound
    assert hip_lib.md_wgrad_wino_workspace_bytes(128, 256, 10) == 10 * 36 * 128 * 256 * 4
    assert hip_lib.md_wgrad_wino_workspace_bytes(96, 128, 1) < 0
    big = 10 * 36 * 128 * 256 * 4
    assert hip_lib.md_wgrad_wino(p, p, p, p, big, 8, 128, 256, 64, 64, 16, 10, 27 * 256, 27, 1, None) == -2    # MD_ERR_UNSUPPORTED
    assert hip_lib.md_wgrad_wino(p, p, p, p, big, 8, 96, 256, 64, 64, 64, 10, 27 * 256, 27, 1, None) == -2
    assert hip_lib.md_wgrad_wino(p, p, p, p, 16, 8, 128, 256, 64, 64, 64, 10, 27 * 256, 27, 1, None) == -1     # workspace too small
    assert hip_lib.md_wgrad_wino(p, p, p, p, big, 1, 128, 256, 64, 64, 64, 64, 27 * 256, 27, 1, None) == -1    # ksplit > B (D - 1)


def test_round3_entry_points_reject_bad_arguments_without_a_gpu(hip_lib):
    """md_conv3_s2 / md_conv3_head / md_pack_batch check their arguments before any launch; MdPackJob matches the header; the
    host-side shape predicates of the new kernels."""
    from meshdiffusion_amd import _lib, hip_ops
    buf = ctypes.create_string_buffer(64)
    p = ctypes.cast(buf, ctypes.c_void_p)
    s2 = hip_lib.md_conv3_s2
    assert s2(None, p, p, None, 0, None, 1, 32, 64, 64, 8, 8, 8, None) == -1          # no input
    assert s2(p, p, p, None, 0, None, 1, 48, 64, 64, 8, 8, 8, None) == -2             # cin % 32
    assert s2(p, p, p, None, 0, None, 1, 32, 64, 64, 6, 8, 8, None) == -2             # D % 4
    assert s2(p, p, p, None, 0, None, 1, 32, 64, 60, 8, 8, 8, None) == -1             # rows_alloc < rows
    hd = hip_lib.md_conv3_head
    assert hd(p, None, p, p, 1, 64, 16, 8, 8, 8, None) == -1                          # the folded affine is required
    assert hd(p, p, p, p, 1, 48, 16, 8, 8, 8, None) == -2 and hd(p, p, p, p, 1, 64, 40, 8, 8, 8, None) == -2
    assert hd(p, p, p, p, 1, 64, 16, 8, 8, 12, None) == -2
    pb = hip_lib.md_pack_batch
    assert pb(None, 1, 1, 0, None) == -1 and pb(p, 0, 1, 0, None) == -1 and pb(p, 1, 0, 0, None) == -1
    assert ctypes.sizeof(_lib.MdPackJob) == 2 * 8 + 5 * 8 + 8 * 4 and _lib.MdPackJob.rows.offset == 56
    header = open(os.path.join(ROOT, "include", "meshdiffusion_hip.h")).read()
    body = header[header.index("typedef struct MdPackJob {"):header.index("} MdPackJob;")]
    assert [f.strip(" ;") for f in body.split("\n")[1:] if f.strip()] == [
        "const float* w", "void* out", "int64_t s_row, s_k, s_tap, n_items, block0", "int32_t rows, kdim, taps, nt, kc, prec, flip, kind"]
    # shape predicates (pure host logic)
    assert hip_ops.conv3_s2_ok(128, 128, 32, 8) and hip_ops.conv3_s2_ok(128, 128, 16, 8)
    assert not hip_ops.conv3_s2_ok(256, 256, 8, 8)          # 32 workgroups: the split-K generic tile is faster
    assert not hip_ops.conv3_s2_ok(128, 128, 4, 8) and not hip_ops.conv3_s2_ok(128, 48, 32, 8)
    assert hip_ops.conv3_s2_ok(128, 128, 32, 1) and not hip_ops.conv3_s2_ok(128, 128, 16, 1)
    assert hip_ops.conv3_head_ok(12, 128, 64) and not hip_ops.conv3_head_ok(20, 64, 12) and not hip_ops.conv3_head_ok(36, 128, 64)


def test_pack_site_registry_rebuilds_stale_entries_and_does_not_keep_layers_alive():
    """hip_ops.prewarm_packs (start of a training forward): the cache entries used since the last prewarm are rebuilt once
    when the parameters changed (PARAM_EPOCH), entries are re-registered by their next use, and the registry holds the layers
    weakly (the builders reference their layer: a strong table would keep every model ever built alive)."""
    import gc
    import torch
    from meshdiffusion_amd import hip_ops as ops
    from meshdiffusion_amd.lib.diffusion.models import layers

    class L(layers.HipLayer):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(3))

    ops.prewarm_packs()                       # start from an empty registry
    calls = []
    lay = L()
    build = lambda: calls.append(1) or "obj"  # noqa: E731
    assert lay._cached("a", [lay.w], build) == "obj" and len(calls) == 1
    assert lay._cached("a", [lay.w], build) == "obj" and len(calls) == 1          # cache hit
    ops.prewarm_packs()
    assert len(calls) == 1                    # nothing changed: nothing rebuilt
    lay._cached("a", [lay.w], build)
    ops.bump_param_epoch()                    # what the fused optimizer does after its raw-pointer update
    ops.prewarm_packs()
    assert len(calls) == 2 and len(ops._PACK_LAYERS) == 0
    lay._cached("a", [lay.w], build)          # already rebuilt: a hit, and registered again
    assert len(calls) == 2 and len(ops._PACK_LAYERS) == 1
    del lay
    gc.collect()
    assert len(ops._PACK_LAYERS) == 0


def test_prewarm_requests_only_consumed_packs_and_failed_flush_resets_owners(monkeypatch):
    """(a) prewarm_packs re-queues only cache entries whose tiles were read in the previous step (a layer on
    the Winograd path looks its direct-tile entry up for rows / kdim only); (b) a flush that raises midway leaves no owner
    marked queued-without-a-job, and `.data` raises instead of handing out None."""
    import torch
    from meshdiffusion_amd import _lib, hip_ops as ops
    from meshdiffusion_amd.lib.diffusion.models import layers

    class Pack:                                # the protocol of PackedWeight / WinoWeight without a device
        made = []

        def __init__(self):
            self._data, self._queued, self.requests = None, False, 0
            Pack.made.append(self)

        def request(self):
            if self._data is None and not self._queued:
                self._queued = True
                self.requests += 1
                ops._PACK_QUEUE.append((self, dict(nbytes=32, w=torch.zeros(1))))

    class L(layers.HipLayer):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(3))

    ops.prewarm_packs()
    ops._PACK_QUEUE.clear()
    done = []
    monkeypatch.setattr(ops, "_flush", lambda queue: done.extend(queue) or [setattr(o, "_data", "tiles") or setattr(o, "_queued", False)
                                                                            for o, _ in queue])
    lay = L()
    looked_up = lay._cached("direct", [lay.w], Pack)      # consulted for its shape only: never read
    read = lay._cached("wino", [lay.w], Pack)
    read.request(); ops.flush_packs()
    assert read._data == "tiles" and looked_up._data is None
    ops.bump_param_epoch()
    ops.prewarm_packs()
    new_direct, new_wino = lay._cached("direct", [lay.w], Pack), lay._cached("wino", [lay.w], Pack)
    assert new_direct is not looked_up and new_wino is not read          # both rebuilt for the new weights ...
    assert new_wino._data == "tiles" and new_direct.requests == 0 and new_direct._data is None   # ... only the consumed one packed

    def boom(queue):
        raise RuntimeError("out of memory")
    monkeypatch.setattr(ops, "_flush", boom)
    ops._PACK_QUEUE.clear()
    a, b = Pack(), Pack()
    a.request(); b.request()
    with pytest.raises(RuntimeError):
        ops.flush_packs()
    assert not a._queued and not b._queued and ops._PACK_QUEUE == []    # the next access retries instead of returning None
    a.request()
    assert a._queued and len(ops._PACK_QUEUE) == 1
    ops._PACK_QUEUE.clear()
    # the real classes raise when a flush produced nothing
    pw = ops.PackedWeight.__new__(ops.PackedWeight)
    pw._data, pw._queued = None, True
    monkeypatch.setattr(ops, "flush_packs", lambda: None)
    with pytest.raises(_lib.MeshDiffusionHipError):
        pw.data


def test_upsample_onto_4cube_grid_stays_off_the_fused_generic_loader():
    """the generic tile's fused fp32 loader has no nearest-x2 fold; fused_operand_ok(pw, ups=1) must say no for
    CFG_C3_LOW (a model deep enough to upsample 2^3 -> 4^3) while the plain conv on a 4^3 grid keeps the fused loader."""
    from types import SimpleNamespace
    from meshdiffusion_amd import hip_ops as ops
    from meshdiffusion_amd.lib.diffusion.models import layers
    low = SimpleNamespace(cfg=ops.CFG_C3_LOW, prec=ops.PREC_BF16X3)
    fast = SimpleNamespace(cfg=ops.CFG_C3_128_FAST, prec=ops.PREC_BF16X3)
    assert ops.conv_cfg_for(4) == ops.CFG_C3_LOW
    if ops.FUSE_GN_APPLY:
        assert layers.fused_operand_ok(low) and not layers.fused_operand_ok(low, ups=1)
        assert layers.fused_operand_ok(fast) and layers.fused_operand_ok(fast, ups=1)


def test_hip_path_refuses_cpu_tensors():
    from meshdiffusion_amd import _lib, hip_ops
    with pytest.raises(_lib.MeshDiffusionHipError):
        hip_ops.ncdhw_to_f32b(torch.zeros(1, 8, 2, 2, 2))
    from meshdiffusion_amd import synth
    from meshdiffusion_amd.lib.diffusion.models import ddpm_res64, utils as mutils  # noqa: F401
    cfg = synth.small_config(); cfg.device = torch.device("cpu")
    model = mutils.create_model(cfg, use_parallel=False).eval()
    with pytest.raises(RuntimeError):
        with torch.no_grad():
            model(torch.zeros(1, 4, 16, 16, 16), torch.zeros(1))


# ---- host-side mirror of the reference API ---------------------------------------------------------------
def test_registry_state_dict_and_param_count():
    from meshdiffusion_amd.config import get_config_res64
    from meshdiffusion_amd.lib.diffusion.models import ddpm_res64, utils as mutils  # noqa: F401
    cfg = get_config_res64(); cfg.device = torch.device("cpu")
    m = mutils.get_model("ddpm_res64")(cfg)
    sd = m.state_dict()
    assert len(sd) == 497
    assert sum(p.numel() for p in m.parameters()) == 365076484   # 365.08 M
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 365076484 - 4 * 64 ** 3
    for k in ("all_modules.0.weight", "all_modules.3.GroupNorm_0.weight", "all_modules.3.Conv_0.weight",
              "all_modules.3.Dense_0.weight", "all_modules.11.NIN_0.W", "all_modules.12.NIN_3.W", "pos_layer.weight", "mask_layer.bias",
              "coords", "mask", "sigmas"):
        assert k in sd, k
    assert tuple(sd["all_modules.3.Conv_0.weight"].shape) == (128, 128, 3, 3, 3)
    with pytest.raises(ValueError):
        mutils.register_model(name="ddpm_res64")(type("X", (), {}))
    from meshdiffusion_amd import synth
    small = synth.small_config(); small.device = torch.device("cpu")
    wrapped = mutils.create_model(small)
    assert all(k.startswith("module.") for k in wrapped.state_dict())


def test_config_shim_and_cli_overrides(tmp_path):
    from meshdiffusion_amd import config as mdc
    c = mdc.get_config_res64()
    rest = mdc.apply_overrides(c, ["--config.eval.batch_size=8", "--config.eval.ckpt_path", "a/b.pth", "--x"])
    assert c.eval.batch_size == 8 and c.eval.ckpt_path == "a/b.pth" and rest == ["--x"]
    assert c.model.ch_mult == (1, 1, 2, 4, 4) and c.model.num_scales == 1000 and c.optim.lr == 2e-5
    # a reference-style config file that imports ml_collections loads through the shim
    d = tmp_path / "configs"; d.mkdir()
    (d / "default_configs.py").write_text(
        "import ml_collections\n\ndef get_default_configs():\n    c = ml_collections.ConfigDict()\n"
        "    c.model = ml_collections.ConfigDict()\n    c.model.nf = 64\n    return c\n")
    (d / "mine.py").write_text("from configs.default_configs import get_default_configs\n\n"
                               "def get_config():\n    c = get_default_configs()\n    c.model.name = 'ddpm_res64'\n    return c\n")
    cfg = mdc.load_config_file(str(d / "mine.py"))
    assert cfg.model.nf == 64 and cfg.model.name == "ddpm_res64"
    sys.path.insert(0, ROOT)
    import main_diffusion
    cfg2, mode = main_diffusion.parse(["--config=res64", "--mode", "uncond_gen", "--config.eval.batch_size=2"])
    assert mode == "uncond_gen" and cfg2.eval.batch_size == 2
    with pytest.raises(SystemExit):
        main_diffusion.parse(["--config=res64", "--mode=bogus"])


def test_vpsde_tables_and_stepper_tables_match_oracle():
    from meshdiffusion_amd.lib.diffusion import sampling, sde_lib
    from oracle import unet_oracle as uo
    sde = sde_lib.VPSDE(0.1, 20.0, 1000, device="cpu")
    betas, sqac, sq1m = uo.vpsde_tables()
    assert torch.equal(sde.discrete_betas, betas) and torch.equal(sde.sqrt_1m_alphas_cumprod, sq1m)
    st = sampling.AncestralStepper(sde, (2, 4, 8, 8, 8), device="cpu", grid_mask=torch.ones(1, 8, 8, 8))
    # fractional labels 999.0, 998.001, ... ; .long() truncation visits each integer 999..0 once
    assert float(st.labels[0, 0]) == 999.0 and abs(float(st.labels[1, 0]) - 998.001) < 1e-3
    ks = (st.timesteps * 999).long()
    assert sorted(ks.tolist()) == list(range(1000))
    assert torch.equal(st.coef[5, 0, 0], betas[ks[5]]) and torch.equal(st.coef[5, 1, 1], sq1m[ks[5]])
    with pytest.raises(AssertionError):
        sampling.AncestralStepper(sde, (1, 4, 8, 8, 8), device="cpu", grid_mask=torch.full((1, 8, 8, 8), 0.5))


def test_ema_and_checkpoint_roundtrip(tmp_path):
    from meshdiffusion_amd import synth
    from meshdiffusion_amd.lib.diffusion import losses
    from meshdiffusion_amd.lib.diffusion.models import ddpm_res64, utils as mutils  # noqa: F401
    from meshdiffusion_amd.lib.diffusion.models.ema import ExponentialMovingAverage
    from meshdiffusion_amd.lib.diffusion.utils import restore_checkpoint, save_checkpoint
    cfg = synth.small_config(); cfg.device = torch.device("cpu")
    m = mutils.create_model(cfg)
    ema = ExponentialMovingAverage(m.parameters(), decay=0.9999)
    before = [s.clone() for s in ema.shadow_params[:3]]
    with torch.no_grad():
        for p in m.parameters():
            if p.requires_grad:
                p.add_(1.0)
    ema.update(m.parameters())      # decay = min(0.9999, 2/11)
    d = 2.0 / 11.0
    assert torch.allclose(ema.shadow_params[0], before[0] + (1 - d) * 1.0, atol=1e-6)
    opt = losses.get_optimizer(cfg, m.parameters())
    state = dict(optimizer=opt, model=m, ema=ema, step=7)
    path = str(tmp_path / "ck" / "checkpoint.pth"); os.makedirs(os.path.dirname(path))
    save_checkpoint(path, state)
    m2 = mutils.create_model(cfg)
    st2 = dict(optimizer=losses.get_optimizer(cfg, m2.parameters()), model=m2,
               ema=ExponentialMovingAverage(m2.parameters(), decay=0.5), step=0)
    st2 = restore_checkpoint(path, st2, device="cpu")
    assert st2["step"] == 7 and st2["ema"].decay == 0.9999
    assert torch.equal(m2.state_dict()["module.all_modules.2.weight"], m.state_dict()["module.all_modules.2.weight"])


def test_dataset_mirror_matches_reference_golden(tmp_path):
    """Host mirror of lib/dataset/shapenet_dmtet_dataset.py vs items produced by the imported reference
    (oracle/gen_golden.py gen_dataset): bit-exact, including the sign(0)->+1 rule on the first depth slab,
    the per-channel jitter drawn from the global CPU RNG, the r<R mask crop and the high-end zero padding."""
    from oracle.gen_golden import dataset_inputs
    from meshdiffusion_amd.lib.dataset.shapenet_dmtet_dataset import ShapeNetDMTetDataset
    gold = np.load(os.path.join(GOLD, "dataset.npz"))
    meta, keep, mask = dataset_inputs(str(tmp_path))
    for tag, kw in (("aug_norm", dict(aug=True, normalize_sdf=True, filter_meta_path=keep)),
                    ("plain", dict(aug=False, normalize_sdf=False, filter_meta_path=None))):
        ds = ShapeNetDMTetDataset(meta, grid_mask=mask, extension="pt", **kw)
        assert len(ds) == int(gold[f"{tag}_len"])
        torch.manual_seed(4321)
        for i in range(len(ds)):
            item = ds[i]
            assert tuple(item.shape) == (4, 8, 8, 8)
            assert np.array_equal(item.numpy(), gold[f"{tag}_{i}"]), (tag, i)
    # .npy storage (the reference's 'npy' branch cannot run: it never imports numpy) gives the same items
    import json
    paths = json.load(open(meta))
    npy = []
    for p in paths:
        q = p[:-3] + ".npy"
        np.save(q, torch.load(p).numpy())
        npy.append(q)
    meta2 = str(tmp_path / "meta_npy.json")
    json.dump(npy, open(meta2, "w"))
    ds = ShapeNetDMTetDataset(meta2, grid_mask=mask, extension="npy", aug=False, normalize_sdf=False)
    assert np.array_equal(ds[1].numpy(), gold["plain_1"])


def test_rank_shard_sampler_partitions_every_epoch():
    from meshdiffusion_amd.lib.diffusion.trainer import RankShardSampler
    world, n = 3, 20
    samplers = [RankShardSampler(n, r, world, seed=5) for r in range(world)]
    for epoch in range(2):
        parts = [list(iter(s)) for s in samplers]
        assert all(len(p) == n // world for p in parts)
        flat = sum(parts, [])
        assert len(set(flat)) == len(flat) and set(flat) <= set(range(n))
        if epoch == 0:
            first = flat
    assert flat != first


def test_graft_entry_build_runs_without_a_gpu():
    """The driver's build check: compiles every HIP source for gfx950, loads the library, imports the host package."""
    sys.path.insert(0, ROOT)
    import __graft_entry__
    path = __graft_entry__.build()
    assert os.path.exists(path) and path.endswith("libmeshdiffusion_hip.so")


def test_obj_round_trip_and_tet_to_grid(tmp_path):
    """Host data formats around the hot path: OBJ writer (eval.py:436-440) and the dmt dict -> grid scatter
    (data/tets_to_3dgrid.py:7-15), which is the inverse of the gather the mesher / cond_gen do."""
    from meshdiffusion_amd import mesh_export
    from meshdiffusion_amd.dmtet import tet_vertices_to_grid_index
    g = torch.Generator().manual_seed(3)
    verts = torch.randn((17, 3), generator=g)
    faces = torch.randint(0, 17, (25, 3), generator=g)
    p = tmp_path / "m.obj"
    mesh_export.save_obj(p, verts, faces)
    v2, f2 = mesh_export.load_obj(p)
    assert np.array_equal(f2, faces.numpy()) and np.abs(v2 - verts.numpy()).max() < 1e-6
    assert open(p).read().splitlines()[0].startswith("v ") and "f 0 " not in open(p).read()   # 1-based faces
    tet = np.load(os.path.join(GOLD, "64_tets_cropped.npz"))
    idx = tet_vertices_to_grid_index(tet["vertices"])
    n = idx.shape[0]
    sdf, deform = torch.randn(n, generator=g), torch.rand((n, 3), generator=g) - 0.5
    grid = mesh_export.tet_to_grid(idx, sdf.unsqueeze(-1), deform, 64)
    assert tuple(grid.shape) == (4, 64, 64, 64)
    assert torch.equal(grid[0, idx[:, 0], idx[:, 1], idx[:, 2]], sdf)
    assert torch.equal(grid[1:, idx[:, 0], idx[:, 1], idx[:, 2]].transpose(0, 1), deform)
    assert int((grid[0] != 0).sum()) == n                     # nothing outside the tet-grid vertices
    d = tmp_path / "dicts"; d.mkdir()
    torch.save({"sdf": sdf, "deform": deform}, d / "dmt_dict_00003.pt")
    out = mesh_export.dicts_to_grids(tet["vertices"], str(d), str(tmp_path / "grids"), 64, range(5))
    assert len(out) == 1 and torch.equal(torch.load(out[0]), grid)


def test_plain_c_host_links_and_runs(tmp_path, hip_lib):
    """"a C/C++ host links -lmeshdiffusion_hip" -- compile tests/c_abi/host.c as C99 against the header,
    link it to the built library and run it (argument validation only, no GPU needed)."""
    import shutil
    from meshdiffusion_amd import _lib
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no C compiler")
    exe = str(tmp_path / "host")
    libdir = os.path.dirname(_lib.LIB_PATH)
    cmd = [gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c_abi", "host.c"), "-o", exe, "-L", libdir, "-lmeshdiffusion_hip",
           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and "ok" in run.stdout, run.stdout + run.stderr
    assert f"sizeof(MdGemmConvArgs)={ctypes.sizeof(_lib.MdGemmConvArgs)}" in run.stdout


def test_ddim_oracle_and_schedule_vs_reference_golden():
    """oracle ddim_step == the unmodified reference's discretize_ddim (float64, bit for bit) on the recorded inputs,
    and the host sampler's 100-point quadratic schedule == sampling.py:548-557."""
    from meshdiffusion_amd.lib.diffusion import sampling
    from oracle import unet_oracle as uo
    gold = np.load(os.path.join(GOLD, "ddim.npz"))
    ts = sampling.ddim_schedule(1000)
    assert len(ts) == 100 and np.array_equal((ts * 1000).round().long().numpy(), gold["seq"])
    assert np.array_equal(sampling.ddim_schedule(1000, "uniform", 100).numpy(), (np.arange(0, 1000, 10) / 1000).astype(np.float32))
    g = torch.Generator().manual_seed(int(gold["step_seed"]))
    x = torch.randn((2, 4, 4, 4, 4), generator=g)
    eps = torch.randn((2, 4, 4, 4, 4), generator=g)
    for n in range(3):
        i = int(gold[f"step{n}_i"])
        xin = x if n == 0 else x.double() * 0.7
        xn, x0p = uo.ddim_step(xin, eps, torch.ones(2) * ts[i], torch.ones(2) * ts[i - 1])
        assert np.array_equal(xn.numpy(), gold[f"step{n}_x_new"]) and np.array_equal(x0p.numpy(), gold[f"step{n}_x0_pred"])


def test_round4_entry_points_reject_bad_arguments_without_a_gpu(hip_lib):
    """The f16f8 / f16f6 entry points (md_wino_prep_f8 / _f6, md_wino_pack_weights_f8 / _f6, md_conv3_wino_f8 / _f6) check their
    arguments before any launch; the f6 operand pass needs whole 16-channel blocks per part; precision names map to the
    cross-term formats on the host."""
    from meshdiffusion_amd import hip_ops
    buf = ctypes.create_string_buffer(64)
    p = ctypes.cast(buf, ctypes.c_void_p)
    for prep in (hip_lib.md_wino_prep_f8, hip_lib.md_wino_prep_f6):
        assert prep(None, None, 16, 0, None, 0, 0, None, p, 1, 8, 8, 8, None) == -1            # no input
        assert prep(p, None, 16, 0, None, 1, 0, None, p, 1, 8, 8, 8, None) == -1               # SiLU without the folded affine
        assert prep(p, None, 16, 0, None, 0, 0, None, p, 1, 8, 8, 7, None) == -1               # odd W
        assert prep(p, None, 16, 0, None, 0, 0, None, p, 1, 8, 8, 24, None) == -2              # W does not divide 256
    assert hip_lib.md_wino_prep_f6(p, None, 24, 0, None, 0, 0, None, p, 1, 8, 8, 8, None) == -1        # 24 channels: not whole K blocks
    assert hip_lib.md_wino_prep_f6(p, p, 32, 8, None, 0, 0, None, p, 1, 8, 8, 8, None) == -1           # second part of 8 channels
    assert hip_lib.md_wino_equaliser(None, p, p, 128, 64, 64 * 27, 27, p, None) == -1 and hip_lib.md_wino_equaliser(p, p, p, 128, 0, 0, 27, p, None) == -1
    assert hip_lib.md_wino_weight_bytes_f8(128, 64) == 128 * 64 * 36 * 4 + 256 and hip_lib.md_wino_weight_bytes_f8(128, 48) < 0
    for pack in (hip_lib.md_wino_pack_weights_f8, hip_lib.md_wino_pack_weights_f6):
        assert pack(None, None, p, 128, 64, 64 * 27, 27, None) == -1 and pack(p, None, p, 96, 64, 64 * 27, 27, None) == -1
    for conv in (hip_lib.md_conv3_wino_f8, hip_lib.md_conv3_wino_f6):
        assert conv(None, p, p, None, 0, None, 0, None, 1, 32, 128, 8, 8, 8, None) == -1
        assert conv(p, p, p, None, 0, None, 0, None, 1, 48, 128, 8, 8, 8, None) == -2    # cin % 32
        assert conv(p, p, p, None, 0, None, 0, None, 1, 32, 128, 6, 8, 8, None) == -2    # D % 4
    assert not hip_ops._SCOPES
    before = (hip_ops.PRECISION, hip_ops.WINO_F8, hip_ops.DEFAULT_PRECISION)
    for mode, fmt in (("bf16x3", False), ("f16f8", "f8"), ("f16f6", "f6")):
        if hip_ops.FORCE_PRECISION:
            break
        with hip_ops.precision_scope(mode):
            assert hip_ops.precision_name() == mode and hip_ops.WINO_F8 == fmt
            assert hip_ops.wino_f8_ok(64) == fmt and hip_ops.wino_f8_ok(64, drop=(0.1, 1)) is False and hip_ops.wino_f8_ok(24) is False
            # an operand that does not come out of a GroupNorm (Upsample: the raw residual stream) keeps bf16x3
            assert hip_ops.wino_f8_ok(64, normalised=False) is False
            with hip_ops.precision_scope(None, training=True):          # a training forward / backward inside: always bf16x3
                assert hip_ops.precision_name() == "bf16x3" and hip_ops.wino_f8_ok(64) is False
            assert hip_ops.precision_name() == mode
    if not hip_ops.FORCE_PRECISION:
        with hip_ops.precision_scope("f16f6"):
            assert hip_ops.wino_f8_ok(64, parts=[(None, 24), (None, 8)]) == "f8"       # parts that are not whole 16-channel blocks fall back
        try:
            with hip_ops.precision_scope("f16f8"):
                raise KeyError("boom")
        except KeyError:
            pass
    # a scope never leaks: the process default is back, whatever happened inside
    assert (hip_ops.PRECISION, hip_ops.WINO_F8, hip_ops.DEFAULT_PRECISION) == before and not hip_ops._SCOPES


def test_trained_like_state_dict_is_deterministic_and_compensated():
    """synth.trained_like_state_dict (the adversarial weights of the round-5 parity gate): same keys / shapes as the template, bit-identical
    between calls, heavy-tailed conv weights, and every rescaled GroupNorm is compensated in its consumer: gamma_c * rms(W[:, c]) keeps
    the spread of the un-rescaled pair (every channel keeps mattering) while gamma itself spans ~2^6."""
    import torch
    from meshdiffusion_amd import synth
    from meshdiffusion_amd.lib.diffusion.models import ddpm_res64, utils as mutils  # noqa: F401
    cfg = synth.small_config(); cfg.device = torch.device("cpu")
    tmpl = mutils.create_model(cfg, use_parallel=False).state_dict()
    a = synth.trained_like_state_dict(tmpl, grid_mask=synth.synthetic_grid_mask(16))
    b = synth.trained_like_state_dict(tmpl, grid_mask=synth.synthetic_grid_mask(16))
    assert list(a) == list(tmpl) and all(a[k].shape == tmpl[k].shape and torch.equal(a[k], b[k]) for k in a)
    gk = [k for k in a if k.endswith("GroupNorm_0.weight") and k.replace("GroupNorm_0.weight", "Conv_0.weight") in a]
    assert len(gk) >= 5
    for k in gk:
        g, w = a[k].abs(), a[k.replace("GroupNorm_0.weight", "Conv_0.weight")]
        assert float(g.max() / g.min()) > 8.0                                  # the GroupNorm scale really spreads
        imp = g * w.pow(2).mean(dim=(0, 2, 3, 4)).sqrt()
        assert float(imp.max() / imp.min()) < 8.0                               # ... and the consumer undoes it (up to gamma's own +-15 % and the weights' tails)
    w = a[gk[0].replace("GroupNorm_0.weight", "Conv_0.weight")]
    kurt = float(((w - w.mean()) ** 4).mean() / w.var() ** 2)
    assert kurt > 4.0                                                           # Student-t tails (a uniform draw has 1.8, a normal one 3)
