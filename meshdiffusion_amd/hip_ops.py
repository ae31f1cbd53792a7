"""Host-side wrappers of the C ABI: torch tensors provide device memory and the stream,
every computation happens in libmeshdiffusion_hip.so.  No fallback paths.

Blocked device layouts: F32B float32 [B][C/8][P][8]; S16B bf16 [B][C/8][2][P][8].
"""
import ctypes as C
import math
import os

import torch

from . import _lib
from ._lib import (A_PACKED, A_S16B, PREC_BF16X3, PREC_FP16X2, CFG_C3_128, CFG_C3_128_V2, CFG_C3_128_SW, CFG_C3_128_PIPE, CFG_C3_128_FAST, CFG_C5_128_K16, CFG_C5_32_K16, CFG_C3_128_W4, CFG_C3X_32, CFG_C5X_32_K16, CFG_C3X_128_K16, CFG_C5X_128, CFG_G1_128_N128, CFG_C3_128_K16, CFG_C3_32, CFG_C3_LOW, CFG_C3_S2,
                   CFG_G1_128, CFG_G1_128_LOW, CFG_G1_64_LOW, CFG_NT_KC, OUT_F32B, OUT_NCDHW, OUT_S16B,
                   MdGemmConvArgs, check)


CFG_ABL1, CFG_ABL2, CFG_ABL3, CFG_ABL4, CFG_ABL5 = 101, 102, 103, 104, 105
CFG_F1, CFG_F3, CFG_F4, CFG_F6, CFG_F7, CFG_F8 = 111, 113, 114, 116, 117, 118   # timing-only ablations of the dedicated kernel (MD_BUILD_ABLATIONS=1)
CFG_FAST_EC = 122   # A/B variant of the dedicated kernel (valid results): early weight commit
# shader cycles over which the dedicated conv kernel spreads the start of each CU's first workgroup (0 = off)
CONV_STAGGER = int(os.environ.get("MD_CONV_STAGGER", "0"))
DEBUG_ACT_FP16 = 2 if os.environ.get("MD_DEBUG_ACT_FP16") == "1" else 0   # precision experiment only
# Arithmetic of the 3x3x3 conv kernels:
#   "bf16x3"  three bf16 MFMAs per product, fp32 range: ~1.7e-5 per U-Net evaluation.  Training always runs in it.
#   "f16f8"   bf16x3 everywhere except the INFERENCE Winograd convs behind a GroupNorm (md_wino_prep_f8 + md_conv3_wino_f8): a product
#             there is fp16(a) fp16(b) + [e4m3(a) e4m3(b_lo 2^11) + e4m3(a_lo 2^11) e4m3(b)] 2^-11 -- one fp16 MFMA + half a
#             K-concatenated scaled fp8 MFMA = 2 matrix-core units per product instead of 3 (1.3e-5 per conv against 5.5e-6).
#   "f16f6"   as f16f8, the cross terms from MX block-scaled e2m3 images (md_wino_prep_f6 + md_conv3_wino_f6): the K = 64 MFMA runs
#             them at twice its e4m3 rate (1.8e-5 per conv).
#   "fp16x2"  weights split fp16, activations one fp16 in the direct kernels (~1e-3 per evaluation): opt-in experiment.
# f16f8 / f16f6 apply only where the operand comes out of GroupNorm (+ SiLU): there the static equaliser (md_wino_equaliser, WINO_EQ)
# flattens the per-channel magnitudes the 4-bit-significand images are sensitive to.  Convs on the raw residual stream (Upsample)
# keep bf16x3: nothing is known about their operand before the data arrives (F8_RAW=1: the round-4 behaviour, A/B only).
#
# The arithmetic is a property of the MODEL CALL, not of the process: DDPMUNet3D.forward / forward_train / backward enter
# `precision_scope(config.model.hip_precision)` and leave it again, so a model never inherits what another one ran in.  Outside any
# scope (direct hip_ops calls from tests and tools) DEFAULT_PRECISION applies (MD_PRECISION, or set_precision()).
_MODES = ("bf16x3", "fp16x2", "f16f8", "f16f6")
DEFAULT_PRECISION = os.environ.get("MD_PRECISION", "bf16x3")
FORCE_PRECISION = os.environ.get("MD_FORCE_PRECISION")      # A/B runs of the test suite / CLI: wins over config.model.hip_precision in INFERENCE scopes
WINO_EQ = os.environ.get("MD_WINO_EQ", "1") == "1"          # A/B: 0 = f16f8 / f16f6 without the static equaliser
F8_RAW = os.environ.get("MD_F8_RAW", "0") == "1"            # A/B: 1 = f16f8 / f16f6 also on un-normalised operands (round 4)


def _decode(mode):
    if mode not in _MODES:
        raise ValueError(f"unknown precision mode {mode!r}")
    return ("bf16x3", mode[3:]) if mode in ("f16f8", "f16f6") else (mode, False)


# the CURRENT arithmetic: PRECISION = operand format of the direct kernels, WINO_F8 = False | "f8" | "f6" (cross-term format of the
# inference Winograd convs).  Written only by set_precision (process default) and precision_scope (a model call).
PRECISION, WINO_F8 = _decode(DEFAULT_PRECISION)
TRAINING_SCOPE = False          # inside precision_scope(training=True): a training forward / backward of a model
_SCOPES = []


def set_precision(mode):
    """Process default: the arithmetic of hip_ops calls made OUTSIDE a model's forward (tests, tools).  Models state their own
    (config.model.hip_precision) and are not affected."""
    global DEFAULT_PRECISION, PRECISION, WINO_F8
    _decode(mode)
    DEFAULT_PRECISION = mode
    if not _SCOPES:
        PRECISION, WINO_F8 = _decode(mode)


class precision_scope:
    """with precision_scope(mode): the arithmetic of one model call.  mode None = the process default.  training=True: the scope of
    a training forward / backward -- always bf16x3 (the backward reads the forward's bf16 T, gradients need bf16's exponent range),
    MD_FORCE_PRECISION does not apply."""

    def __init__(self, mode, training=False):
        if training:
            mode = "bf16x3"
        else:
            mode = FORCE_PRECISION or mode or DEFAULT_PRECISION
        self.state = _decode(mode) + (bool(training),)

    def __enter__(self):
        global PRECISION, WINO_F8, TRAINING_SCOPE
        _SCOPES.append((PRECISION, WINO_F8, TRAINING_SCOPE))
        PRECISION, WINO_F8, TRAINING_SCOPE = self.state
        return self

    def __exit__(self, *exc):
        global PRECISION, WINO_F8, TRAINING_SCOPE
        PRECISION, WINO_F8, TRAINING_SCOPE = _SCOPES.pop()
        return False


def precision_name():
    return ("f16" + WINO_F8) if (WINO_F8 and PRECISION == "bf16x3") else PRECISION


def fast_prec(cfg):
    """Operand format to use for a conv that will run on configuration `cfg`."""
    return PREC_FP16X2 if (PRECISION == "fp16x2" and cfg == CFG_C3_128_FAST) else PREC_BF16X3


PARAM_EPOCH = 0   # bumped by optimisers that update parameters through raw pointers (no tensor _version bump)


def bump_param_epoch():
    global PARAM_EPOCH
    PARAM_EPOCH += 1


# A/B (MD_NIN_SIDE_STREAM=1): ResnetBlock shortcut GEMM on a second HIP stream next to Conv_0.  Measured neutral on
# MI355X (62.3 vs 62.8 sample-steps/s): a 125 KB-LDS conv workgroup and the GEMM never share a CU, so the two
# kernels only trade CUs.  Off by default.
NIN_SIDE_STREAM = os.environ.get("MD_NIN_SIDE_STREAM", "0") == "1"
_SIDE = {}


def side_stream():
    dev = torch.cuda.current_device()
    if dev not in _SIDE:
        _SIDE[dev] = torch.cuda.Stream(device=dev)
    return _SIDE[dev]


PROFILE = None   # set to a list by bench.py to collect (cfg, flops, start_event, end_event) per GEMM/conv launch
AUDIT = None     # set to a list by tools/audit_precision.py: every f16f8 / f16f6 conv launch is repeated in bf16x3 and the difference recorded


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _require_cuda(t, name):
    if not t.is_cuda:
        raise _lib.MeshDiffusionHipError(f"{name} must live on the GPU: the HIP path has no CPU fallback")


def f32b_empty(B, Cc, P, device):
    return torch.empty((B, Cc // 8, P, 8), dtype=torch.float32, device=device)


def s16b_empty(B, Cc, P, device):
    return torch.empty((B, Cc // 8, 2, P, 8), dtype=torch.bfloat16, device=device)


# ---------------------------------------------------------------------------------------------
# weights
# ---------------------------------------------------------------------------------------------
class PackedWeight:
    """Split-bf16 WPK tiles of one conv / NIN weight (built on the device by md_pack_weights).  The tiles are packed on the
    first access of `.data`: a layer whose convolution runs through the Winograd path never packs its direct tiles (in
    training every weight version would otherwise be packed twice per step)."""

    def __init__(self, w, kind, cfg, device, prec=PREC_BF16X3):
        self.prec = prec
        nt, kc = CFG_NT_KC[cfg]
        w = w.detach().to(device=device, dtype=torch.float32).contiguous()
        _require_cuda(w, "weight")
        base_off = 0
        if kind == "conv":  # [Co][Ci][k][k][k]
            rows, kdim, taps = w.shape[0], w.shape[1], w.shape[2] * w.shape[3] * w.shape[4]
            s_row, s_k, s_tap = kdim * taps, taps, 1
        elif kind == "conv_dgrad":  # the data-gradient conv of [Co][Ci][k][k][k]: W'[ci][co][t] = W[co][ci][T-1-t],
            # read in place through strides (tap stride -1 from the last tap): no flipped / transposed copy
            taps = w.shape[2] * w.shape[3] * w.shape[4]
            rows, kdim = w.shape[1], w.shape[0]
            s_row, s_k, s_tap = taps, rows * taps, -1
            base_off = taps - 1
        elif kind == "nin":  # NIN W [Ci][Co]
            rows, kdim, taps = w.shape[1], w.shape[0], 1
            s_row, s_k, s_tap = 1, rows, 0
        elif kind == "rows":  # plain [rows][K]
            rows, kdim, taps = w.shape[0], w.shape[1], 1
            s_row, s_k, s_tap = kdim, 1, 0
        else:
            raise ValueError(kind)
        self.rows, self.taps, self.cfg = rows, taps, cfg
        self.kdim = ((kdim + kc - 1) // kc) * kc  # K padded to the chunk size (zero filled)
        self._src = (w, rows, kdim, taps, s_row, s_k, s_tap, nt, kc, base_off)   # `w` is this weight version (cache key)
        self._data = None
        self._queued = False

    def request(self):
        """Queue the packing job (no launch): the next `.data` access of ANY queued weight, or flush_packs(), packs all of
        them in one md_pack_batch launch."""
        if self._data is None and not self._queued:
            lib = _lib.load()
            w, rows, kdim, taps, s_row, s_k, s_tap, nt, kc, base_off = self._src
            nbytes = lib.md_packed_weight_bytes(rows, kdim, taps, nt, kc)
            if nbytes <= 0:
                raise _lib.MeshDiffusionHipError("md_packed_weight_bytes failed")
            self._queued = True
            _PACK_QUEUE.append((self, dict(kind=_lib.PACK_WPK, w=w, w_off=4 * base_off, nbytes=nbytes, rows=rows, kdim=kdim, taps=taps,
                                           s_row=s_row, s_k=s_k, s_tap=s_tap, nt=nt, kc=kc, prec=self.prec, flip=0)))

    @property
    def data(self):
        if self._data is None:
            self.request()
            flush_packs()
            if self._data is None:
                raise _lib.MeshDiffusionHipError("packed weight requested but not produced by flush_packs()")
        return self._data


_PACK_QUEUE = []      # (owner, job) pairs waiting for flush_packs()
PACK_BATCH = os.environ.get("MD_PACK_BATCH", "1") == "1"   # A/B switch: 0 = one md_pack_weights / md_wino_pack_weights launch per weight
PACK_TILED = os.environ.get("MD_PACK_TILED", "1") == "1"   # A/B switch: 0 = the batch packs every weight one item per thread


def flush_packs():
    """Pack every queued weight: one md_pack_batch launch (PACK_BATCH) or one launch per weight."""
    global _PACK_QUEUE
    if not _PACK_QUEUE:
        return
    queue, _PACK_QUEUE = _PACK_QUEUE, []
    try:
        _flush(queue)
    except BaseException:
        # an allocation or a launch failed midway: no owner may stay "queued" without a job behind it (request() would be a
        # no-op and .data would hand out None from then on) -- they go back to "not packed" and the next access retries
        for owner, _ in queue:
            if owner._data is None:
                owner._queued = False
        raise


def _flush(queue):
    lib = _lib.load()
    outs = []
    for owner, j in queue:
        outs.append(torch.empty(j["nbytes"] // 2, dtype=torch.bfloat16, device=j["w"].device))
    if PACK_BATCH and len(queue) > 1 and len({j["w"].device for _, j in queue}) == 1:
        def tiled(j):      # block-cooperative form (md_pack_tiled_kernel): the 3x3x3 weights -- most of the parameters
            return (PACK_TILED and j["kind"] == _lib.PACK_WPK and j["taps"] > 1 and abs(j["s_tap"]) == 1
                    and 16 * j["kc"] * j["taps"] <= 13824 and j["nt"] % 16 == 0)

        def mode(j):       # md_pack_batch `tiled` argument for this job
            if j["kind"] == _lib.PACK_WINO_F6:
                return 0
            if PACK_TILED and j["kind"] == _lib.PACK_WINO:
                return 3
            if not tiled(j):
                return 0
            m = j["taps"] * 100 + j["kc"]
            return m if m in (2732, 2716, 932, 916) else 1

        for md in sorted({mode(j) for _, j in queue}, reverse=True):
            sel = [(j, out) for (owner, j), out in zip(queue, outs) if mode(j) == md]
            jobs = (_lib.MdPackJob * len(sel))()
            block0 = 0
            for i, (j, out) in enumerate(sel):
                J = jobs[i]
                J.w, J.out = j["w"].data_ptr() + j["w_off"], out.data_ptr()
                J.s_row, J.s_k, J.s_tap = j["s_row"], j["s_k"], j["s_tap"]
                J.n_items, J.block0 = j["nbytes"] // 16, block0
                J.rows, J.kdim, J.taps, J.nt, J.kc, J.prec, J.flip, J.kind = (j["rows"], j["kdim"], j["taps"], j["nt"], j["kc"],
                                                                             j["prec"], j["flip"], j["kind"])
                if md == 3:
                    block0 += (j["rows"] // 32) * (j["kdim"] // 16)
                elif md:
                    block0 += -(-j["rows"] // j["nt"]) * (j["nt"] // 16) * -(-j["kdim"] // j["kc"])
                else:
                    block0 += (J.n_items + 255) // 256
            table = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(outs[0].device)
            check(lib.md_pack_batch(_ptr(table), len(sel), block0, md, _stream()), "md_pack_batch")
    else:
        for (owner, j), out in zip(queue, outs):
            wp = C.c_void_p(j["w"].data_ptr() + j["w_off"])
            if j["kind"] == _lib.PACK_WPK:
                check(lib.md_pack_weights(wp, _ptr(out), j["rows"], j["kdim"], j["taps"], j["s_row"], j["s_k"], j["s_tap"], j["nt"],
                                          j["kc"], j["prec"], _stream()), "md_pack_weights")
            elif j["kind"] == _lib.PACK_WINO_F6:      # a table of one job (the fixed-pre-scale f16f6 form exists as a batch job only)
                jobs = (_lib.MdPackJob * 1)()
                J = jobs[0]
                J.w, J.out = j["w"].data_ptr() + j["w_off"], out.data_ptr()
                J.s_row, J.s_k, J.s_tap, J.n_items, J.block0 = j["s_row"], j["s_k"], 0, j["nbytes"] // 16, 0
                J.rows, J.kdim, J.taps, J.nt, J.kc, J.prec, J.flip, J.kind = j["rows"], j["kdim"], 9, 0, 0, j["prec"], j["flip"], j["kind"]
                table = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(out.device)
                check(lib.md_pack_batch(_ptr(table), 1, (J.n_items + 255) // 256, 0, _stream()), "md_pack_batch")
            else:
                check(lib.md_wino_pack_weights(wp, _ptr(out), j["rows"], j["kdim"], j["s_row"], j["s_k"], j["flip"], _stream()),
                      "md_wino_pack_weights")
    for (owner, j), out in zip(queue, outs):
        owner._data = out
        owner._src = None
        owner._queued = False


# Sites of the packed-weight caches (layers.HipLayer._cached) that were used since the last prewarm: a training step changes
# every weight, so the forward of the next step would rebuild each cache entry at its first use -- one packing launch per
# weight.  prewarm_packs() (start of a training forward) rebuilds the entries the previous step used, queues their packing
# jobs and flushes them as md_pack_batch launches.  The (params, builder) pairs live on the layers themselves (they reference
# the layer: a global table would keep every model ever built alive); here only a weak set of the layers that have some.
import weakref  # noqa: E402

_PACK_LAYERS = weakref.WeakSet()


def note_pack_use(layer, name, params, builder):
    used = layer.__dict__.get("_md_used")
    if used is None:
        used = layer.__dict__["_md_used"] = {}
        _PACK_LAYERS.add(layer)
    if name not in used:
        used[name] = (list(params), builder)


def prewarm_packs():
    layers = list(_PACK_LAYERS)
    _PACK_LAYERS.clear()
    n = 0
    for layer in layers:
        used = layer.__dict__.pop("_md_used", None) or {}
        if not PACK_BATCH:
            continue
        cache = layer.__dict__.get("_md_cache") or {}
        for name, (params, builder) in used.items():
            old = cache.get(name)
            # only what the previous step CONSUMED: a layer on the Winograd path looks its direct-tile entry up for
            # rows / kdim but never reads `.data` -- packing those tiles every step would cost ~4 B per parameter for nothing
            consumed = old is not None and getattr(old[1], "_data", None) is not None
            obj = layer._cached(name, params, builder, mark=False)
            if consumed and hasattr(obj, "request"):
                obj.request()
                n += 1
    flush_packs()
    return n


def pack_s16b_from_matrix(w_kp, device):
    """[K][P] fp32 matrix -> S16B [1][K/8][2][P][8] (used for a weight that plays the B operand)."""
    lib = _lib.load()
    w = w_kp.detach().to(device=device, dtype=torch.float32).contiguous()
    K, P = w.shape
    out = s16b_empty(1, K, P, device)
    check(lib.md_ncdhw_to_s16b(_ptr(w), _ptr(out), 1, K, K, P, _stream()), "md_ncdhw_to_s16b")
    return out


# ---------------------------------------------------------------------------------------------
# GEMM / conv
# ---------------------------------------------------------------------------------------------
def gemm_conv(*, cfg, a, b, out, batch, rows, rows_alloc, kdim, dims, bias=None, bias_bstride=0,
              residual=None, res_bstride=0, alpha=1.0, ups=0, a_src=A_PACKED, a_rows=0, a_bstride=0,
              b_bstride=None, out_mode=OUT_F32B, ksplit=1, prec=PREC_BF16X3, stats=None, stagger=None,
              b_f32=None):
    """stats: optional zeroed float64 [batch][rows_alloc][2] receiving per-(sample, channel) sum / sum of squares of
    the output (CFG_C3_128_FAST without split-K only).
    b_f32: dict(parts=[(F32B tensor, C), ...] (1 or 2), ac=[B][K][2] or None, silu=bool): the B operand is read as fp32
    and GroupNorm affine + SiLU + the bf16 split happen in the kernel's halo loader (MD_B_F32B_GN; `b` is ignored)."""
    lib = _lib.load()
    D, H, W = dims
    args = MdGemmConvArgs()
    args.a, args.out = a.data_ptr(), out.data_ptr()
    if b_f32 is not None:
        parts = b_f32["parts"]
        assert 1 <= len(parts) <= 2 and sum(c for _, c in parts) == kdim and all(c % 8 == 0 for _, c in parts)
        pin = D * H * W // (8 if ups else 1)
        args.b, args.b_split = parts[0][0].data_ptr(), parts[0][1]
        b_bstride = parts[0][1] * pin                       # floats between batches of part 1
        if len(parts) == 2:
            args.b2, args.b2_bstride = parts[1][0].data_ptr(), parts[1][1] * pin
        ac = b_f32.get("ac")
        args.b_ac = ac.data_ptr() if ac is not None else None
        args.b_silu = 1 if b_f32.get("silu") else 0
        args.b_mode = _lib.B_F32B_GN
    else:
        args.b = b.data_ptr()
    args.bias = bias.data_ptr() if bias is not None else None
    args.residual = residual.data_ptr() if residual is not None else None
    args.alpha = alpha
    args.cfg, args.batch, args.rows, args.rows_alloc, args.kdim = cfg, batch, rows, rows_alloc, kdim
    args.D, args.H, args.W, args.ups = D, H, W, ups
    args.a_src, args.out_mode, args.a_rows = a_src, out_mode, a_rows
    args.a_bstride, args.bias_bstride, args.res_bstride = a_bstride, bias_bstride, res_bstride
    if b_bstride is None:
        pin = D * H * W
        if ups:
            pin //= 8
        elif cfg == CFG_C3_S2:
            pin *= 8
        b_bstride = (kdim // 8) * 2 * pin * 8
    args.b_bstride = b_bstride
    args.prec = prec
    args.stats = stats.data_ptr() if stats is not None else None
    args.stagger = CONV_STAGGER if stagger is None else int(stagger)
    part = None
    if ksplit > 1:
        if out_mode != OUT_F32B:
            raise ValueError("split-K supports F32B output only")
        args.ksplit = ksplit
        part = torch.empty((ksplit * batch * rows_alloc * D * H * W,), dtype=torch.float32, device=out.device)
        args.partial = part.data_ptr()
    if PROFILE is None:
        check(lib.md_gemm_conv(C.byref(args), _stream()), f"md_gemm_conv(cfg={cfg})")
    else:  # bench.py: HIP events on the launch stream around this launch (algorithmic flops, 1x)
        taps = _lib.cfg_info(cfg)["taps"]
        flops = 2.0 * batch * rows * kdim * taps * D * H * W
        # algorithmic HBM bytes of this launch: operand in, weights, output (+ residual), each once
        pin = D * H * W // (8 if ups else 1) * (8 if cfg == CFG_C3_S2 else 1)
        abytes = 4.0 * (batch * kdim * pin + rows * kdim * taps + batch * rows * D * H * W * (2 if residual is not None else 1))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.md_gemm_conv(C.byref(args), _stream()), f"md_gemm_conv(cfg={cfg})")
        e1.record()
        PROFILE.append((cfg, flops, e0, e1, abytes,
                        f"{kdim}->{rows}@{D}x{H}x{W}" + ("/ups" if ups else "") + (f"/ks{ksplit}" if ksplit > 1 else "") +
                        ("/res" if residual is not None else "") + ("/stats" if stats is not None else "")))
    return out


# ---------------------------------------------------------------------------------------------
# Winograd F(2,3)-along-w path of the 3x3x3 convolution (inference)
# ---------------------------------------------------------------------------------------------
WINO = os.environ.get("MD_WINO", "1") == "1"   # A/B switch: md_wino_prep + md_conv3_wino instead of the fused direct kernel
WINO_PREP_V2 = os.environ.get("MD_WINO_PREP_V2", "1") == "1"   # two-phase operand pass (csrc/wino_prep2.hip): same bits as md_wino_prep, 13 % faster
WINO_TRAIN_FWD = os.environ.get("MD_WINO_TRAIN_FWD", "1") == "1"   # A/B switch: the same for the forward convs of a training step


class WinoWeight:
    """Conv3d weight [Co][Ci][3][3][3] -> G-transformed split-bf16 fragment tiles (md_wino_pack_weights).
    kind "conv": the forward convolution; "conv_dgrad": its data gradient W'[ci][co][t] = W[co][ci][26 - t], read in place."""

    def __init__(self, w, device, kind="conv"):
        lib = _lib.load()
        w = w.detach().to(device=device, dtype=torch.float32).contiguous()
        _require_cuda(w, "weight")
        assert w.dim() == 5 and tuple(w.shape[2:]) == (3, 3, 3)
        if kind == "conv":
            self.rows, self.kdim = w.shape[0], w.shape[1]
            s_row, s_k, flip = self.kdim * 27, 27, 0
        elif kind == "conv_dgrad":
            self.rows, self.kdim = w.shape[1], w.shape[0]
            s_row, s_k, flip = 27, self.rows * 27, 1
        else:
            raise ValueError(kind)
        nbytes = lib.md_wino_weight_bytes(self.rows, self.kdim)
        if nbytes <= 0:
            raise _lib.MeshDiffusionHipError("md_wino_weight_bytes: unsupported weight shape")
        self._data, self._queued = None, False
        self._src = dict(kind=_lib.PACK_WINO, w=w, w_off=0, nbytes=nbytes, rows=self.rows, kdim=self.kdim, taps=9, s_row=s_row,
                         s_k=s_k, s_tap=0, nt=0, kc=0, prec=PREC_BF16X3, flip=flip)

    def request(self):
        """Queue the packing job (see PackedWeight.request)."""
        if self._data is None and not self._queued:
            self._queued = True
            _PACK_QUEUE.append((self, self._src))

    @property
    def data(self):
        if self._data is None:
            self.request()
            flush_packs()
            if self._data is None:
                raise _lib.MeshDiffusionHipError("packed weight requested but not produced by flush_packs()")
        return self._data


class WinoWeightF8:
    """Conv3d weight [Co][Ci][3][3][3] -> the "f16f8" fragments of md_conv3_wino_f8 (md_wino_pack_weights_f8: fp16 hi fragments +
    K-concatenated e4m3 fragments of the power-of-two pre-scaled G-transformed weights, header with the scale).  Inference only."""

    def __init__(self, w, device, fmt="f8", eq=None):
        """eq: float [Cin] from wino_equaliser() or None; the fragments hold w[:, c] / eq[c] and the operand pass of every launch
        with this object must be given the same vector (conv3_wino checks it)."""
        lib = _lib.load()
        w = w.detach().to(device=device, dtype=torch.float32).contiguous()
        _require_cuda(w, "weight")
        assert w.dim() == 5 and tuple(w.shape[2:]) == (3, 3, 3) and fmt in ("f8", "f6")
        self.rows, self.kdim, self.fmt = w.shape[0], w.shape[1], fmt      # fmt "f6": md_wino_pack_weights_f6 (MX e2m3 cross terms)
        assert eq is None or (eq.is_cuda and eq.dtype == torch.float32 and eq.numel() == self.kdim and eq.is_contiguous())
        self.eq = eq
        nbytes = lib.md_wino_weight_bytes_f8(self.rows, self.kdim)
        if nbytes <= 0:
            raise _lib.MeshDiffusionHipError("md_wino_weight_bytes_f8: unsupported weight shape")
        self.data = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=device)
        pack = lib.md_wino_pack_weights_f6 if fmt == "f6" else lib.md_wino_pack_weights_f8
        check(pack(_ptr(w), _ptr(eq), _ptr(self.data), self.rows, self.kdim, self.kdim * 27, 27, _stream()), "md_wino_pack_weights_" + fmt)


# Training, round 6: the data-gradient convs of the layers on the Winograd path run in the f16f6 arithmetic (2/3 .. 3/4 of the bf16x3
# kernel's matrix-core energy: the kernel is power-bound) -- their operand is lifted by DGRAD_TSCALE (a power of
# two: gradient magnitudes into the fp16 plane's normal range, divided out again by the conv's launch), their weights are packed by the
# step's md_pack_batch table with a fixed pre-scale.  A/B switch: MD_DGRAD_F6=0 = bf16x3 data gradients (rounds 2-5).
DGRAD_F6 = os.environ.get("MD_DGRAD_F6", "1") == "1"
# the lift of the data gradient: "dyn" (default) = per launch from the tensor's own maximum (md_absmax: one more read of dy, no host
# round trip: operand pass and conv both derive the exponent from the same device word); an integer = the constant 2^that (A/B)
DGRAD_LIFT = os.environ.get("MD_DGRAD_LIFT", "dyn")
DGRAD_TSCALE = 2.0 ** int(DGRAD_LIFT) if DGRAD_LIFT != "dyn" else 64.0
DGRAD_WSCALE_LOG2 = 8
_AMAX_SLOTS = {}


def amax_slot(device):
    """A zeroed int32 [1] device word out of the per-(device, stream) arena (refilled when used up)."""
    device = torch.device(device)
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    arena = _AMAX_SLOTS.get(key)
    if arena is None or arena[1] >= arena[0].numel():
        arena = _AMAX_SLOTS[key] = [torch.zeros(1024, dtype=torch.int32, device=device), 0]
    word = arena[0][arena[1]:arena[1] + 1]
    arena[1] += 1
    return word


def absmax_word(x):
    """int32 [1] device word holding max |x| of the fp32 tensor as a float bit pattern (md_absmax) -- for wino_prep(dual, f8="f6", amax=...)
    and conv3_wino(amax=...)."""
    lib = _lib.load()
    word = amax_slot(x.device)
    check(lib.md_absmax(_ptr(x), x.numel(), _ptr(word), _stream()), "md_absmax")
    return word


class WinoWeightF6Dgrad:
    """Conv3d weight [Co][Ci][3][3][3] -> the f16f6 fragments of its DATA-GRADIENT conv W'[ci][co][t] = W[co][ci][26 - t] (read in
    place), packed by md_pack_batch (MD_PACK_WINO_F6: fixed pre-scale 2^DGRAD_WSCALE_LOG2, no equaliser: the operand is a gradient)."""
    fmt, eq = "f6", None

    def __init__(self, w, device):
        lib = _lib.load()
        w = w.detach().to(device=device, dtype=torch.float32).contiguous()
        _require_cuda(w, "weight")
        assert w.dim() == 5 and tuple(w.shape[2:]) == (3, 3, 3)
        self.rows, self.kdim = w.shape[1], w.shape[0]
        nbytes = lib.md_wino_weight_bytes_f8(self.rows, self.kdim)
        if nbytes <= 0:
            raise _lib.MeshDiffusionHipError("md_wino_weight_bytes_f8: unsupported weight shape")
        self._data, self._queued = None, False
        self._src = dict(kind=_lib.PACK_WINO_F6, w=w, w_off=0, nbytes=nbytes, rows=self.rows, kdim=self.kdim, taps=9, s_row=27,
                         s_k=self.rows * 27, s_tap=0, nt=0, kc=0, prec=DGRAD_WSCALE_LOG2, flip=1)

    request = WinoWeight.request
    data = WinoWeight.data


def wino_equaliser(gamma, beta, w, a2m=None):
    """float [Cin] on the device: the per-input-channel power-of-two equaliser of a GroupNorm(gamma, beta) -> SiLU -> Conv3d(w)
    pair for the f16f8 / f16f6 arithmetic (md_wino_equaliser, csrc/wino_eq.hip).  a2m: float [Cin] MEASURED mean squares of the
    operand (wino_operand_ms over a calibration evaluation) in place of the static estimate from (gamma, beta); gamma = beta = None
    (with a2m) for a conv no GroupNorm precedes (Upsample)."""
    lib = _lib.load()
    w = w.detach().to(dtype=torch.float32).contiguous()
    _require_cuda(w, "weight")
    cout, cin = w.shape[0], w.shape[1]
    assert w.dim() == 5 and tuple(w.shape[2:]) == (3, 3, 3)
    assert (gamma is None) == (beta is None) and (gamma is not None or a2m is not None)
    if gamma is not None:
        gamma = gamma.detach().to(device=w.device, dtype=torch.float32).contiguous()
        beta = beta.detach().to(device=w.device, dtype=torch.float32).contiguous()
        assert gamma.numel() == cin and beta.numel() == cin
    eq = torch.empty(cin, dtype=torch.float32, device=w.device)
    if a2m is not None:
        assert a2m.is_cuda and a2m.dtype == torch.float32 and a2m.numel() == cin and a2m.is_contiguous()
        check(lib.md_wino_equaliser_measured(_ptr(gamma), _ptr(beta), _ptr(w), _ptr(a2m), cout, cin, cin * 27, 27, _ptr(eq), _stream()),
              "md_wino_equaliser_measured")
    else:
        check(lib.md_wino_equaliser(_ptr(gamma), _ptr(beta), _ptr(w), cout, cin, cin * 27, 27, _ptr(eq), _stream()), "md_wino_equaliser")
    return eq


# Calibration (DDPMUNet3D.calibrate): while this is a dict, every inference conv on the Winograd path records the per-channel mean
# squares of the operand it reads -- {(id(layer), site): [layer, site, sum of ms vectors, evaluations]} -- from which the layer's
# MEASURED equaliser is built (layer._md_act_ms[site]; layers.conv3_wino_packed).
CALIBRATE = None


def wino_operand_ms(parts, ac, silu, B, P):
    """float [Cin]: mean over samples and positions of the squared operand (GroupNorm affine + SiLU applied when `ac`) of up to two
    concatenated F32B parts -- md_wino_operand_ms."""
    lib = _lib.load()
    assert 1 <= len(parts) <= 2
    cin = sum(c for _, c in parts)
    x2, c2 = (parts[1][0], parts[1][1]) if len(parts) == 2 else (None, 0)
    ms = torch.zeros(cin, dtype=torch.float32, device=parts[0][0].device)
    check(lib.md_wino_operand_ms(_ptr(parts[0][0]), _ptr(x2), parts[0][1], c2, _ptr(ac), 1 if silu else 0, B, P, _ptr(ms), _stream()),
          "md_wino_operand_ms")
    return ms


# fewest workgroups the Winograd kernel is launched with.  256 (one per CU) through round 4; with the f16f6 kernel half a chip of Winograd
# workgroups beats the direct bf16x3 kernel with split-K: res64 B = 1 15.00 -> 14.50 ms per step at 128 (the 32^3 level), 14.59 at 64, 15.38 at
# 32 (profiles/r05_b1_wino_floor.txt); sampling at batches >= 2 of the registered configs is unaffected (every level already has >= 256).
# Training steps (precision_scope(training=True)) keep 256: their kernels are the ones the gradient goldens and the B = 8 bench pinned.
WINO_MIN_WGS = int(os.environ.get("MD_WINO_MIN_WGS", "128"))
WINO_MIN_WGS_TRAIN = int(os.environ.get("MD_WINO_MIN_WGS_TRAIN", "256"))


def wino_ok(rows, kdim, S, B):
    """Shapes md_conv3_wino takes AND fills the chip with (one workgroup per CU, 128 rows x 4x8x8 positions each)."""
    return (WINO and PRECISION == "bf16x3" and rows % 128 == 0 and kdim % 32 == 0 and S % 8 == 0
            and B * (S ** 3 // 256) * (rows // 128) >= (max(WINO_MIN_WGS, WINO_MIN_WGS_TRAIN) if TRAINING_SCOPE else WINO_MIN_WGS))


WGRAD_NIN = os.environ.get("MD_WGRAD_NIN", "1") == "1"   # NIN weight gradients straight from S16B tensors (md_wgrad_nin)


def wgrad_nin_ok(co, ci, P):
    return WGRAD_NIN and co % 128 == 0 and ci % 128 == 0 and P % 16 == 0


def wgrad_nin(dy_s16, x_s16, B, co, ci, P, dw):
    """dw[ci][co] += sum x[ci] dy[co] over samples and positions, both operands S16B -- md_wgrad_nin (no PB16 tensors)."""
    lib = _lib.load()
    units = (co // 128) * (ci // 128)
    ksplit = max(1, min(256 // units, B * P // 16 // 8))
    nbytes = lib.md_wgrad_nin_workspace_bytes(co, ci, ksplit)
    if nbytes <= 0:
        raise _lib.MeshDiffusionHipError("md_wgrad_nin_workspace_bytes: unsupported shape")
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dy_s16.device)
    ev = _prof_begin()
    check(lib.md_wgrad_nin(_ptr(dy_s16), _ptr(x_s16), _ptr(dw), _ptr(ws), nbytes, B, co, ci, P, ksplit, 1, co, _stream()),
          "md_wgrad_nin")
    _prof_end(ev, "wgrad_nin", 2.0 * B * co * ci * P, 4.0 * B * (co + ci) * P, f"{ci}->{co}@{P}")


_WINO_SCRATCH = {}


def _wino_scratch(n_bf16, device, slot="t"):
    """T is written by md_wino_prep and read by the md_conv3_wino launched right after it on the same stream, so every
    (prep, conv) pair of a process can share ONE buffer, grown to the largest operand seen: no multi-GB allocation per conv
    (the caching allocator otherwise splits / re-merges 2-4 GB blocks among the training tape's tensors)."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream, slot)
    buf = _WINO_SCRATCH.get(key)
    if buf is None or buf.numel() < n_bf16:
        _WINO_SCRATCH.pop(key, None)
        buf = None
        buf = _WINO_SCRATCH[key] = torch.empty(n_bf16, dtype=torch.bfloat16, device=device)
    return buf[:n_bf16]


def release_scratch():
    """Drop the per-stream Winograd operand buffers (several GB after a training / sampling run at 64^3): the next
    wino_prep allocates again."""
    _WINO_SCRATCH.clear()


def wino_f8_ok(S, drop=None, keep=False, parts=None, normalised=True):
    """The f16f8 / f16f6 arithmetic of the Winograd path: inference launches (no dropout, T not kept for a backward) on grids the
    two-phase operand pass takes, whose operand comes out of a GroupNorm (`normalised`: the static equaliser applies; the raw
    residual stream in front of an Upsample conv stays in bf16x3).  Returns False or the cross-term format ("f8" / "f6"; f6 needs
    whole 16-channel blocks per part and falls back to f8 otherwise)."""
    if not (WINO_F8 and PRECISION == "bf16x3" and not drop and not keep and 256 % S == 0 and (normalised or F8_RAW)):
        return False
    if WINO_F8 == "f6" and parts is not None and any(c % 16 for _, c in parts):
        return "f8"
    return WINO_F8


def wino_prep(parts, ac, silu, ups, B, S, drop=None, keep=False, dual=False, sums=None, f8=False, eq=None, tscale=1.0, amax=None):
    """fp32 F32B parts (+ folded GroupNorm affine, SiLU, nearest-x2 upsampling) -> transformed split operand T.
    drop = (p, seed): training dropout after SiLU, the mask gn_apply(drop=...) produces for the same pair.
    keep: T goes to its own tensor instead of the shared scratch buffer (training forward: the Winograd weight gradient of the
    backward reads it again).  dual: returns (T, U) -- U = the dY operand of md_wgrad_wino (md_wino_prep_dual); sums (with dual):
    zeroed float [B, C] receiving the per-(sample, channel) sums of the tensor in the same pass (bias gradients).
    f8: False | "f8" (True) | "f6": the operand format of md_conv3_wino_f8 / _f6; eq (with f8): the layer's equaliser (wino_equaliser)."""
    lib = _lib.load()
    cin = sum(c for _, c in parts)
    assert 1 <= len(parts) <= 2
    assert eq is None or (f8 and eq.numel() == cin and eq.dtype == torch.float32 and eq.is_cuda), "eq belongs to the f16f8 / f16f6 operand"
    assert ac is not None or not silu, "SiLU is applied together with the folded GroupNorm affine (pass `ac`)"
    nbytes = lib.md_wino_operand_bytes(B, cin, S, S, S)
    if nbytes <= 0:
        raise _lib.MeshDiffusionHipError("md_wino_operand_bytes: unsupported operand shape")
    dev = parts[0][0].device
    t = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=dev) if keep else _wino_scratch(nbytes // 2, dev)
    x2, c2 = (parts[1][0], parts[1][1]) if len(parts) == 2 else (None, 0)
    ev = _prof_begin()
    args = (_ptr(parts[0][0]), _ptr(x2), parts[0][1], c2, _ptr(ac), 1 if silu else 0, 1 if ups else 0)
    tail = (B, S, S, S, drop[0] if drop else 0.0, drop[1] if drop else 0, _stream())
    if f8 and dual:
        # training backward: T = the f16f6 operand of tscale x the output gradient (data-gradient conv), U = the bf16 operand of the
        # unscaled gradient (md_wgrad_wino), sums = its channel sums
        assert f8 == "f6" and len(parts) == 1 and ac is None and not (ups or keep or drop or eq is not None) and 256 % S == 0
        u = _wino_scratch(nbytes // 2, dev, slot="u")
        check(lib.md_wino_prep_dual_f6(_ptr(parts[0][0]), cin, _ptr(t), _ptr(u), _ptr(sums), float(tscale), _ptr(amax), B, S, S, S, _stream()),
              "md_wino_prep_dual_f6")
    elif f8:
        assert not (dual or keep or drop), "the f16f8 / f16f6 operand is an inference format"
        if f8 == "f6":
            check(lib.md_wino_prep_f6(*args, _ptr(eq), _ptr(t), B, S, S, S, _stream()), "md_wino_prep_f6")
        else:
            check(lib.md_wino_prep_f8(*args, _ptr(eq), _ptr(t), B, S, S, S, _stream()), "md_wino_prep_f8")
    elif dual:
        if 256 % S:
            raise _lib.MeshDiffusionHipError("md_wino_prep_dual needs W | 256")
        u = _wino_scratch(nbytes // 2, dev, slot="u")
        check(lib.md_wino_prep_dual(*args, _ptr(t), _ptr(u), _ptr(sums), *tail), "md_wino_prep_dual")
    else:
        fn = lib.md_wino_prep_v2 if (WINO_PREP_V2 and 256 % S == 0) else lib.md_wino_prep
        check(fn(*args, _ptr(t), *tail), "md_wino_prep")
    _prof_end(ev, "wino_prep", 0.0, 4.0 * B * cin * (S ** 3 // (8 if ups else 1)) + (16.0 if dual else 8.0) * B * cin * S ** 3,   # fp32 in, 2 x bf16 x 2 out
              f"{cin}@{S}x{S}x{S}" + ("/ups" if ups else "") + ("/dual" if dual else "") + (("/f6" if f8 == "f6" else "/f8") if f8 else ""))
    # what this operand is, for conv3_wino's pairing check (an f8 T under f6 weights, or a T equalised with another layer's vector,
    # would be silently wrong: the kernel cannot tell)
    t._md_fmt = ("f6" if f8 == "f6" else "f8") if f8 else False
    t._md_eq = eq.data_ptr() if eq is not None else 0
    return (t, u) if dual else t


# Winograd weight gradient (csrc/wgrad_wino.hip): the training backward of the layers on the Winograd path contracts the
# forward's operand T (kept on the tape) with the transformed output gradient -- no S16B activations, no PB16 re-layout
WGRAD_WINO = os.environ.get("MD_WGRAD_WINO", "1") == "1"


def wgrad_wino_ok(co, ci, S, B):
    """Layers whose weight gradient md_wgrad_wino takes: on the Winograd forward path, whole 128-channel tiles, rows of 16 / 32 pairs."""
    return WGRAD_WINO and WINO_TRAIN_FWD and S in (32, 64) and co % 128 == 0 and ci % 128 == 0 and wino_ok(co, ci, S, B)


def wgrad_wino(u_dy, t_act, B, co, ci, S, dw):
    """dw[co][ci][27] += weight gradient from U (md_wino_prep_dual of dY) and the forward's T -- md_wgrad_wino."""
    lib = _lib.load()
    units = (co // 128) * (ci // 128) * 12
    ksplit = max(1, min(256 // units, B * (S - 1)))
    nbytes = lib.md_wgrad_wino_workspace_bytes(co, ci, ksplit)
    if nbytes <= 0:
        raise _lib.MeshDiffusionHipError("md_wgrad_wino_workspace_bytes: unsupported shape")
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=u_dy.device)
    ev = _prof_begin()
    check(lib.md_wgrad_wino(_ptr(u_dy), _ptr(t_act), _ptr(dw), _ptr(ws), nbytes, B, co, ci, S, S, S, ksplit, ci * 27, 27, 1,
                            _stream()), "md_wgrad_wino")
    _prof_end(ev, "wgrad_wino", 2.0 * B * co * ci * 27 * S ** 3, 16.0 * B * (co + ci) * S ** 3 / 2, f"{ci}->{co}@{S}x{S}x{S}")


WINO_VARIANT = int(os.environ.get("MD_WINO_VARIANT", "0"))


def conv3_wino(ww, t, B, S, *, bias=None, bias_bstride=0, residual=None, res_bstride=0, stats=None, out=None, variant=None, out_scale=1.0,
               amax=None):
    lib = _lib.load()
    P = S ** 3
    if out is None:
        out = f32b_empty(B, ww.rows, P, t.device)
    ev = _prof_begin()
    f8 = isinstance(ww, (WinoWeightF8, WinoWeightF6Dgrad))        # the weight object fixes the arithmetic; `t` must come from wino_prep(f8=...) accordingly
    t_fmt = getattr(t, "_md_fmt", None)
    if t_fmt is not None:                    # an operand that went through wino_prep: format and equaliser must be the weight object's
        want = ww.fmt if f8 else False
        if t_fmt != want or getattr(t, "_md_eq", 0) != ((ww.eq.data_ptr() if ww.eq is not None else 0) if f8 else 0):
            raise _lib.MeshDiffusionHipError(f"conv3_wino: operand format {t_fmt!r} / equaliser does not belong to these weights ({want!r})")
    if f8 and (out_scale != 1.0 or amax is not None):
        assert ww.fmt == "f6"
        check(lib.md_conv3_wino_f6_scaled(_ptr(t), _ptr(ww.data), _ptr(out), _ptr(bias), bias_bstride, _ptr(residual), res_bstride,
                                          _ptr(stats), B, ww.kdim, ww.rows, S, S, S, float(out_scale), _ptr(amax), _stream()),
              "md_conv3_wino_f6_scaled")
    elif f8:
        fn = lib.md_conv3_wino_f6 if ww.fmt == "f6" else lib.md_conv3_wino_f8
        check(fn(_ptr(t), _ptr(ww.data), _ptr(out), _ptr(bias), bias_bstride, _ptr(residual), res_bstride,
                 _ptr(stats), B, ww.kdim, ww.rows, S, S, S, _stream()), "md_conv3_wino_" + ww.fmt)
    else:
        check(lib.md_conv3_wino(_ptr(t), _ptr(ww.data), _ptr(out), _ptr(bias), bias_bstride, _ptr(residual), res_bstride,
                                _ptr(stats), B, ww.kdim, ww.rows, S, S, S, WINO_VARIANT if variant is None else variant, _stream()),
              "md_conv3_wino")
    _prof_end(ev, "wino", 2.0 * B * ww.rows * ww.kdim * 27 * P,
              4.0 * (2 * B * ww.kdim * P + ww.rows * ww.kdim * 36 + B * ww.rows * P * (2 if residual is not None else 1)),
              f"{ww.kdim}->{ww.rows}@{S}x{S}x{S}" + ("/res" if residual is not None else "") + ("/stats" if stats is not None else "")
              + (("/" + ww.fmt) if f8 else ""))
    return out


# Stride-2 3x3x3 convolution of Downsample on the raw fp32 tensor (csrc/conv3_s2.hip): no split pass, slab-wise K loop
CONV3_S2 = os.environ.get("MD_CONV3_S2", "1") == "1"   # A/B switch: 0 = md_gn_apply (split) + md_gemm_conv(CFG_C3_S2)
CFG_S2_PACK = CFG_C3_128_K16                           # tile geometry of its packed weights: nt = 128, kc = 16, 27 taps


S2_MIN_WGS = int(os.environ.get("MD_S2_MIN_WGS", "100"))   # no split-K in md_conv3_s2: below this the generic tile (split-K) is faster


def conv3_s2_ok(rows, kdim, S_out, B=None):
    """Shapes md_conv3_s2 takes; with `B`: and is launched with enough workgroups for (measured, B = 8: 16^3 outputs 0.16 ms
    against 0.20 + the split pass; 8^3 outputs = 32 workgroups 0.25 against 0.10)."""
    ok = CONV3_S2 and PRECISION == "bf16x3" and kdim % 32 == 0 and rows % 8 == 0 and S_out % 8 == 0
    return ok and (B is None or B * (S_out ** 3 // 256) * ((rows + 127) // 128) >= S2_MIN_WGS)


def conv3_s2(pw, x, B, S_out, *, bias=None, bias_bstride=0, stats=None, out=None):
    """out F32B [B][rows][S_out^3] = stride-2 conv (pad (0, 1)) of the F32B tensor x [B][kdim/8][(2 S_out)^3][8] with the
    CFG_S2_PACK tiles `pw` -- md_conv3_s2."""
    lib = _lib.load()
    P = S_out ** 3
    rows_alloc = ((pw.rows + 7) // 8) * 8
    if out is None:
        out = f32b_empty(B, rows_alloc, P, x.device)
    ev = _prof_begin()
    check(lib.md_conv3_s2(_ptr(x), _ptr(pw.data), _ptr(out), _ptr(bias), bias_bstride, _ptr(stats), B, pw.kdim, pw.rows,
                          rows_alloc, S_out, S_out, S_out, _stream()), "md_conv3_s2")
    _prof_end(ev, "s2", 2.0 * B * pw.rows * pw.kdim * 27 * P, 4.0 * (B * pw.kdim * 8 * P + pw.rows * pw.kdim * 27 + B * pw.rows * P),
              f"{pw.kdim}->{pw.rows}@{S_out}x{S_out}x{S_out}" + ("/stats" if stats is not None else ""))
    return out


# The dx-folded 3x3x3 stem in its own kernel (csrc/conv3_stem.hip), with the GroupNorm sums of its output
CONV3_STEM = os.environ.get("MD_CONV3_STEM", "1") == "1"   # A/B switch: 0 = md_gemm_conv(CFG_C3X_128_K16) + md_gn_stats


def conv3_stem_ok(rows, kdim, S):
    return CONV3_STEM and PRECISION == "bf16x3" and kdim == 16 and rows % 8 == 0 and S % 8 == 0


def conv3_stem(pw, x16, B, S, *, bias=None, residual=None, stats=None):
    """out F32B [B][rows][S^3] of the dx-folded stem on the x-folded S16B operand (md_ncdhw_to_s16b_xfold) -- md_conv3_stem."""
    lib = _lib.load()
    P = S ** 3
    out = f32b_empty(B, pw.rows, P, x16.device)
    ev = _prof_begin()
    check(lib.md_conv3_stem(_ptr(x16), _ptr(pw.data), _ptr(out), _ptr(bias), _ptr(residual), _ptr(stats), B, pw.rows, S, S, S,
                            _stream()), "md_conv3_stem")
    _prof_end(ev, "stem", 2.0 * B * pw.rows * pw.kdim * 9 * P, 4.0 * (B * pw.kdim * P + pw.rows * pw.kdim * 9 + B * pw.rows * P),
              f"{pw.kdim}->{pw.rows}@{S}x{S}x{S}" + ("/res" if residual is not None else "") + ("/stats" if stats is not None else ""))
    return out


# GroupNorm + SiLU + dx-folded 3x3x3 head in one kernel (csrc/conv3_head.hip): no md_gn_apply pass, no generic tile
CONV3_HEAD = os.environ.get("MD_CONV3_HEAD", "1") == "1"   # A/B switch: 0 = md_gn_apply + md_gemm_conv(CFG_C3X_32)
CFG_HEAD_PACK = CFG_C5X_32_K16                             # tile geometry of its packed weights: nt = 32, kc = 16


def conv3_head_ok(rows, cin, S):
    return CONV3_HEAD and FUSE_GN_APPLY and PRECISION == "bf16x3" and rows <= 32 and cin % 32 == 0 and S % 8 == 0


def conv3_head(pw, x, ac, B, S, rows_alloc):
    """y F32B [B][rows_alloc][S^3] (rows = (co, kw)) of the dx-folded head on the un-normalised F32B tensor x with the folded
    GroupNorm affine `ac`; SiLU inside -- md_conv3_head."""
    lib = _lib.load()
    P = S ** 3
    y = f32b_empty(B, rows_alloc, P, x.device)
    ev = _prof_begin()
    check(lib.md_conv3_head(_ptr(x), _ptr(ac), _ptr(pw.data), _ptr(y), B, pw.kdim, rows_alloc, S, S, S, _stream()), "md_conv3_head")
    _prof_end(ev, "head", 2.0 * B * pw.rows * pw.kdim * 9 * P, 4.0 * (B * pw.kdim * P + pw.rows * pw.kdim * 9 + B * rows_alloc * P),
              f"{pw.kdim}->{pw.rows}@{S}x{S}x{S}")
    return y


def _prof_begin():
    if PROFILE is None:
        return None
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    return e0


def _prof_end(e0, cfg, flops, abytes, tag):
    if e0 is None:
        return
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    PROFILE.append((cfg, flops, e0, e1, abytes, tag))


# ---------------------------------------------------------------------------------------------
# GroupNorm (+SiLU) + split, with concatenated sources
# ---------------------------------------------------------------------------------------------
FUSE_GN_APPLY = os.environ.get("MD_FUSE_GN_APPLY", "1") == "1"   # GroupNorm affine + SiLU + split inside the conv's halo loader (inference)
FUSE_GN_STATS = True     # take GroupNorm sums from the producing conv's epilogue when it recorded them (A/B switch)
SPLITK_STATS = os.environ.get("MD_SPLITK_STATS", "1") == "1"   # ... and from the split-K finish kernel (8^3 / 4^3 levels)


# GroupNorm sum buffers ([B][C][2] float64, zero before the producing conv's epilogue / md_gn_stats adds into them) come
# out of one arena per device that the U-Net zeroes ONCE at the start of a forward (stats_arena_reset): ~90 fill launches
# per sampling step otherwise (torch.zeros per conv, md_zero per GroupNorm).  Without a reset nothing is ever handed out
# twice: an exhausted arena is replaced by a fresh zeroed one.
STATS_ARENA = os.environ.get("MD_STATS_ARENA", "1") == "1"
_STATS_ARENAS = {}
_STATS_ARENA_DOUBLES = 1 << 20


def stats_arena_reset(device):
    """Start of a forward: everything handed out before is dead (consumed by the GroupNorms of the previous forward)."""
    if not STATS_ARENA:
        return
    a = _STATS_ARENAS.get(device)
    if a is None:
        return
    if a["off"] > 0:
        a["buf"][:a["off"]].zero_()
    a["off"] = 0


def stats_zeros(B, rows, device):
    """Zeroed float64 [B][rows][2]."""
    n = B * rows * 2
    if not STATS_ARENA or n > _STATS_ARENA_DOUBLES // 4:
        return torch.zeros((B, rows, 2), dtype=torch.float64, device=device)
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    a = _STATS_ARENAS.get(device)
    if a is None or a["off"] + n > _STATS_ARENA_DOUBLES:
        a = _STATS_ARENAS[device] = dict(buf=torch.zeros(_STATS_ARENA_DOUBLES, dtype=torch.float64, device=device), off=0)
    out = a["buf"][a["off"]:a["off"] + n].view(B, rows, 2)
    a["off"] += (n + 15) & ~15
    return out


def gn_params(parts, gamma, beta, B, P, eps=1e-6, groups=32, want_ac=False):
    """parts: list of (F32B tensor, C).  Returns the float4 params tensor [B][Ctot][4] (want_ac: and the folded affine
    [B][Ctot][2] = (rstd*gamma, beta - mean*rstd*gamma) that md_gemm_conv's fused operand loader applies).
    A part whose producer attached `_md_sums` (layers.run_conv3(want_stats=True)) is not read again."""
    lib = _lib.load()
    dev = parts[0][0].device
    ctot = sum(c for _, c in parts)
    cached = [getattr(t, "_md_sums", None) if FUSE_GN_STATS else None for t, _ in parts]
    if len(parts) == 1 and cached[0] is not None:
        sums = cached[0]           # the producing conv already accumulated them in its epilogue
    else:
        sums = stats_zeros(B, ctot, dev) if any(c is None for c in cached) else torch.empty((B, ctot, 2), dtype=torch.float64, device=dev)
        off = 0
        for (t, c), cs in zip(parts, cached):
            if cs is not None:
                sums[:, off:off + c] = cs
            else:
                check(lib.md_gn_stats(_ptr(t), _ptr(sums), B, c, P, ctot, off, _stream()), "md_gn_stats")
            off += c
    params = torch.empty((B, ctot, 4), dtype=torch.float32, device=dev)
    ac = torch.empty((B, ctot, 2), dtype=torch.float32, device=dev) if want_ac else None
    check(lib.md_gn_finalize(_ptr(sums), _ptr(gamma), _ptr(beta), _ptr(params), B, ctot, groups, P,
                             eps, _ptr(ac), _stream()), "md_gn_finalize")
    return (params, ac) if want_ac else params


def next_dropout_seed():
    """A fresh 62-bit mask seed from torch's global CPU generator (so `torch.manual_seed` pins the masks)."""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


def dropout_scale(B, C, P, p, seed, device):
    """The mask md_gn_apply applies for (p, seed) as an F32B tensor of {0, 1/(1-p)} (parity tests)."""
    lib = _lib.load()
    out = f32b_empty(B, C, P, device)
    check(lib.md_dropout_scale(_ptr(out), B, C, P, C, 0, float(p), int(seed), _stream()), "md_dropout_scale")
    return out


def gn_apply(parts, params, B, P, norm=True, silu=True, out=None, fp16=False, want_raw=False, drop=None):
    """fp16=True: MD_PREC_FP16X2 operand format (plane 0 = fp16(y), plane 1 untouched).
    want_raw=True: also return the bf16 split of the raw input (one read, two writes).
    drop=(p, seed): training dropout after the activation (mask regenerated from the seed in the backward)."""
    dp, dseed = (float(drop[0]), int(drop[1])) if drop else (0.0, 0)
    lib = _lib.load()
    dev = parts[0][0].device
    ctot = sum(c for _, c in parts)
    if out is None:
        out = s16b_empty(B, ctot, P, dev)
    raw = s16b_empty(B, ctot, P, dev) if want_raw else None
    off = 0
    for t, c in parts:
        check(lib.md_gn_apply(_ptr(t), _ptr(params) if norm else None, _ptr(out), _ptr(raw), B, c, P, ctot, off,
                              1 if norm else 0, (1 if silu else 0) | DEBUG_ACT_FP16 | (4 if fp16 else 0), dp, dseed,
                              _stream()),
              "md_gn_apply")
        off += c
    return (out, raw) if want_raw else out


# ---------------------------------------------------------------------------------------------
# small ops
# ---------------------------------------------------------------------------------------------
def fold_dx(y, bias, B, co, kx, rows_alloc, S, out=None):
    """Second half of the dx-folded head conv: F32B [B][rows_alloc][S^3] with rows (co, dx) -> NCDHW [B, co, S, S, S]."""
    lib = _lib.load()
    if out is None:
        out = torch.empty((B, co, S, S, S), dtype=torch.float32, device=y.device)
    check(lib.md_fold_dx(_ptr(y), _ptr(bias), _ptr(out), B, co, kx, rows_alloc, S, S, S, _stream()), "md_fold_dx")
    return out


def timestep_embedding(t, dim):
    lib = _lib.load()
    _require_cuda(t, "timesteps")
    t = t.to(torch.float32).contiguous()
    emb = torch.empty((t.shape[0], dim), dtype=torch.float32, device=t.device)
    check(lib.md_timestep_embedding(_ptr(t), _ptr(emb), t.shape[0], dim, _stream()), "md_timestep_embedding")
    return emb


def linear(x, w, bias, silu_in=False):
    lib = _lib.load()
    B, in_dim = x.shape
    out_dim = w.shape[0]
    y = torch.empty((B, out_dim), dtype=torch.float32, device=x.device)
    check(lib.md_linear(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), B, in_dim, out_dim, 1 if silu_in else 0,
                        _stream()), "md_linear")
    return y


def ncdhw_to_s16b(x, c_pad):
    lib = _lib.load()
    _require_cuda(x, "x")
    x = x.to(torch.float32).contiguous()
    B, Cc = x.shape[0], x.shape[1]
    P = x[0, 0].numel()
    out = s16b_empty(B, c_pad, P, x.device)
    check(lib.md_ncdhw_to_s16b(_ptr(x), _ptr(out), B, Cc, c_pad, P, _stream()), "md_ncdhw_to_s16b")
    return out


def ncdhw_to_s16b_xfold(x, kx, c_pad):
    """[B,C,S,S,S] fp32 -> S16B with kx x-shifted copies per channel (operand of the dx-folded stem conv)."""
    lib = _lib.load()
    _require_cuda(x, "x")
    x = x.to(torch.float32).contiguous()
    B, Cc, D, H, W = x.shape
    out = s16b_empty(B, c_pad, D * H * W, x.device)
    check(lib.md_ncdhw_to_s16b_xfold(_ptr(x), _ptr(out), B, Cc, kx, c_pad, D, H, W, _stream()), "md_ncdhw_to_s16b_xfold")
    return out


def ncdhw_to_f32b(x):
    lib = _lib.load()
    _require_cuda(x, "x")
    x = x.to(torch.float32).contiguous()
    B, Cc = x.shape[0], x.shape[1]
    P = x[0, 0].numel()
    out = f32b_empty(B, Cc, P, x.device)
    check(lib.md_ncdhw_to_f32b(_ptr(x), _ptr(out), B, Cc, P, _stream()), "md_ncdhw_to_f32b")
    return out


def f32b_to_ncdhw(x, spatial):
    lib = _lib.load()
    B, CG, P, _ = x.shape
    out = torch.empty((B, CG * 8) + tuple(spatial), dtype=torch.float32, device=x.device)
    check(lib.md_f32b_to_ncdhw(_ptr(x), _ptr(out), B, CG * 8, P, _stream()), "md_f32b_to_ncdhw")
    return out


def s16b_to_ncdhw(x, spatial):
    lib = _lib.load()
    B, CG, _, P, _ = x.shape
    out = torch.empty((B, CG * 8) + tuple(spatial), dtype=torch.float32, device=x.device)
    check(lib.md_s16b_to_ncdhw(_ptr(x), _ptr(out), B, CG * 8, P, _stream()), "md_s16b_to_ncdhw")
    return out


NIN_STREAM = os.environ.get("MD_NIN_STREAM", "1") == "1"   # persistent streaming kernel for the HBM-bound shortcut NINs


def nin_stream_ok(parts, rows, P):
    k = sum(c for _, c in parts)
    return (NIN_STREAM and rows == 128 and k in (128, 256) and len(parts) <= 2 and all(c % 16 == 0 for _, c in parts)
            and P % 256 == 0 and P >= 32768)


def nin_f32(parts, pw, bias, B, P):
    """out F32B [B][128][P] = NIN(cat(parts)) with `pw` = PackedWeight(W, "nin", CFG_G1_128 / _N128) -- md_nin_f32."""
    lib = _lib.load()
    out = f32b_empty(B, 128, P, parts[0][0].device)
    x2, c2 = (parts[1][0], parts[1][1]) if len(parts) == 2 else (None, 0)
    check(lib.md_nin_f32(_ptr(parts[0][0]), _ptr(x2), parts[0][1], c2, _ptr(pw.data), _ptr(bias), _ptr(out), B, 128, P, 0,
                         _stream()), "md_nin_f32")
    return out


FUSE_ATTN = os.environ.get("MD_FUSE_ATTN", "1") == "1"   # fused QK^T / online softmax / PV kernel where it applies (inference)


def attn_fused_ok(Cc, n_tokens):
    return FUSE_ATTN and Cc == 256 and n_tokens % 128 == 0


def attn_fwd(qk, vT, bias_v, B, Cc, n_tokens, scale):
    """qk: S16B [B][2C][N] (q | k), vT: S16B over tokens [B][N][C] -> o: S16B [B][C][N] (md_attn_fwd)."""
    lib = _lib.load()
    o = s16b_empty(B, Cc, n_tokens, qk.device)
    check(lib.md_attn_fwd(_ptr(qk), _ptr(vT), _ptr(o), _ptr(bias_v), B, Cc, n_tokens, float(scale), _stream()), "md_attn_fwd")
    return o


def softmax_keys(s, B, nk, nq):
    lib = _lib.load()
    p = torch.empty((B, nk // 8, 2, nq, 8), dtype=torch.bfloat16, device=s.device)
    check(lib.md_softmax_keys(_ptr(s), _ptr(p), B, nk, nq, _stream()), "md_softmax_keys")
    return p


def ancestral_step(x, eps, z, mask, coef):
    """x, eps, z: [B,C,D,H,W] fp32; mask [P] or None; coef [B,4] = beta, sigma, sqrt(1-beta), sqrt(beta)."""
    lib = _lib.load()
    for name, t in (("x", x), ("eps", eps), ("z", z)):
        _require_cuda(t, name)
    x, eps, z = x.contiguous(), eps.contiguous(), z.contiguous()
    B, Cc = x.shape[0], x.shape[1]
    P = x[0, 0].numel()
    x_out, xm_out = torch.empty_like(x), torch.empty_like(x)
    check(lib.md_ancestral_step(_ptr(x), _ptr(eps), _ptr(z), _ptr(mask), _ptr(coef), _ptr(x_out),
                                _ptr(xm_out), B, Cc, P, _stream()), "md_ancestral_step")
    return x_out, xm_out


def ddim_step(x64, eps, mask, coef, partial=None, pmask=None, ch=0):
    """x64 [B,C,D,H,W] float64 state, eps float32 U-Net output, mask [P] or None, coef [B,4] float64 = a1, a2, r1, r2;
    partial / pmask: [P] float32 each or None.  Returns (x_new f64, x0_pred f64, x_new as f32)."""
    lib = _lib.load()
    for name, t in (("x", x64), ("eps", eps)):
        _require_cuda(t, name)
    assert x64.dtype == torch.float64 and eps.dtype == torch.float32 and coef.dtype == torch.float64
    x64, eps, coef = x64.contiguous(), eps.contiguous(), coef.contiguous()
    B, Cc = x64.shape[0], x64.shape[1]
    P = x64[0, 0].numel()
    x_out, x0_out = torch.empty_like(x64), torch.empty_like(x64)
    x_f32 = torch.empty_like(eps)
    check(lib.md_ddim_step(_ptr(x64), _ptr(eps), _ptr(mask), _ptr(coef), _ptr(partial), _ptr(pmask), int(ch), _ptr(x_out),
                           _ptr(x0_out), _ptr(x_f32), B, Cc, P, _stream()), "md_ddim_step")
    return x_out, x0_out, x_f32


def inpaint_blend_(x, src, pmask, gmask, ch, src_bstride=0):
    """In place on channel `ch` of x: (x*(1-pmask) + src*pmask) * gmask."""
    lib = _lib.load()
    B, Cc = x.shape[0], x.shape[1]
    P = x[0, 0].numel()
    check(lib.md_inpaint_blend(_ptr(x), _ptr(src), _ptr(pmask), _ptr(gmask), B, Cc, ch, P, src_bstride,
                               _stream()), "md_inpaint_blend")
    return x


def inpaint_renoise_(x, x_mean, z, pmask, gmask, coef, ch):
    """In place: re-noise channel `ch` inside pmask (see md_inpaint_renoise)."""
    lib = _lib.load()
    B, Cc = x.shape[0], x.shape[1]
    P = x[0, 0].numel()
    check(lib.md_inpaint_renoise(_ptr(x), _ptr(x_mean), _ptr(z), _ptr(pmask), _ptr(gmask), _ptr(coef), B, Cc,
                                 ch, P, _stream()), "md_inpaint_renoise")
    return x


def ksplit_for(cfg, batch, rows, kdim, spatial_out):
    """Split the K-chunk loop when output tiles alone cannot fill the 256 CUs (4^3 / 8^3 levels)."""
    info = _lib.cfg_info(cfg)
    tiles = max(1, spatial_out ** 3 // info["cols"]) * batch * ((rows + info["nt"] - 1) // info["nt"])
    ncc = kdim // info["kc"]
    ks = 1
    while tiles * ks < 256 and ks * 2 <= ncc and ks < 32:
        ks *= 2
    return ks


def conv_cfg_for(spatial, stride=1):
    """Pick the 3x3x3 tile configuration for an OUTPUT grid of edge `spatial`."""
    if stride == 2:
        return CFG_C3_S2
    # CFG_C3_128_FAST: dedicated kernel (F32B output only); same WPK weight tiles as CFG_C3_128
    return CFG_C3_128_FAST if spatial % 8 == 0 else CFG_C3_LOW


NIN_N128 = os.environ.get("MD_NIN_N128", "1") == "1"   # A/B: 128-column GEMM tile (3 workgroups / CU) for the HBM-bound shortcut NINs


def gemm_cfg_for(ncols, nrows, hbm_bound=False):
    """hbm_bound: a 1x1x1 layer over a big grid with few channels (ResnetBlock shortcut): prefer the 128-column tile."""
    if hbm_bound and NIN_N128 and ncols % 128 == 0 and ncols >= 32768:
        return CFG_G1_128_N128
    if ncols % 256 == 0:
        return CFG_G1_128
    return CFG_G1_128_LOW if nrows % 128 == 0 else CFG_G1_64_LOW


def attn_scale(channels):
    return float(int(channels) ** (-0.5))


__all__ = [n for n in dir() if not n.startswith("_")]
_ = math
