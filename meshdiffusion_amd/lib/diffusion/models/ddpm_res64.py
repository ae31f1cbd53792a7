"""DDPM 3-D U-Net score network for 64^3 x 4 DMTet grids on MI355X.

Drop-in for the reference's lib/diffusion/models/ddpm_res64.py (`DDPMRes64`, __init__ :41-124,
forward :126-199): same registry name, constructor argument (config), state-dict keys/shapes
(`all_modules.{i}.*`, `pos_layer`, `mask_layer`, `coords`, `mask`, `sigmas`) and call signature
`model(x[B,4,R,R,R], labels[B]) -> eps_hat[B,4,R,R,R]`.

The forward never leaves the HIP library: activations stay in the 8-channel blocked layouts
(F32B residual stream, S16B split-bf16 conv operands) between kernels; `torch.cat` of skip
connections is expressed as multi-part GroupNorm/split calls, the nearest-neighbour upsample is
folded into the conv's halo load, and `pos_layer(coords) + mask_layer(mask)` (input independent)
is computed once per weight version and added in the stem conv's epilogue.
"""
import numpy as np
import torch
import torch.nn as nn

from . import layers, utils
from .... import hip_ops as ops

ResnetBlockDDPM = layers.ResnetBlockDDPM
Upsample = layers.Upsample
Downsample = layers.Downsample
AttnBlock = layers.AttnBlock
conv3x3 = layers.ddpm_conv3x3
default_initializer = layers.default_init


def _dense(in_dim, out_dim):
    lin = nn.Linear(in_dim, out_dim)
    lin.weight.data = default_initializer()(lin.weight.data.shape)
    nn.init.zeros_(lin.bias)
    return lin


class DDPMUNet3D(layers.HipLayer):
    """Shared U-Net body; the two registered models differ only in the three class attributes."""
    KSIZE = 3            # stem / mask_layer / pos_layer / head kernel extent
    USE_COORDS = True    # h0 += pos_layer(coords)
    LEVEL0_BLOCKS = None  # None -> num_res_blocks

    def __init__(self, config):
        super().__init__()
        m = config.model
        self.act = layers.get_act(config)
        self.register_buffer("sigmas", torch.tensor(utils.get_sigmas(config)))
        self.nf = nf = m.nf
        ch_mult = m.ch_mult
        self.num_res_blocks = nrb = m.num_res_blocks
        self.attn_resolutions = attn_res = m.attn_resolutions
        self.num_resolutions = nres = len(ch_mult)
        self.all_resolutions = all_res = [config.data.image_size // (2 ** i) for i in range(nres)]
        self.conditional = m.conditional
        self.centered = config.data.centered
        self.scale_by_sigma = m.scale_by_sigma
        self.img_size = R = config.data.image_size
        self.num_freq = int(np.log2(R))
        channels = config.data.num_channels
        if not self.conditional:
            raise NotImplementedError("unconditional (no timestep) variant is not used by this path")
        block = lambda **kw: ResnetBlockDDPM(act=self.act, temb_dim=4 * nf, dropout=m.dropout, **kw)  # noqa: E731

        ks = self.KSIZE
        edge_conv = (lambda i, o, **kw: layers._conv(i, o, ks, 1, ks // 2, kw.get("init_scale", 1.0)))
        mods = [_dense(nf, 4 * nf), _dense(4 * nf, 4 * nf)]
        # constant inputs kept as (frozen) parameters because they are part of the checkpoint
        if self.USE_COORDS:
            self.coords = nn.Parameter(torch.zeros(1, 3, R, R, R), requires_grad=False)
        self.mask = nn.Parameter(torch.zeros(1, 1, R, R, R), requires_grad=False)
        self.pos_layer = edge_conv(3, nf)
        self.mask_layer = edge_conv(1, nf)
        mods.append(edge_conv(channels, nf))
        skip_ch, in_ch = [nf], nf
        for lvl in range(nres):
            out_ch = nf * ch_mult[lvl]
            for _ in range(self._blocks_at(lvl)):
                mods.append(block(in_ch=in_ch, out_ch=out_ch))
                in_ch = out_ch
                if all_res[lvl] in attn_res:
                    mods.append(AttnBlock(channels=in_ch))
                skip_ch.append(in_ch)
            if lvl != nres - 1:
                mods.append(Downsample(channels=in_ch, with_conv=m.resamp_with_conv))
                skip_ch.append(in_ch)
        mods += [block(in_ch=in_ch), AttnBlock(channels=in_ch), block(in_ch=in_ch)]
        for lvl in reversed(range(nres)):
            out_ch = nf * ch_mult[lvl]
            for _ in range(self._blocks_at(lvl) + 1):
                mods.append(block(in_ch=in_ch + skip_ch.pop(), out_ch=out_ch))
                in_ch = out_ch
            if all_res[lvl] in attn_res:
                mods.append(AttnBlock(channels=in_ch))
            if lvl != 0:
                mods.append(Upsample(channels=in_ch, with_conv=m.resamp_with_conv))
        assert not skip_ch
        mods.append(nn.GroupNorm(num_channels=in_ch, num_groups=32, eps=1e-6))
        mods.append(edge_conv(in_ch, channels, init_scale=0.0))
        self.all_modules = nn.ModuleList(mods)
        self.out_channels = channels
        # arithmetic of this model's inference convs: "bf16x3" | "f16f8" | "f16f6" | "fp16x2" (hip_ops); None = the
        # process default.  It is entered as a scope by every call and left again: a property of the model, not of the process.
        self.hip_precision = m.get("hip_precision", None) if hasattr(m, "get") else None

    def _blocks_at(self, lvl):
        return self.LEVEL0_BLOCKS if (lvl == 0 and self.LEVEL0_BLOCKS) else self.num_res_blocks

    # ---- cached, input-independent pieces --------------------------------------------------
    def _stem_const(self):
        """pos_layer(coords) + mask_layer(mask) (+ both biases) as one F32B [1][nf][P] tensor."""
        ps = [self.mask, self.mask_layer.weight, self.mask_layer.bias]
        if self.USE_COORDS:
            ps += [self.coords, self.pos_layer.weight, self.pos_layer.bias]

        def build():
            R = self.img_size
            dev = self.mask.device
            cfg = self._stem_cfg()
            t = None
            if self.USE_COORDS:
                wp = ops.PackedWeight(self.pos_layer.weight, "conv", cfg, dev)
                c16 = ops.ncdhw_to_s16b(self.coords.detach(), 16)
                t = layers.run_conv3(wp, c16, 1, R, bias=self.pos_layer.bias)
            wm = ops.PackedWeight(self.mask_layer.weight, "conv", cfg, dev)
            m16 = ops.ncdhw_to_s16b(self.mask.detach(), 16)
            return layers.run_conv3(wm, m16, 1, R, bias=self.mask_layer.bias, residual=t)

        return self._cached("stem_const", ps, build)

    def _film_table(self):
        """All ResnetBlocks' Dense_0 weights stacked [sum(out_ch), 4*nf] with (Dense_0.bias + Conv_0.bias):
        one md_linear launch per evaluation instead of 37."""
        blocks = [m for m in self.all_modules if isinstance(m, ResnetBlockDDPM)]
        ps = [p for b in blocks for p in (b.Dense_0.weight, b.Dense_0.bias, b.Conv_0.bias)]

        def build():
            w = torch.cat([b.Dense_0.weight.detach() for b in blocks], dim=0).contiguous()
            bias = torch.cat([(b.Dense_0.bias.detach() + b.Conv_0.bias.detach()) for b in blocks]).contiguous()
            offs, o = {}, 0
            for b in blocks:
                offs[id(b)] = o
                o += b.out_ch
            return w, bias, offs, o

        return self._cached("film", ps, build)

    # ---- training (bf16x3 operands; both architectures) --------------
    def _autograd_anchor(self):
        a = self.__dict__.get("_md_anchor")
        if a is None or a.device != self.mask.device:
            a = torch.zeros((), device=self.mask.device, requires_grad=True)
            self.__dict__["_md_anchor"] = a
        return a

    def forward_train(self, x, labels):
        """Forward pass that records what `backward` needs.  Returns (eps_hat NCDHW, ctx).  Always bf16x3 (hip_ops.precision_scope)."""
        with ops.precision_scope(None, training=True):
            return self._forward_train(x, labels)

    def backward(self, ctx, d_eps):
        """Accumulate d(loss)/d(parameter) into `.grad` for every trainable parameter, given d(loss)/d(eps_hat)."""
        with ops.precision_scope(None, training=True):
            return self._backward(ctx, d_eps)

    def _forward_train(self, x, labels):
        if self.scale_by_sigma:
            # the inference forward divides by sigma[labels] (ddpm_res64.py:196-198 of the reference); neither registered
            # config enables it, and the HIP backward does not carry the 1/sigma factor
            raise NotImplementedError("training with model.scale_by_sigma=True is not implemented on the HIP path")
        assert tuple(x.shape[1:]) == (self.out_channels, self.img_size, self.img_size, self.img_size)
        ops.stats_arena_reset(x.device)
        ops.prewarm_packs()      # every weight changed since the last step: re-pack what that step used in one launch
        mods = self.all_modules
        B, R = x.shape[0], self.img_size
        P = R ** 3
        i = 0
        emb = layers.get_timestep_embedding(labels, self.nf)
        t1 = ops.linear(emb, mods[0].weight, mods[0].bias); i += 1
        temb = ops.linear(t1, mods[1].weight, mods[1].bias, silu_in=True); i += 1
        stem = mods[i]; i += 1
        xin = x if self.centered else 2 * x - 1.0
        # unfolded operand of the three input convolutions' weight gradients (stem on x, mask_layer on the mask, pos_layer on
        # the coordinates all produce h0 and share its gradient): ONE 16-slot operand [x 0..3 | mask 4 | coords 5..7 | 0]
        # and one md_wgrad launch instead of three 128 x 128-tile launches that each re-read dY nine times
        srcs = [xin, self.mask.detach().expand(B, -1, -1, -1, -1)]
        if self.USE_COORDS:
            srcs.append(self.coords.detach().expand(B, -1, -1, -1, -1))
        x16 = ops.ncdhw_to_s16b(torch.cat(srcs, dim=1).contiguous(), 16)
        h = self._stem_forward(stem, xin, B, R)
        fw, fb, foffs, ftot = self._film_table()
        film = ops.linear(temb, fw, fb, silu_in=True)
        acts, tape = {}, []
        nid = [0]

        def new(t, c, p):
            nid[0] += 1
            acts[nid[0]] = (t, c, p)
            return nid[0]

        def res(block, in_ids):
            parts = [(acts[v][0], acts[v][1]) for v in in_ids]
            p = acts[in_ids[0]][2]
            tp = []
            o = foffs[id(block)]
            y = block.forward_blocked(parts, B, p, temb, bias0=film.view(-1)[o:], bias0_stride=ftot, tape=tp)
            out = new(y, block.out_ch, p)
            tape.append(("res", block, tp[0], in_ids, out))
            return out

        def attn(block, vid):
            t, c, p = acts[vid]
            tp = []
            y = block.forward_blocked(t, B, p, tape=tp)
            out = new(y, c, p)
            tape.append(("attn", block, tp[0], [vid], out))
            return out

        v0 = new(h, self.nf, P)
        hs = [v0]
        for lvl in range(self.num_resolutions):
            for _ in range(self._blocks_at(lvl)):
                v = res(mods[i], [hs[-1]]); i += 1
                if self.all_resolutions[lvl] in self.attn_resolutions:
                    v = attn(mods[i], v); i += 1
                hs.append(v)
            if lvl != self.num_resolutions - 1:
                t, c, p = acts[hs[-1]]
                tp = []
                y = mods[i].forward_blocked(t, c, B, p, tape=tp)
                out = new(y, c, p // 8)
                tape.append(("down", mods[i], tp[0], [hs[-1]], out)); i += 1
                hs.append(out)
        v = hs[-1]
        v = res(mods[i], [v]); i += 1
        v = attn(mods[i], v); i += 1
        v = res(mods[i], [v]); i += 1
        for lvl in reversed(range(self.num_resolutions)):
            for _ in range(self._blocks_at(lvl) + 1):
                v = res(mods[i], [v, hs.pop()]); i += 1
            if self.all_resolutions[lvl] in self.attn_resolutions:
                v = attn(mods[i], v); i += 1
            if lvl != 0:
                t, c, p = acts[v]
                tp = []
                y = mods[i].forward_blocked(t, c, B, p, tape=tp)
                out = new(y, c, p * 8)
                tape.append(("up", mods[i], tp[0], [v], out)); i += 1
                v = out
        assert not hs
        gn = mods[i]; i += 1
        hl, c, p = acts[v]
        prm = ops.gn_params([(hl, c)], gn.weight, gn.bias, B, p, eps=gn.eps, groups=gn.num_groups)
        a = ops.gn_apply([(hl, c)], prm, B, p, norm=True, silu=True)
        head = mods[i]
        out = self._head_forward(head, a, B, R)
        ctx = dict(B=B, R=R, P=P, emb=emb, t1=t1, temb=temb, x16=x16, v0=v0, last=v, acts=acts, tape=tape, gn_prm=prm,
                   a_final=a, foffs=foffs, ftot=ftot, fw=fw)
        return out, ctx

    def _backward(self, ctx, d_eps):
        from . import backward as bw
        from torch.nn import functional as F
        mods = self.all_modules
        B, R, P, acts, tape = ctx["B"], ctx["R"], ctx["P"], ctx["acts"], ctx["tape"]
        dev = d_eps.device
        grads = {}

        def acc(vid, g):
            if vid in grads:
                grads[vid].add_(g)
            else:
                grads[vid] = g

        # head conv + final GroupNorm/SiLU
        d8 = torch.zeros((B, 8, R, R, R), dtype=torch.float32, device=dev)
        d8[:, :self.out_channels] = d_eps
        dy = ops.ncdhw_to_f32b(d8)
        gn, head = mods[-2], mods[-1]
        hl, c, p = acts[ctx["last"]]
        d_a = bw.conv3_backward(self, "head", head, dy, ctx["a_final"], B, R)
        acc(ctx["last"], bw.gn_backward([(hl, c)], d_a, ctx["gn_prm"], gn, B, p, silu=True)[0])
        del d_a
        d_film = torch.zeros((B, ctx["ftot"]), dtype=torch.float32, device=dev)
        hook = self.__dict__.get("_grad_ready_hook")     # parallel.GradReducer.ready during multi-GPU training

        def announce(layer):
            # gradients that are final once this layer's backward ran (Dense_0 / Conv_0.bias of a ResnetBlock
            # are completed by the FiLM algebra at the end)
            if hook is not None:
                late = {id(layer.Dense_0.weight), id(layer.Dense_0.bias), id(layer.Conv_0.bias)} \
                    if isinstance(layer, ResnetBlockDDPM) else set()
                hook([p for p in layer.parameters() if id(p) not in late])

        if hook is not None:
            hook(list(gn.parameters()) + list(head.parameters()))
        for kind, layer, sv, ins, out in reversed(tape):
            g = grads.pop(out)
            if kind == "res":
                dparts, dbias0 = layer.backward_blocked(sv, g)
                o = ctx["foffs"][id(layer)]
                d_film[:, o:o + layer.out_ch] = dbias0
                for vid, dp in zip(ins, dparts):
                    acc(vid, dp)
            else:
                acc(ins[0], layer.backward_blocked(sv, g))
            del g
            sv.clear()          # the layer's saved tensors (Winograd operands T, S16B activations) are dead: free them now
            announce(layer)
        # stem: h0 = conv(x) + pos_layer(coords) + mask_layer(mask) (+ biases)
        g0 = grads.pop(ctx["v0"])
        stem = mods[2]
        bsum = bw.channel_sums(g0, B, self.nf, P).sum(0)     # all three biases receive the same sum of g0
        convs = [(stem, 0, stem.weight.shape[1]), (self.mask_layer, stem.weight.shape[1], 1)]
        if self.USE_COORDS:
            convs.append((self.pos_layer, stem.weight.shape[1] + 1, 3))
        ksz = stem.weight.shape[-1]
        taps, pad = ksz ** 3, ksz // 2
        dw = torch.zeros((self.nf, 16, taps), dtype=torch.float32, device=g0.device)
        dy_pb = bw.to_pb16(g0, B, self.nf, R, 0, pad=pad, zhalo=False)
        act_pb = bw.to_pb16(ctx["x16"], B, 16, R, 1, pad=pad)
        bw.wgrad(dy_pb, act_pb, B, self.nf, 16, R, taps, dw, 16 * taps, taps, 1)
        del dy_pb, act_pb
        for conv, c0, c in convs:
            bw._grad_of(conv.weight).add_(dw[:, c0:c0 + c].reshape(conv.weight.shape))
            bw._grad_of(conv.bias).add_(bsum[:self.nf])
        del g0
        # FiLM table + timestep MLP (tiny [B,512] algebra: torch ops on the device)
        temb, t1, emb = ctx["temb"], ctx["t1"], ctx["emb"]
        s2 = F.silu(temb)
        d_fw = d_film.t() @ s2
        d_fb = d_film.sum(0)
        for blk, o in ((m, ctx["foffs"][id(m)]) for m in mods if isinstance(m, ResnetBlockDDPM)):
            bw._grad_of(blk.Dense_0.weight).add_(d_fw[o:o + blk.out_ch])
            bw._grad_of(blk.Dense_0.bias).add_(d_fb[o:o + blk.out_ch])
            bw._grad_of(blk.Conv_0.bias).add_(d_fb[o:o + blk.out_ch])

        def dsilu(z):
            sg = torch.sigmoid(z)
            return sg * (1 + z * (1 - sg))

        d_temb = (d_film @ ctx["fw"]) * dsilu(temb)
        bw._grad_of(mods[1].weight).add_(d_temb.t() @ F.silu(t1))
        bw._grad_of(mods[1].bias).add_(d_temb.sum(0))
        d_t1 = (d_temb @ mods[1].weight.detach()) * dsilu(t1)
        bw._grad_of(mods[0].weight).add_(d_t1.t() @ emb)
        bw._grad_of(mods[0].bias).add_(d_t1.sum(0))
        ops.bump_param_epoch()   # nothing cached depends on grads, but keep caches honest if an optimizer steps next

    def grad_completion_order(self):
        """Trainable parameters in the order `backward` finishes (and announces) their gradients: final GroupNorm + head,
        then the layers of the tape in reverse; last the ones that are only complete when the backward returns (every
        ResnetBlock's Dense_0 / Conv_0.bias -- FiLM algebra --, the timestep MLP, stem, mask_layer, pos_layer).  This is
        the layout of parallel.FlatGrads, so that finished gradients are a contiguous prefix of the flat buffer."""
        mods = list(self.all_modules)
        order, late, seen = [], [], set()

        def put(dst, ps):
            for p in ps:
                if p.requires_grad and id(p) not in seen:
                    seen.add(id(p))
                    dst.append(p)

        put(order, list(mods[-2].parameters()) + list(mods[-1].parameters()))
        for m in reversed(mods[3:-2]):
            if isinstance(m, ResnetBlockDDPM):
                put(late, [m.Dense_0.weight, m.Dense_0.bias, m.Conv_0.bias])
            put(order, m.parameters())
        unused = set() if self.USE_COORDS else {id(p) for p in self.pos_layer.parameters()}   # ddpm_res128 never runs pos_layer
        put(late, [p for p in self.parameters() if id(p) not in unused])
        return order + late

    def _stem_cfg(self):
        return ops.CFG_C3_128_K16 if self.KSIZE == 3 else ops.CFG_C5_128_K16

    def _head_cfg(self):
        return ops.CFG_C3_32 if self.KSIZE == 3 else ops.CFG_C5_32_K16

    def _stem_forward(self, stem, xin, B, R):
        """The k^3 stem conv from 4 channels, dx-folded the other way round: the k x-shifted copies of the input become
        extra input channels (K = 12 or 20 instead of 4 per tap) of a k x k x 1 conv, so the matrix cores do k times
        fewer, fuller K steps.  pos_layer / mask_layer outputs (input independent) come in as the residual."""
        k = self.KSIZE
        cfg, c_pad = (ops.CFG_C3X_128_K16, 16) if k == 3 else (ops.CFG_C5X_128, 32)

        def build():
            w = stem.weight.detach()                                   # [co][ci][kz][ky][kx]
            w2 = w.permute(0, 1, 4, 2, 3).reshape(w.shape[0], w.shape[1] * k, k, k, 1).contiguous()
            return ops.PackedWeight(w2, "conv", cfg, w.device)
        pw = self._cached(f"stem_fold{cfg}", [stem.weight], build)
        xf = ops.ncdhw_to_s16b_xfold(xin, k, c_pad)
        if k == 3 and ops.conv3_stem_ok(pw.rows, pw.kdim, R):
            # dedicated kernel; the GroupNorm sums of h0 (first ResnetBlock) come from its epilogue
            stats = ops.stats_zeros(B, pw.rows, xin.device) if ops.FUSE_GN_STATS else None
            out = ops.conv3_stem(pw, xf, B, R, bias=stem.bias, residual=self._stem_const(), stats=stats)
            if stats is not None:
                out._md_sums = stats
            return out
        return layers.run_conv3(pw, xf, B, R, bias=stem.bias, residual=self._stem_const(), res_bstride=0)

    def _head_forward_fused(self, head, h, ac, B, R):
        """_head_forward on the un-normalised tensor `h` (F32B) with the folded GroupNorm affine `ac`: md_conv3_head."""
        k, co = self.KSIZE, self.out_channels

        def build():
            w = head.weight.detach()                                   # [co][ci][kz][ky][kx]
            w2 = w.permute(0, 4, 1, 2, 3).reshape(co * k, w.shape[1], k, k, 1).contiguous()
            return ops.PackedWeight(w2, "conv", ops.CFG_HEAD_PACK, w.device)
        pw = self._cached("head_fold_fused", [head.weight], build)
        rows_alloc = ((co * k + 7) // 8) * 8
        y = ops.conv3_head(pw, h, ac, B, R, rows_alloc)
        return ops.fold_dx(y, head.bias, B, co, k, rows_alloc, R)

    def _head_forward(self, head, a, B, R):
        """The k^3 head conv to 4 channels, dx-folded: a k x k x 1 conv whose rows are the (co, dx) pairs (12 or 20 of the
        32 rows of an MFMA tile instead of 4, with k times fewer taps), then `md_fold_dx` adds the k x-shifted columns
        and the bias and writes NCDHW.  Same products as the reference conv, summed in a different order."""
        k, co = self.KSIZE, self.out_channels
        cfg = ops.CFG_C3X_32 if k == 3 else ops.CFG_C5X_32_K16

        def build():
            w = head.weight.detach()                                   # [co][ci][kz][ky][kx]
            w2 = w.permute(0, 4, 1, 2, 3).reshape(co * k, w.shape[1], k, k, 1).contiguous()
            return ops.PackedWeight(w2, "conv", cfg, w.device)
        pw = self._cached(f"head_fold{cfg}", [head.weight], build)
        rows_alloc = ((co * k + 7) // 8) * 8
        y = layers.run_conv3(pw, a, B, R, rows_alloc=rows_alloc)
        return ops.fold_dx(y, head.bias, B, co, k, rows_alloc, R)

    # ---- forward ------------------------------------------------------------------------------
    def forward(self, x, labels):
        if not x.is_cuda:
            raise RuntimeError(f"{type(self).__name__} (meshdiffusion_amd) runs on the GPU only: no CPU fallback")
        if torch.is_grad_enabled() and self.training:
            # training: the HIP forward records a tape and the HIP backward accumulates parameter .grad; torch
            # autograd only sees one opaque node (so `loss.backward()` of the reference's step_fn works)
            return _UNetTrainFn.apply(x, labels, self, self._autograd_anchor())
        with ops.precision_scope(self.hip_precision):
            return self._forward_eval(x, labels)

    def calibrate(self, x, labels, bar=4e-5):
        """Load-time calibration of the reduced-precision conv arithmetic on THESE weights (call it once after `load_state_dict` /
        `restore_checkpoint`, in eval mode; the trained checkpoints are external downloads, README.md:35-37 of the reference, so the
        first real one meets the f16f8 / f16f6 formats unseen).  x, labels: one batch, or lists of batches (e.g. a noise batch at a
        few timesteps).  Two steps, both on the model's own inference path:
          1. one evaluation per batch with `hip_ops.CALIBRATE` on: every Winograd conv records the per-channel mean squares of the
             operand it reads; each layer keeps the mean over the batches (`layer._md_act_ms[site]`), from which its equaliser is
             rebuilt (hip_ops.wino_equaliser(a2m=...)): GroupNorm normalises groups of channels, the static estimate assumes unit
             variance per CHANNEL; and the Upsample convs, which read the raw residual stream, get an equaliser at all;
          2. one audited evaluation per batch (`hip_ops.AUDIT`: every reduced-precision launch repeated in bf16x3 on the same operand):
             a conv whose relative difference exceeds `bar` on any batch is taken off the reduced-precision path for good
             (`layer.md_bf16x3_sites`).
        Returns {"measured": n convs, "audited": n launches, "worst": largest difference kept, "demoted": [(module name, site, diff)]}."""
        xs, ls = (list(x), list(labels)) if isinstance(x, (list, tuple)) else ([x], [labels])
        assert len(xs) == len(ls) and not self.training
        names = {id(m): n for n, m in self.named_modules()}
        ops.CALIBRATE = {}
        try:
            with torch.no_grad():
                for xi, li in zip(xs, ls):
                    self.forward(xi, li)
            cal = ops.CALIBRATE
        finally:
            ops.CALIBRATE = None
        for owner, site, tot, n in cal.values():
            owner.__dict__.setdefault("_md_act_ms", {})[site] = (tot / float(n)).contiguous()
        worst = {}
        n_audited = 0
        for xi, li in zip(xs, ls):
            ops.AUDIT = []
            try:
                with torch.no_grad():
                    self.forward(xi, li)
                recs = ops.AUDIT
            finally:
                ops.AUDIT = None
            n_audited += len(recs)
            for r in recs:
                key = (id(r["owner"]), r["site"])
                if key not in worst or r["rel_l2"] > worst[key]["rel_l2"]:
                    worst[key] = r
        demoted = []
        for r in worst.values():
            if r["rel_l2"] > bar:
                r["owner"].md_bf16x3_sites = tuple(sorted(set(getattr(r["owner"], "md_bf16x3_sites", ())) | {r["site"]}))
                demoted.append((names.get(id(r["owner"]), "?"), r["site"], r["rel_l2"]))
        kept = [r["rel_l2"] for r in worst.values() if r["rel_l2"] <= bar]
        return dict(measured=len(cal), audited=n_audited, worst=max(kept) if kept else 0.0, demoted=sorted(demoted))

    def _forward_eval(self, x, labels):
        mods = self.all_modules
        B, R = x.shape[0], self.img_size
        P = R ** 3
        assert tuple(x.shape[1:]) == (self.out_channels, R, R, R)
        ops.stats_arena_reset(x.device)      # one fill for every GroupNorm sum buffer of this evaluation
        i = 0
        temb = layers.get_timestep_embedding(labels, self.nf)
        temb = ops.linear(temb, mods[i].weight, mods[i].bias); i += 1
        temb = ops.linear(temb, mods[i].weight, mods[i].bias, silu_in=True); i += 1

        h_in = x if self.centered else 2 * x - 1.0
        stem = mods[i]; i += 1
        h = self._stem_forward(stem, h_in, B, R)

        fw, fb, foffs, ftot = self._film_table()
        film = ops.linear(temb, fw, fb, silu_in=True)            # [B, sum(out_ch)]

        def res(block, parts, p):
            o = foffs[id(block)]
            return block.forward_blocked(parts, B, p, temb, bias0=film.view(-1)[o:], bias0_stride=ftot)

        hs = [(h, self.nf, P)]
        for lvl in range(self.num_resolutions):
            for _ in range(self._blocks_at(lvl)):
                t, c, p = hs[-1]
                h = res(mods[i], [(t, c)], p); c = mods[i].out_ch; i += 1
                if self.all_resolutions[lvl] in self.attn_resolutions:
                    h = mods[i].forward_blocked(h, B, p); i += 1
                hs.append((h, c, p))
            if lvl != self.num_resolutions - 1:
                t, c, p = hs[-1]
                hs.append((mods[i].forward_blocked(t, c, B, p), c, p // 8)); i += 1

        h, c, p = hs[-1]
        h = res(mods[i], [(h, c)], p); i += 1
        h = mods[i].forward_blocked(h, B, p); i += 1
        h = res(mods[i], [(h, c)], p); i += 1

        for lvl in reversed(range(self.num_resolutions)):
            for _ in range(self._blocks_at(lvl) + 1):
                st, sc, sp = hs.pop()
                assert sp == p
                h = res(mods[i], [(h, c), (st, sc)], p); c = mods[i].out_ch; i += 1
            if self.all_resolutions[lvl] in self.attn_resolutions:
                h = mods[i].forward_blocked(h, B, p); i += 1
            if lvl != 0:
                h = mods[i].forward_blocked(h, c, B, p); p *= 8; i += 1
        assert not hs

        gn = mods[i]; i += 1
        head = mods[i]; i += 1
        assert i == len(mods)
        if self.KSIZE == 3 and ops.conv3_head_ok(self.out_channels * 3, c, R):
            # GroupNorm affine + SiLU + split inside the head kernel's loader: no GroupNorm-apply pass over the 64^3 tensor
            _, ac = ops.gn_params([(h, c)], gn.weight, gn.bias, B, p, eps=gn.eps, groups=gn.num_groups, want_ac=True)
            out = self._head_forward_fused(head, h, ac, B, R)
        else:
            prm = ops.gn_params([(h, c)], gn.weight, gn.bias, B, p, eps=gn.eps, groups=gn.num_groups)
            a = ops.gn_apply([(h, c)], prm, B, p, norm=True, silu=True)
            out = self._head_forward(head, a, B, R)

        if self.scale_by_sigma:
            out = out / self.sigmas[labels.long(), None, None, None, None].to(out.dtype)
        return out


@utils.register_model(name="ddpm_res64")
class DDPMRes64(DDPMUNet3D):
    """lib/diffusion/models/ddpm_res64.py: 3x3x3 stem/head, `coords` positional input."""
    KSIZE, USE_COORDS, LEVEL0_BLOCKS = 3, True, None


class _UNetTrainFn(torch.autograd.Function):
    """One opaque autograd node around the HIP forward/backward.  Parameter gradients are accumulated straight
    into `.grad` by `DDPMUNet3D.backward`; the `anchor` input only makes autograd schedule this node.
    The gradient w.r.t. the input `x` is NOT produced (the DDPM loss never needs it): `x.grad` stays None."""

    @staticmethod
    def forward(ctx, x, labels, model, anchor):
        with torch.no_grad():
            out, tape_ctx = model.forward_train(x, labels)
        ctx.model, ctx.tape_ctx = model, tape_ctx
        return out

    @staticmethod
    def backward(ctx, d_eps):
        with torch.no_grad():
            ctx.model.backward(ctx.tape_ctx, d_eps.contiguous())
        ctx.tape_ctx = None
        return None, None, None, None
