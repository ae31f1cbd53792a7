"""Microbenchmark of md_conv3_wino (and md_wino_prep) on the hot shapes: production schedule, the A/B schedule and the
timing-only ablations (library built with MD_BUILD_ABLATIONS=1).  HIP events on the launch stream, median of `--reps`.

    MD_BUILD_ABLATIONS=1 python -m meshdiffusion_amd.build --force && python tools/bench_wino.py
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meshdiffusion_amd import hip_ops as ops  # noqa: E402

VARIANTS = {0: "production", 2: "abl: halo traffic in the prologue only (real data)", 4: "abl: weight loads in the prologue only (real data)",
            6: "abl: halo + weights in the prologue only (real data)", 22: "abl: 6 + no epilogue", 1: "abl: no halo traffic (constant operands: clocks higher)", 9: "abl: no halo, no LDS reads",
            16: "abl: no epilogue", 25: "abl: MFMA loop + weight loads only, no epilogue",
            32: "abl: halo from a private L2-resident 30 KB", 64: "abl: halo as a private contiguous HBM stream"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--variants", default="0,1,9,16,25,32,64")
    ap.add_argument("--shapes", default="128:128:64:8,256:128:64:8,256:256:32:8,512:256:16:8")
    ap.add_argument("--out", default=None)
    ap.add_argument("--prep-v2", action="store_true", help="also time the experimental two-phase operand pass")
    ap.add_argument("--f8", action="store_true", help="also time the f16f8 arithmetic (md_wino_prep_f8 + md_conv3_wino_f8), interleaved with "
                                                       "the bf16x3 production kernel (same box, same clocks)")
    ap.add_argument("--no-stats", action="store_true", help="launch without the GroupNorm-sum epilogue")
    ap.add_argument("--no-res", action="store_true", help="launch without the residual operand")
    ap.add_argument("--zero-data", action="store_true", help="all-zero activations and weights: same instruction stream and duty, far less switching power -- the control "
                                                               "experiment for the power-bound reading of the stamps (shader clock rises, cycles per workgroup do not change)")
    ap.add_argument("--dump", default=None, help="with --stamps: save the raw stamp array of every shape to <dump>_<shape>.npy")
    ap.add_argument("--stamps", action="store_true", help="variant 128 (MD_BUILD_ABLATIONS=1): per-wave s_memtime stamps of the first 1024 workgroups")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    rows = []
    for sh in a.shapes.split(","):
        cin, cout, S, B = [int(v) for v in sh.split(":")]
        g = torch.Generator().manual_seed(1)
        x = ops.ncdhw_to_f32b(torch.randn((B, cin, S, S, S), generator=g).to(dev))
        w = (torch.randn((cout, cin, 3, 3, 3), generator=g) * 0.03).to(dev)
        if a.zero_data:
            x.zero_(); w.zero_()
        bias = torch.randn((B, cout), generator=g).to(dev)
        res = ops.f32b_empty(B, cout, S ** 3, dev).normal_()
        ac = torch.stack([1.0 + 0.1 * torch.randn((B, cin), generator=g), 0.1 * torch.randn((B, cin), generator=g)], -1).contiguous().to(dev)
        ww = ops.WinoWeight(w, dev)
        out = ops.f32b_empty(B, cout, S ** 3, dev)
        stats = torch.zeros((B, cout, 2), dtype=torch.float64, device=dev)
        flops = 2.0 * B * cout * cin * 27 * S ** 3

        def timed(fn):
            ts = []
            for _ in range(a.reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            return sorted(ts)[len(ts) // 2]

        ops.WINO_PREP_V2 = False
        t = ops.wino_prep([(x, cin)], ac, True, False, B, S)
        ms_prep = timed(lambda: ops.wino_prep([(x, cin)], ac, True, False, B, S))
        rows.append(dict(shape=sh, kernel="md_wino_prep", ms=round(ms_prep, 4), gbs=round(12.0 * B * cin * S ** 3 / ms_prep / 1e6, 1)))
        print(json.dumps(rows[-1]), flush=True)
        if a.prep_v2 and 256 % S == 0:
            ops.WINO_PREP_V2 = True
            ms2 = timed(lambda: ops.wino_prep([(x, cin)], ac, True, False, B, S))
            ops.WINO_PREP_V2 = False
            rows.append(dict(shape=sh, kernel="md_wino_prep_v2", ms=round(ms2, 4), gbs=round(12.0 * B * cin * S ** 3 / ms2 / 1e6, 1)))
            print(json.dumps(rows[-1]), flush=True)
        if a.f8 and 256 % S == 0:
            ww8, ww6 = ops.WinoWeightF8(w, dev), ops.WinoWeightF8(w, dev, "f6")
            kw = dict(bias=bias, bias_bstride=cout, residual=None if a.no_res else res, res_bstride=0 if a.no_res else cout * S ** 3,
                      stats=None if a.no_stats else stats, out=out)
            for rep in range(3):           # a, b, a, b, a, b
                ops.WINO_PREP_V2 = True
                t = ops.wino_prep([(x, cin)], ac, True, False, B, S)
                ms_p = timed(lambda: ops.wino_prep([(x, cin)], ac, True, False, B, S))
                ms_a = timed(lambda: ops.conv3_wino(ww, t, B, S, variant=0, **kw))
                t8 = ops.wino_prep([(x, cin)], ac, True, False, B, S, f8=True)
                ms_p8 = timed(lambda: ops.wino_prep([(x, cin)], ac, True, False, B, S, f8=True))
                ms_b = timed(lambda: ops.conv3_wino(ww8, t8, B, S, **kw))
                t6 = ops.wino_prep([(x, cin)], ac, True, False, B, S, f8="f6")      # the f16f6 operand (same scratch buffer)
                ms_p6 = timed(lambda: ops.wino_prep([(x, cin)], ac, True, False, B, S, f8="f6"))
                ms_c = timed(lambda: ops.conv3_wino(ww6, t6, B, S, **kw))
                ops.WINO_PREP_V2 = False
                rows.append(dict(shape=sh, kernel="A/B bf16x3 vs f16f8", rep=rep, bf16x3_ms=round(ms_a, 4), f16f8_ms=round(ms_b, 4),
                                 ratio=round(ms_b / ms_a, 4), f16f6_ms=round(ms_c, 4), f6_over_f8=round(ms_c / ms_b, 4),
                                 prep_ms=round(ms_p, 4), prep_f8_ms=round(ms_p8, 4), prep_f6_ms=round(ms_p6, 4),
                                 bf16x3_tflops_alg=round(flops / ms_a / 1e9, 1), f16f8_tflops_alg=round(flops / ms_b / 1e9, 1)))
                print(json.dumps(rows[-1]), flush=True)
            t = ops.wino_prep([(x, cin)], ac, True, False, B, S)      # restore the bf16 operand in the shared scratch buffer
        for v in [int(k) for k in a.variants.split(",")]:
            try:
                ms = timed(lambda: ops.conv3_wino(ww, t, B, S, bias=bias, bias_bstride=cout, residual=None if a.no_res else res,
                                                  res_bstride=0 if a.no_res else cout * S ** 3,
                                                  stats=None if a.no_stats else stats, out=out, variant=v))
            except Exception as e:  # variant not built
                print(f"variant {v}: {e}", flush=True)
                continue
            rows.append(dict(shape=sh, kernel="md_conv3_wino", variant=v, what=VARIANTS.get(v, "?"), ms=round(ms, 4),
                             tflops_alg=round(flops / ms / 1e9, 1), issued_frac_of_peak=round(flops * 2 / 3 * 3 / ms / 1e9 / 2500.0, 4)))
            print(json.dumps(rows[-1]), flush=True)
        if a.stamps:
            dbg = torch.zeros((1024, 4, 12), dtype=torch.int64, device=dev)
            for _ in range(2):
                if a.f8:      # a library built with MD_EXTRA_DEFINES=-DW8_STAMPS: md_conv3_wino_f8 runs its stamping instantiation
                    t8 = ops.wino_prep([(x, cin)], ac, True, False, B, S, f8=True)
                    ops.conv3_wino(ops.WinoWeightF8(w, dev), t8, B, S, bias=bias, bias_bstride=cout, residual=None if a.no_res else res,
                                   res_bstride=0 if a.no_res else cout * S ** 3, stats=dbg, out=out)
                    continue
                ops.conv3_wino(ww, t, B, S, bias=bias, bias_bstride=cout, residual=None if a.no_res else res,
                               res_bstride=0 if a.no_res else cout * S ** 3, stats=dbg, out=out, variant=128)
            torch.cuda.synchronize()
            if a.dump:
                import numpy as np
                np.save(f"{a.dump}_{sh.replace(':', '_')}.npy", dbg.cpu().numpy())
            st = dbg.cpu().double()
            st = st[st[:, 0, 0] > 0]
            tt = st[:, :, :9]
            names = ["start->first fragments (prologue)", "main loop", "loop end->barrier (wait for the slowest wave)",
                     "round 0", "round 1", "round 2", "round 3", "after the last round"]
            d = tt[:, :, 1:] - tt[:, :, :-1]
            wg = tt[:, :, 8].max(1).values - tt[:, :, 0].min(1).values
            # s_memtime ticks: report raw and relative to the workgroup's whole duration
            rep = dict(shape=sh, kernel="md_conv3_wino stamps", n_wg=int(st.shape[0]), wg_ticks_mean=round(float(wg.mean()), 1),
                       phases={n: round(float(d[:, :, k].mean()), 1) for k, n in enumerate(names)},
                       phases_frac={n: round(float(d[:, :, k].mean() / wg.mean()), 4) for k, n in enumerate(names)},
                       loop_end_skew_ticks=round(float((tt[:, :, 2].max(1).values - tt[:, :, 2].min(1).values).mean()), 1),
                       first_gen_start_spread=round(float(tt[:256, :, 0].max() - tt[:256, :, 0].min()), 1))
            # effective shader clock of a workgroup: s_memtime ticks per s_memrealtime tick (100 MHz); gap between consecutive workgroups of a CU in real time
            rt0, rt1 = st[:, :, 10].min(1).values, st[:, :, 11].max(1).values
            rep["wg_real_us_mean"] = round(float((rt1 - rt0).mean()) / 100.0, 2)
            rep["shader_clock_ghz"] = round(float((wg / (rt1 - rt0).clamp_min(1)).mean()) * 0.1, 3)
            hw = dbg.cpu()[: st.shape[0], 0, 9]
            cu_real = {}
            for i in range(st.shape[0]):
                cu_real.setdefault(int(hw[i]) & ~0xF, []).append((float(rt0[i]), float(rt1[i])))   # same (XCC, SE, CU) id; wave slot masked
            g2 = []
            for v in cu_real.values():
                v.sort()
                g2 += [v[k + 1][0] - v[k][1] for k in range(len(v) - 1) if v[k + 1][0] - v[k][1] < 2000]      # < 20 us: really the next workgroup
            if g2:
                rep["real_gap_us_between_consecutive_wgs"] = round(sum(g2) / len(g2) / 100.0, 2)
                rep["n_gaps"] = len(g2)
            # second generation: gap between a workgroup's end and the next start on the same CU (hw id + xcc id)
            ids = dbg.cpu()[:, 0, 9][: st.shape[0]]
            cu = {}
            for i in range(st.shape[0]):
                cu.setdefault(int(ids[i]) & ~0xF, []).append((float(tt[i, :, 0].min()), float(tt[i, :, 8].max())))
            gaps = []
            for v in cu.values():
                v.sort()
                gaps += [v[k + 1][0] - v[k][1] for k in range(len(v) - 1)]
            if gaps:
                rep["gap_between_workgroups_on_a_cu_ticks"] = round(sum(gaps) / len(gaps), 1)
                rep["n_cu_ids"] = len(cu)
            rows.append(rep)
            print(json.dumps(rep), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
