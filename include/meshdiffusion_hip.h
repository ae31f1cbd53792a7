/*
 * meshdiffusion_hip.h -- C ABI of libmeshdiffusion_hip.so (gfx950 / MI355X).
 *
 * This is the drop-in boundary underneath MeshDiffusion's Python score-model
 * API.  The reference has NO C/FFI boundary on this path: its
 * "kernels" are ATen calls made from Python modules.  Each entry point below
 * therefore cites the reference Python call site it replaces
 * (paths relative to the reference repo root).
 *
 * Conventions
 *   - plain pointers + sizes only; no torch types.  All pointers are DEVICE
 *     pointers owned by the caller (PyTorch allocator); the library never
 *     allocates, frees or retains device memory.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*),
 *     re-entrant and stateless.
 *   - return value: 0 = ok, negative = MD_ERR_*, positive = hipError_t.
 *
 * Device tensor layouts (see DESIGN.md "Data layout in HBM")
 *   NCDHW : float32 [B][C][P]                      (the reference's layout, P = D*H*W)
 *   F32B  : float32 [B][C/8][P][8]                 (8-channel blocked fp32)
 *   S16B  : bf16    [B][C/8][2][P][8]              (8-channel blocked split-bf16:
 *                                                   plane 0 = hi = bf16(x),
 *                                                   plane 1 = lo = bf16(x - hi))
 *   WPK   : bf16    [rows/NT][K/KC][taps][KC/8][2][NT][8]  packed split-bf16
 *                                                   weight tiles (one tile = one LDS image)
 *   PB16  : bf16    [guard + Pp + guard][ceil(B/8)][2][C][8 samples]   position-major, sample-blocked
 *                                                   split-bf16 on the zero-padded grid (Pp positions)
 *                                                   with zeroed guard positions: operands of md_wgrad
 */
#ifndef MESHDIFFUSION_HIP_H
#define MESHDIFFUSION_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MD_OK 0
#define MD_ERR_BAD_ARG (-1)
#define MD_ERR_UNSUPPORTED (-2)
#define MD_ERR_NO_DEVICE (-3)

#define MD_ABI_VERSION 15

/* ---- tile configurations of md_gemm_conv (compile-time instantiations) ---- */
enum {
  MD_CFG_C3_128 = 0,     /* 3x3x3 s1, tile 4x8x8,  NT=128, KC=32  (main conv)        */
  MD_CFG_C3_128_K16 = 1, /* 3x3x3 s1, tile 4x8x8,  NT=128, KC=16  (stem, Cin<=16)    */
  MD_CFG_C3_32 = 2,      /* 3x3x3 s1, tile 4x8x8,  NT=32,  KC=32  (head, Cout<=32)   */
  MD_CFG_C3_LOW = 3,     /* 3x3x3 s1, tile 4x4x4,  NT=128, KC=32  (4^3 level)        */
  MD_CFG_C3_S2 = 4,      /* 3x3x3 s2 pad(0,1), tile 4x4x4, NT=128, KC=32 (Downsample)*/
  MD_CFG_G1_128 = 5,     /* 1x1x1 / GEMM, 256 cols, NT=128, KC=32                    */
  MD_CFG_G1_128_LOW = 6, /* 1x1x1 / GEMM, 64 cols,  NT=128, KC=32                    */
  MD_CFG_G1_64_LOW = 7,  /* 1x1x1 / GEMM, 64 cols,  NT=64,  KC=32                    */
  MD_CFG_C3_128_V2 = 8,  /* C3_128 + conflict-free halo layout + software-pipelined loads     */
  MD_CFG_C3_128_SW = 9,  /* (A/B) conflict-free halo layout only                              */
  MD_CFG_C3_128_PIPE = 10, /* (A/B) pipelined loads only                                       */
  /* 11..13: reserved (ids of retired experiments; md_gemm_conv answers MD_ERR_UNSUPPORTED)                  */
  MD_CFG_C3_128_FAST = 14, /* dedicated kernel for the hot conv: C3_128_V2 layout, taps unrolled, F32B out */
  MD_CFG_C5_128_K16 = 15, /* 5x5x5 s1 pad 2, tile 4x8x8, NT=128, KC=16 (ddpm_res128 stem / mask_layer)  */
  MD_CFG_C5_32_K16 = 16,  /* 5x5x5 s1 pad 2, tile 4x8x8, NT=32,  KC=16 (ddpm_res128 head)               */
  MD_CFG_C3_128_W4 = 17,  /* experiment: 3x3x3, tile 4x4x8, NT=128, 4 waves (2 workgroups/CU)          */
  MD_CFG_C3X_32 = 18,     /* 3x3x1 taps (one dx column of a 3x3x3 kernel), tile 4x8x8, NT=32, KC=32: dx-folded head */
  MD_CFG_C5X_32_K16 = 19, /* 5x5x1 taps, NT=32, KC=16: dx-folded 5x5x5 head of ddpm_res128                        */
  MD_CFG_C3X_128_K16 = 20, /* 3x3x1 taps, NT=128, KC=16: dx-folded 3x3x3 stem (K = 4 channels x 3 dx)                    */
  MD_CFG_C5X_128 = 21,     /* 5x5x1 taps, NT=128, KC=32: dx-folded 5x5x5 stem of ddpm_res128 (K = 4 channels x 5 dx)     */
  MD_CFG_G1_128_N128 = 22, /* 1x1x1 / GEMM, 128 cols, NT=128, KC=32: three workgroups per CU (HBM-bound shortcut NINs)         */
  MD_CFG_COUNT = 23
};

enum { MD_OUT_F32B = 0, MD_OUT_S16B = 1, MD_OUT_NCDHW = 2 };
enum { MD_A_PACKED = 0, MD_A_S16B = 1 };
enum { MD_B_S16B = 0, MD_B_F32B_GN = 1 };
/* MD_PREC_BF16X3: both operands split bf16 (hi, lo), 3 MFMAs per product (default, ~1e-5 per U-Net eval).
 * MD_PREC_FP16X2: weights split fp16 (hi, lo), activations ONE fp16 (plane 0 only), 2 MFMAs per product
 *                 (~1e-3 per eval, 7e-5 after 999 sampler steps; MD_CFG_C3_128_FAST only). */
enum { MD_PREC_BF16X3 = 0, MD_PREC_FP16X2 = 1 };

/*
 * md_gemm_conv: out[b][i][j] = alpha * sum_{tap,k} A[i][tap][k] * B[b][k][pos_j + tap]
 *                              + bias[b*bias_bstride + i] + residual[b][i][j]
 * computed on the matrix cores with a bf16x3 split (hi*hi + hi*lo + lo*hi,
 * fp32 accumulate).  i = output row (output channel), j = spatial position.
 * With ksplit > 1 (F32B output only) the K-chunk range is divided over grid.z workgroups that write
 * raw partial sums to `partial`; a second, deterministic kernel adds the slices in order and applies
 * alpha / bias / residual.  Used for the 4^3 and 8^3 levels where tiles alone cannot fill 256 CUs.
 *
 * Replaces: nn.Conv3d 3x3x3 (lib/diffusion/models/layers.py:118-124 used at
 * :654,:662, ddpm_res64.py:85-87,:121), Downsample pad+stride-2 conv
 * (layers.py:626-643), Upsample nearest x2 + conv (layers.py:611-623, `ups`=1
 * folds the interpolate into the halo load), NIN 1x1x1 (layers.py:573-582)
 * and the two attention einsums (layers.py:602,606; a_src = MD_A_S16B).
 */
typedef struct MdGemmConvArgs {
  const void* a;         /* WPK weights (a_src=0) or S16B tensor [B][K/8][2][a_rows][8] (a_src=1) */
  const void* b;         /* S16B activations [B][K/8][2][P_in][8]                               */
  void* out;             /* F32B / S16B [B][rows_alloc/8].. / NCDHW [B][rows][P]                */
  const float* bias;     /* may be NULL                                                        */
  const float* residual; /* F32B [.][rows_alloc/8][P][8], may be NULL                          */
  float alpha;
  int32_t cfg;           /* MD_CFG_*                                                           */
  int32_t batch;
  int32_t rows;          /* logical output rows (Cout); stores are guarded by it               */
  int32_t rows_alloc;    /* channel count of out/residual tensors (multiple of 8)              */
  int32_t kdim;          /* K per tap (Cin), multiple of the cfg's KC                          */
  int32_t D, H, W;       /* OUTPUT spatial dims; GEMM cfgs use D=H=1, W=P                      */
  int32_t ups;           /* 1: input is nearest-upsampled x2 on the fly                        */
  int32_t a_src;         /* MD_A_*                                                             */
  int32_t out_mode;      /* MD_OUT_*                                                           */
  int32_t a_rows;        /* a_src=1: rows (P) of the A tensor                                  */
  int64_t a_bstride;     /* a_src=1: elements (bf16) between batches of A; 0 = shared          */
  int64_t bias_bstride;  /* floats between batches of bias; 0 = shared                         */
  int64_t res_bstride;   /* floats between batches of residual; 0 = shared                     */
  int64_t b_bstride;     /* elements (bf16) between batches of B; 0 = shared                   */
  float* partial;        /* ksplit>1: fp32 workspace [ksplit][B][rows_alloc/8][P][8]            */
  int32_t ksplit;        /* split the K-chunk loop over this many workgroups (grid.z); 0/1 = off */
  int32_t prec;          /* MD_PREC_*: operand format of A (weights) and B (activations)        */
  double* stats;         /* optional [B][rows_alloc][2]: per-(sample, output channel) sum and sum of   */
                         /* squares of the values written to `out`, ADDED with fp64 atomics (caller    */
                         /* zeroes) -- the GroupNorm statistics of the consumer without another pass   */
                         /* over the tensor.  MD_CFG_C3_128_FAST, F32B output, no split-K; else error  */
  int32_t stagger;       /* MD_CFG_C3_128_FAST: shader cycles over which the start of the first workgroup */
                         /* of each CU is spread (0 = off); measured neutral on MI355X, kept as A/B switch */
  int32_t b_mode;        /* MD_B_S16B (0) or MD_B_F32B_GN (MD_CFG_C3_128_FAST, MD_PREC_BF16X3 only): `b` (and  */
                         /* `b2`) are fp32 F32B tensors [B][C/8][P_in][8]; the kernel applies                 */
                         /* y = x*ac[c][0] + ac[c][1] (GroupNorm affine), SiLU (b_silu) and the bf16 hi/lo    */
                         /* split while it stages the halo tile in LDS -- nn.GroupNorm + nn.SiLU              */
                         /* (layers.py:676-681) without a pass of their own; positions outside the grid are   */
                         /* zero AFTER the transform (the conv pads the activated tensor)                     */
  const void* b2;        /* MD_B_F32B_GN: second part of a channel-concatenated input (torch.cat([h, skip], 1), */
                         /* ddpm_res64.py:174-176): K channels [b_split, kdim); NULL when b_split >= kdim    */
  const float* b_ac;     /* MD_B_F32B_GN: [B][kdim][2] = (rstd*gamma, beta - mean*rstd*gamma) from            */
                         /* md_gn_finalize, or NULL: no affine, no SiLU (plain split: Upsample's conv)        */
  int64_t b2_bstride;    /* floats between batches of b2 (b_bstride: floats between batches of b in this mode) */
  int32_t b_split;       /* channels taken from `b` (multiple of 8)                                           */
  int32_t b_silu;        /* apply SiLU after the affine                                                      */
} MdGemmConvArgs;

int md_abi_version(void);
/* number of visible HIP devices, or negative MD_ERR */
int md_device_count(void);

int md_gemm_conv(const MdGemmConvArgs* args, void* stream);
/* bytes of `partial` workspace md_gemm_conv needs for these args (0 when ksplit <= 1) */
int64_t md_gemm_conv_partial_bytes(const MdGemmConvArgs* args);
/* bytes of LDS and threads per workgroup of a cfg (for DESIGN/bench reporting) */
int md_gemm_conv_cfg_info(int32_t cfg, int32_t* nt, int32_t* kc, int32_t* cols,
                          int32_t* taps, int32_t* lds_bytes, int32_t* threads);

/*
 * md_pack_weights: fp32 weights -> WPK tiles (device side, run once per load).
 *   w      : float32, element (row i, k, tap) at  w[i*s_row + k*s_k + tap*s_tap]
 *            (Conv3d weight [Co][Ci][27]: s_row=Ci*27, s_k=27, s_tap=1;
 *             NIN W [Ci][Co] (layers.py:576): s_row=1, s_k=Co, s_tap=0)
 *   rows,kdim : logical sizes; tiles are zero-padded to NT / KC multiples.
 */
int md_pack_weights(const float* w, void* wpk, int32_t rows, int32_t kdim, int32_t taps,
                    int64_t s_row, int64_t s_k, int64_t s_tap, int32_t nt, int32_t kc,
                    int32_t prec, void* stream);
int64_t md_packed_weight_bytes(int32_t rows, int32_t kdim, int32_t taps, int32_t nt, int32_t kc);

/*
 * GroupNorm statistics + apply (+SiLU) + bf16 split.
 * Replaces nn.GroupNorm(32, C, eps=1e-6) + nn.SiLU (layers.py:652,660,676,681,
 * :589; ddpm_res64.py:120,186) and torch.cat([h, skip], 1) (ddpm_res64.py:174-176):
 * a concatenated input is expressed as two calls with channel offsets.
 *
 * md_gn_stats : accumulates per-(b, channel) sum / sum-of-squares (double) of an
 *               F32B tensor x[B][C/8][P][8] into sums[B][c_total][2] at channel
 *               offset c_off.  `sums` must be zeroed by the caller (md_zero).
 * md_gn_finalize: per-(b,c) params float4 = (mean(group), rstd(group)*gamma[c], beta[c], rstd(group))
 *               (biased variance, as torch's GroupNorm); optionally also the folded affine `ac`.
 * md_gn_apply : y = (x-mean)*rstd*gamma + beta (norm=1) ; y = silu(y) (silu=1); writes the
 *               split-bf16 S16B tensor out[B][c_total/8][2][P][8] at c_off.
 *               norm=0 copies/splits raw x (used for the NIN shortcut input).
 *               out_raw (may be NULL): second S16B tensor of the same shape receiving the bf16 split of
 *               the raw x from the same read (ResnetBlock: GN path and NIN shortcut share one pass).
 *               `silu` is a bit field: 1 = apply SiLU, 4 = MD_PREC_FP16X2 output (plane 0 = fp16(y),
 *               plane 1 not written), 2 = debug: round y to fp16 before the bf16 split.
 *               drop_p > 0 (training, nn.Dropout of ResnetBlockDDPM, layers.py:682): after SiLU, y is zeroed with
 *               probability drop_p and scaled by 1/(1-drop_p) otherwise.  The mask is a counter-based hash of
 *               (drop_seed, b, channel quad, position) -- md_gn_bwd_stats/apply regenerate it from the same
 *               (drop_p, drop_seed); md_dropout_scale writes it out as an F32B tensor of {0, 1/(1-p)} (tests).
 */
int md_gn_stats(const float* x, double* sums, int32_t batch, int32_t C, int64_t P,
                int32_t c_total, int32_t c_off, void* stream);
int md_gn_finalize(const double* sums, const float* gamma, const float* beta,
                   float* params, int32_t batch, int32_t c_total, int32_t groups,
                   int64_t P, float eps, float* ac, void* stream);
                   /* ac (may be NULL): float [B][c_total][2] = (rstd*gamma, beta - mean*rstd*gamma), the affine
                    * form consumed by md_gemm_conv's MD_B_F32B_GN operand mode */
int md_gn_apply(const float* x, const float* params, void* out, void* out_raw, int32_t batch,
                int32_t C, int64_t P, int32_t c_total, int32_t c_off, int32_t norm,
                int32_t silu, float drop_p, uint64_t drop_seed, void* stream);
int md_dropout_scale(float* out, int32_t batch, int32_t C, int64_t P, int32_t c_total, int32_t c_off,
                     float drop_p, uint64_t drop_seed, void* stream);
int md_zero(void* p, int64_t bytes, void* stream);

/*
 * Timestep embedding + dense layers (layers.py:542-556, ddpm_res64.py:132-136,
 * layers.py:679-680).
 * md_timestep_embedding: emb[b] = [sin(t_b*f_j) | cos(t_b*f_j)], f_j = exp(-ln(1e4)*j/(dim/2-1)).
 * md_linear: y[b][o] = sum_i act(x[b][i]) * w[o*in + i] + bias[o]  (act = SiLU if silu_in).
 */
int md_timestep_embedding(const float* t, float* emb, int32_t batch, int32_t dim, void* stream);
int md_linear(const float* x, const float* w, const float* bias, float* y, int32_t batch,
              int32_t in_dim, int32_t out_dim, int32_t silu_in, void* stream);

/*
 * md_fold_dx: second half of the dx-folded head convolution (ddpm_res64.py:121,189 / ddpm_res128.py:132,208).  The conv
 * runs with rows = (co, dx) and k x k x 1 taps (MD_CFG_C3X_32 / MD_CFG_C5X_32_K16: the 4 output channels would fill 4
 * of the 32 rows of an MFMA tile, the kx (co, dx) pairs fill 12 / 20), y: F32B [B][rows_alloc/8][P][8]; this kernel adds
 * the kx shifted columns: out[b][co][z][y][x] = bias[co] + sum_dx y[b][co*kx + dx][z][y][x + dx - kx/2] (x in range),
 * NCDHW fp32 [B][co][D*H*W].
 */
int md_fold_dx(const float* y, const float* bias, float* out, int32_t batch, int32_t co, int32_t kx, int32_t rows_alloc,
               int32_t D, int32_t H, int32_t W, void* stream);
/* NCDHW fp32 -> S16B with kx x-shifted copies per channel (channel ci*kx + dx = x[ci] shifted by dx - kx/2 along x, zero
 * outside): operand of the dx-folded stem conv (ddpm_res64.py:87,140 / ddpm_res128.py:90,150), MD_CFG_C3X_128_K16 / C5X_128. */
int md_ncdhw_to_s16b_xfold(const float* x, void* out, int32_t batch, int32_t C, int32_t kx, int32_t c_pad, int32_t D, int32_t H,
                           int32_t W, void* stream);
/* NCDHW fp32 [B][C][P] -> S16B [B][c_pad/8][2][P][8] (channels >= C zero filled). */
int md_ncdhw_to_s16b(const float* x, void* out, int32_t batch, int32_t C, int32_t c_pad,
                     int64_t P, void* stream);
/* F32B [B][C/8][P][8] <-> NCDHW [B][C][P] (tests / debugging / parity probes). */
int md_f32b_to_ncdhw(const float* x, float* out, int32_t batch, int32_t C, int64_t P, void* stream);
int md_ncdhw_to_f32b(const float* x, float* out, int32_t batch, int32_t C, int64_t P, void* stream);
int md_s16b_to_ncdhw(const void* x, float* out, int32_t batch, int32_t C, int64_t P, void* stream);

/*
 * Attention softmax over keys (layers.py:603-605).  s: F32B-like [B][N/8][N][8]
 * holding S^T (keys blocked by 8, queries as positions); p: S16B [B][N/8][2][N][8].
 */
int md_softmax_keys(const float* s, void* p, int32_t batch, int32_t n_keys, int32_t n_q,
                    void* stream);

/*
 * md_nin_f32: the ResnetBlock shortcut NIN_0 (1x1x1 channel mixing, layers.py:573-582 used at :667,:688) for the HBM-bound
 * shapes, as a persistent, barrier-free streaming kernel: the whole WPK weight matrix (MD_CFG_G1_128 tiles, K <= 256) is
 * resident in LDS, the fp32 input goes HBM -> registers -> bf16 split -> MFMA.
 *   x1, x2 : F32B [B][c1/8][P][8], [B][c2/8][P][8]: the channel parts of torch.cat([h, skip], 1) (x2 NULL when c2 = 0)
 *   wpk    : md_pack_weights(W, rows = cout, kdim = c1 + c2, taps = 1, nt = 128, kc = 32)
 *   out    : F32B [B][cout/8][P][8] = W^T x + bias
 * Supported: cout == 128, c1 + c2 in {128, 256}, c1 % 16 == c2 % 16 == 0, P % 256 == 0; otherwise MD_ERR_UNSUPPORTED
 * (md_gemm_conv handles every shape).  n_cu: workgroups to launch (0 = 256, one per CU).
 */
int md_nin_f32(const float* x1, const float* x2, int32_t c1, int32_t c2, const void* wpk, const float* bias,
               float* out, int32_t batch, int32_t cout, int64_t P, int32_t n_cu, void* stream);

/*
 * Winograd F(2,3)-along-w path of the 3x3x3 stride-1 convolution (inference): nn.GroupNorm + nn.SiLU + nn.Conv3d
 * (layers.py:676-681, :118-124), on torch.cat([h, skip], 1) (ddpm_res64.py:174-176) or on the nearest-x2 upsampled input
 * (layers.py:618-623).  2/3 of the matrix-core work of the direct form; bf16x3 products, fp32 accumulation.
 *
 * md_wino_prep: fp32 parts -> T[B][C/8][4][2][D][H][W/2][8 bf16]  (C = c1 + c2; 4 = transformed inputs d0-d2, d1+d2, d2-d1,
 *   d1-d3 of the output pair (2i, 2i+1), d_k = activated input at x = 2i-1+k, zero outside the grid; 2 = bf16 hi / lo plane)
 *   x1, x2 : F32B [B][c1/8][Pin][8], [B][c2/8][Pin][8] (x2 NULL when c2 = 0); Pin = D*H*W, or D*H*W/8 with ups = 1
 *   ac     : [B][C][2] folded GroupNorm affine of md_gn_finalize (y = x*ac[c][0] + ac[c][1]), or NULL: no affine, no SiLU
 *   D, H, W: OUTPUT grid of the convolution (W even); md_wino_operand_bytes gives the size of T
 *   drop_p > 0 (training, one part, no upsampling): nn.Dropout after SiLU with the mask of md_gn_apply for the same
 *            (drop_p, drop_seed) -- the forward conv and the taped activation see the same zeros
 * md_wino_pack_weights: Conv3d weight fp32 -> transformed, split tiles in fragment order; element (row, k, tap t27 = (kd*3+kh)*3+kw)
 *   is read at w[row*s_row + k*s_k + t27] ([Cout][Cin][3][3][3]: s_row = Cin*27, s_k = 27, flip = 0) or, with flip = 1, at
 *   w[row*s_row + k*s_k + 26 - t27] (the data-gradient conv of W[Co][Ci][27]: cout = Ci, cin = Co, s_row = 27, s_k = Ci*27)
 *   [Cout/128][Cin/16][kd*3+kh][4][row tile 4][plane 2][k-group 2][row 32][8 bf16]  (md_wino_weight_bytes)
 * md_conv3_wino: out F32B [B][cout/8][P][8] = conv(T, wpk) + bias[b*bias_bstride + co] + residual; stats as in MdGemmConvArgs
 *   Supported: cout % 128 == 0, cin % 32 == 0, D % 4 == 0, H % 8 == 0, W % 8 == 0, D*H*W < 2^28; else MD_ERR_UNSUPPORTED.
 *   variant: 0 (others are A/B schedules and timing-only ablations of tools/bench_wino.py).
 */
int64_t md_wino_operand_bytes(int32_t batch, int32_t cin, int32_t D, int32_t H, int32_t W);
int md_wino_prep(const float* x1, const float* x2, int32_t c1, int32_t c2, const float* ac, int32_t silu, int32_t ups,
                 void* t_out, int32_t batch, int32_t D, int32_t H, int32_t W, float drop_p, uint64_t drop_seed, void* stream);
int64_t md_wino_weight_bytes(int32_t cout, int32_t cin);
int md_wino_pack_weights(const float* w, void* wpk, int32_t cout, int32_t cin, int64_t s_row, int64_t s_k, int32_t flip,
                         void* stream);
int md_conv3_wino(const void* t_in, const void* wpk, float* out, const float* bias, int64_t bias_bstride,
                  const float* residual, int64_t res_bstride, double* stats, int32_t batch, int32_t cin, int32_t cout,
                  int32_t D, int32_t H, int32_t W, int32_t variant, void* stream);

/*
 * The same convolution (same reference lines, inference only) in the "f16f8" arithmetic: every product a*b of the Winograd-domain
 * contraction is  fp16(a) fp16(b)  +  [e4m3(a) e4m3(b_lo 2^11) + e4m3(a_lo 2^11) e4m3(b)] 2^-11,  a_lo = a - fp16(a): one
 * v_mfma_f32_32x32x16_f16 plus half a K-concatenated v_mfma_scale_f32_32x32x64_f8f6f4 (OCP e4m3, E8M0 scale) = two 32-cycle
 * matrix-core units per product where bf16x3 issues three; fp32 accumulation.  Error of one conv vs fp64 1.3e-5 (bf16x3 5.5e-6).
 * md_wino_prep_f8: as md_wino_prep (no dropout; W must divide 256), T of the same geometry with plane 0 = 8 fp16 and plane 1 =
 *   [e4m3(t) x 8 | e4m3((t - fp16(t)) 2^11) x 8]; size: md_wino_operand_bytes.
 * md_wino_pack_weights_f8 (two kernels: max |w| -> power-of-two pre-scale 2^sw kept on the device, then the fragments):
 *   [Cout/128][Cin*9/32 step pairs][4][row tile 4][piece 4][lane 64][16 B] + a 256-byte header {max |w|, sw, 2^-sw};
 *   size: md_wino_weight_bytes_f8.  Forward orientation only (s_row = Cin*27, s_k = 27 for [Cout][Cin][3][3][3]).
 * md_conv3_wino_f8: arguments, supported shapes and outputs as md_conv3_wino with T / wpk from the two calls above.
 * eq (md_wino_prep_f8 / _f6 and md_wino_pack_weights_f8 / _f6; may be NULL = none): float [Cin], the static per-input-channel
 *   power-of-two equaliser written by md_wino_equaliser.  The operand pass multiplies the activated value of channel c by eq[c], the
 *   weight fragments hold w[:, c] / eq[c] (both exact): the convolution is unchanged, but the channels of a K block reach the
 *   4-bit-significand cross-term images at comparable magnitudes whatever the GroupNorm affine in front of the conv is.  The SAME
 *   vector must be given to the operand pass and to the weight packing of a layer.
 * md_wino_equaliser (csrc/wino_eq.hip): eq[c] = 2^round(log2(g_c / a_c) / 2), clamped to 2^+-14, with a_c = rms of
 *   silu(gamma_c z + beta_c) over z ~ N(0, 1) (64-point midpoint rule on [-6, 6]) -- what nn.GroupNorm + nn.SiLU
 *   (layers.py:652,660,676-681) hand the conv, per channel -- and g_c = rms of w[:, c, :, :, :]; w element (co, ci, t27) at
 *   w[co * s_row + ci * s_k + t27].  gamma / beta: the GroupNorm affine over the (concatenated) Cin input channels.  The exponent is
 *   additionally bounded by the fp16 headroom of the operand: eq[c] (8 |gamma_c| + |beta_c|) 2 < 2^15.
 * md_wino_equaliser_measured (ABI 15): the same with a_c^2 = a2m[c], the MEASURED per-channel mean square of the operand (float [Cin]
 *   from md_wino_operand_ms over a calibration evaluation); gamma / beta may both be NULL (a conv on a tensor no GroupNorm precedes:
 *   Upsample, layers.py:611-623).  GroupNorm normalises groups of channels, so a channel's own scale inside its group -- which the
 *   static estimate cannot see -- is part of the measurement.  All exponents are then shifted by one common u so that
 *   max_c eq[c] sqrt(a2m[c]) = 2^0 (+-1/2): the operand's fp16 plane sits mid-range whatever the tensor's absolute magnitude.
 * md_wino_operand_ms (ABI 15): ms[c] += mean over samples and positions of act(x[c])^2 for up to two concatenated F32B parts, act =
 *   the conv's operand transform (x * ac[..][0] + ac[..][1], SiLU) or the identity (ac NULL); ms zeroed by the caller.
 */
int md_wino_equaliser(const float* gamma, const float* beta, const float* w, int32_t cout, int32_t cin, int64_t s_row, int64_t s_k,
                      float* eq, void* stream);
int md_wino_equaliser_measured(const float* gamma, const float* beta, const float* w, const float* a2m, int32_t cout, int32_t cin,
                               int64_t s_row, int64_t s_k, float* eq, void* stream);
int md_wino_operand_ms(const float* x1, const float* x2, int32_t c1, int32_t c2, const float* ac, int32_t silu, int32_t batch, int64_t P,
                       float* ms, void* stream);
int md_wino_prep_f8(const float* x1, const float* x2, int32_t c1, int32_t c2, const float* ac, int32_t silu, int32_t ups,
                    const float* eq, void* t_out, int32_t batch, int32_t D, int32_t H, int32_t W, void* stream);
int64_t md_wino_weight_bytes_f8(int32_t cout, int32_t cin);
int md_wino_pack_weights_f8(const float* w, const float* eq, void* wpk, int32_t cout, int32_t cin, int64_t s_row, int64_t s_k, void* stream);
int md_conv3_wino_f8(const void* t_in, const void* wpk, float* out, const float* bias, int64_t bias_bstride,
                     const float* residual, int64_t res_bstride, double* stats, int32_t batch, int32_t cin, int32_t cout,
                     int32_t D, int32_t H, int32_t W, void* stream);
/*
 * Training (ABI 15): the data-gradient convs of a step in the f16f6 arithmetic.
 * md_absmax: amax_bits[0] = max(amax_bits[0], max |x|) as the bit pattern of the float (the caller zeroes the word; x 16-byte aligned).
 * md_wino_prep_dual_f6: as md_wino_prep_dual for ONE raw fp32 part (an output gradient; c % 16 == 0, W | 256), except that T is the
 *   f16f6 operand of md_conv3_wino_f6 of 2^k x the tensor (gradient magnitudes lifted into the fp16 plane's normal range, saturated
 *   at its ends): amax_bits != NULL: k from the word md_absmax filled for this tensor (max |x| 2^k in [16, 32)); NULL: 2^k = tscale.
 *   U (md_wgrad_wino's operand) and sums are those of the unscaled tensor, bit-identical to md_wino_prep_dual's.
 * md_conv3_wino_f6_scaled: md_conv3_wino_f6 whose result is multiplied by out_scale (exact: a power of two) and, with amax_bits, by
 *   the inverse of the lift the operand pass derived from the same word.
 * The weights: an MD_PACK_WINO_F6 job of md_pack_batch (fixed pre-scale, `flip` for the data-gradient orientation).
 */
int md_absmax(const float* x, int64_t n, uint32_t* amax_bits, void* stream);
int md_wino_prep_dual_f6(const float* x, int32_t c, void* t_out, void* u_out, float* sums, float tscale, const uint32_t* amax_bits,
                         int32_t batch, int32_t D, int32_t H, int32_t W, void* stream);
int md_conv3_wino_f6_scaled(const void* t_in, const void* wpk, float* out, const float* bias, int64_t bias_bstride,
                            const float* residual, int64_t res_bstride, double* stats, int32_t batch, int32_t cin, int32_t cout,
                            int32_t D, int32_t H, int32_t W, float out_scale, const uint32_t* amax_bits, void* stream);

/*
 * The same convolution in the "f16f6" arithmetic: as f16f8, the two cross terms from MX block-scaled OCP e2m3 images (a K block = 16
 * channels x (value, (value - fp16(value)) 2^11) with one power-of-two scale), which the same K = 64 MFMA executes at twice its e4m3
 * rate (format code 2; measured 0.85 of the f16f8 pair-step in the fp16 mix, tools/probes/f6_probe.hip).  Error of one conv vs fp64
 * 1.7e-5 (tools/f16f8_numerics.py).  Buffers, sizes and arguments as the _f8 calls; plane 1 of T / pieces 2-3 of the weight
 * fragments hold the 32-byte record [6 dwords of codes | E8M0 byte | 0] of a block, split over its two 8-channel items.
 * md_wino_prep_f6 needs c1 and c2 to be multiples of 16 (whole K blocks per part).
 */
int md_wino_prep_f6(const float* x1, const float* x2, int32_t c1, int32_t c2, const float* ac, int32_t silu, int32_t ups,
                    const float* eq, void* t_out, int32_t batch, int32_t D, int32_t H, int32_t W, void* stream);
int md_wino_pack_weights_f6(const float* w, const float* eq, void* wpk, int32_t cout, int32_t cin, int64_t s_row, int64_t s_k, void* stream);
int md_conv3_wino_f6(const void* t_in, const void* wpk, float* out, const float* bias, int64_t bias_bstride,
                     const float* residual, int64_t res_bstride, double* stats, int32_t batch, int32_t cin, int32_t cout,
                     int32_t D, int32_t H, int32_t W, void* stream);

/*
 * md_conv3_stem: the dx-folded 3x3x3 input convolution from 4 channels (csrc/conv3_stem.hip; ddpm_res64.py:87-92 applied
 * :138-146: conv3x3(channels, nf)(x) + pos_layer(coords) + mask_layer(mask)) with the GroupNorm sums of its output: replaces
 * md_gemm_conv(MD_CFG_C3X_128_K16) + md_gn_stats.
 *   x16      : S16B [B][2][2][P][8]: the operand of md_ncdhw_to_s16b_xfold(kx = 3, c_pad = 16)
 *   wpk      : md_pack_weights(W2, rows = cout, kdim = 12 (padded to 16), taps = 9, nt = 128, kc = 16), W2[co][(ci, kx)][kd][kh]
 *   out      : F32B [B][cout/8][P][8] = conv + bias[co] + residual[co][p]
 *   residual : F32B [cout/8][P][8] shared by the batch (the input-independent terms), or NULL;  bias: [cout] or NULL
 *   stats    : optional zeroed double [B][cout][2] += per-(sample, channel) (sum, sum of squares) of out
 * Supported: cout % 8 == 0, D % 4 == 0, H % 8 == 0, W % 8 == 0; else MD_ERR_UNSUPPORTED.
 */
int md_conv3_stem(const void* x16, const void* wpk, float* out, const float* bias, const float* residual, double* stats,
                  int32_t batch, int32_t cout, int32_t D, int32_t H, int32_t W, void* stream);

/*
 * md_conv3_head: GroupNorm + SiLU + the dx-folded 3x3x3 output head in one kernel for inference (csrc/conv3_head.hip;
 * ddpm_res64.py:120-121 applied :186-189): replaces md_gn_apply + md_gemm_conv(MD_CFG_C3X_32); md_fold_dx finishes the conv.
 *   x   : F32B [B][cin/8][P][8] (the un-normalised tensor);  ac: [B][cin][2] folded GroupNorm affine of md_gn_finalize
 *   wpk : md_pack_weights(W2, rows = co * 3, kdim = cin, taps = 9, nt = 32, kc = 16), W2[(co, kw)][ci][kd][kh] (see md_fold_dx)
 *   y   : F32B [B][rows_alloc/8][P][8], rows (co, kw); no bias
 * Supported: cin % 32 == 0, rows_alloc in {8, 16, 24, 32}, D % 4 == 0, H % 8 == 0, W % 8 == 0, D*H*W < 2^30.
 */
int md_conv3_head(const float* x, const float* ac, const void* wpk, float* y, int32_t batch, int32_t cin, int32_t rows_alloc,
                  int32_t D, int32_t H, int32_t W, void* stream);

/*
 * md_pack_batch: several md_pack_weights / md_wino_pack_weights jobs in one launch (csrc/pack_batch.hip; a training step
 * re-packs every weight once: ~260 launches otherwise).  Bit-identical to the single-weight entry points.
 *   jobs_dev     : DEVICE array of n_jobs MdPackJob, sorted by block0; job j owns the blocks from block0 on
 *   total_blocks : sum over jobs of their block counts
 *   tiled = 0    : one 16-byte item per thread; a job has ceil(n_items / 256) blocks
 *     kind MD_PACK_WPK : fields as the arguments of md_pack_weights (n_items = md_packed_weight_bytes / 16)
 *     kind MD_PACK_WINO: rows = cout, kdim = cin, s_row, s_k, flip as md_wino_pack_weights (n_items = md_wino_weight_bytes / 16)
 *     kind MD_PACK_WINO_F6 (ABI 15): the f16f6 fragments of md_conv3_wino_f6 with a FIXED pre-scale 2^prec in place of the max |w|
 *                        pass of md_wino_pack_weights_f6 (no equaliser), `flip` as above; n_items = md_wino_weight_bytes_f8 / 16
 *                        (fragments + the 256-byte header, which the job writes)
 *   tiled = 1    : MD_PACK_WPK jobs only, taps > 1 with s_tap = +-1, 16 * kc * taps <= 13824 and nt % 16 == 0: a block stages 16
 *                  rows x one K chunk x all taps through LDS (contiguous reads, 256-byte write runs); a job has
 *                  ceil(rows / nt) * (nt / 16) * ceil(kdim / kc) blocks.  The form for the 3x3x3 weights, which hold most of the
 *                  parameters.  tiled = taps * 100 + kc (2732, 2716, 932, 916): the same with that geometry compiled in
 *                  (every job of the table must have it; float4 source reads).
 *   tiled = 3    : MD_PACK_WINO jobs, block-cooperative (a block = 32 rows x 16 channels x 27 taps); a job has
 *                  (cout / 32) * (cin / 16) blocks.
 */
enum { MD_PACK_WPK = 0, MD_PACK_WINO = 1, MD_PACK_WINO_F6 = 2 };
typedef struct MdPackJob {
  const float* w;
  void* out;
  int64_t s_row, s_k, s_tap, n_items, block0;
  int32_t rows, kdim, taps, nt, kc, prec, flip, kind;
} MdPackJob;
int md_pack_batch(const MdPackJob* jobs_dev, int32_t n_jobs, int64_t total_blocks, int32_t tiled, void* stream);

/*
 * md_conv3_s2: the stride-2 3x3x3 convolution of Downsample (layers.py:626-643: F.pad(x, (0, 1, 0, 1, 0, 1)) +
 * nn.Conv3d(C, C, 3, stride=2, padding=0)) for inference, reading the raw fp32 tensor (csrc/conv3_s2.hip): replaces the
 * split pass md_gn_apply(norm = 0) + md_gemm_conv(MD_CFG_C3_S2).  bf16x3 products, fp32 accumulation.
 *   x      : F32B [B][cin/8][(2D)(2H)(2W)][8]
 *   wpk    : md_pack_weights(W, rows = cout, kdim = cin, taps = 27, nt = 128, kc = 16)
 *   out    : F32B [B][rows_alloc/8][D*H*W][8] = conv + bias[b*bias_bstride + co]   (D, H, W: OUTPUT grid)
 *   stats  : optional zeroed double [B][rows_alloc][2] += per-(sample, channel) (sum, sum of squares) of out
 * Supported: cin % 32 == 0, D % 4 == 0, H % 8 == 0, W % 8 == 0, D*H*W < 2^27; else MD_ERR_UNSUPPORTED (md_gemm_conv
 * with MD_CFG_C3_S2 handles every shape).
 */
int md_conv3_s2(const float* x, const void* wpk, float* out, const float* bias, int64_t bias_bstride, double* stats,
                int32_t batch, int32_t cin, int32_t rows, int32_t rows_alloc, int32_t D, int32_t H, int32_t W, void* stream);

/* md_wino_prep in two phases through LDS (csrc/wino_prep2.hip), bit-identical output, 13 % faster (the host package's default);
 * additionally needs W | 256 and D*H*W % 256 == 0 (whole rows per workgroup), else MD_ERR_UNSUPPORTED. */
int md_wino_prep_v2(const float* x1, const float* x2, int32_t c1, int32_t c2, const float* ac, int32_t silu, int32_t ups,
                    void* t_out, int32_t batch, int32_t D, int32_t H, int32_t W, float drop_p, uint64_t drop_seed, void* stream);

/* md_wino_prep_v2 with a second output u_out in the layout of T: per output pair (x = 2i, 2i+1) of the (activated) tensor
 * v the four values (v[2i], v[2i] + v[2i+1], v[2i] - v[2i+1], v[2i+1]), hi / lo bf16 planes -- the dY operand of
 * md_wgrad_wino when the tensor is an output gradient (training backward: one pass over dY feeds both the Winograd data
 * gradient conv, T, and the Winograd weight gradient, U).  sums (optional, float [B][c1 + c2], accumulated with atomics, not
 * with ups): += per-(sample, channel) sums of the activated tensor over the grid -- the bias gradient of the same pass.
 * Same shape restrictions as md_wino_prep_v2. */
int md_wino_prep_dual(const float* x1, const float* x2, int32_t c1, int32_t c2, const float* ac, int32_t silu, int32_t ups,
                      void* t_out, void* u_out, float* sums, int32_t batch, int32_t D, int32_t H, int32_t W, float drop_p,
                      uint64_t drop_seed, void* stream);

/*
 * md_attn_fwd: fused single-head self-attention (AttnBlock.forward, layers.py:595-608: the two einsums :602,:606 and the
 * softmax :604) -- QK^T, online softmax over the keys and PV in one kernel, bf16x3 MFMA for both contractions; the
 * [B][N][N] score matrix is never materialised.
 *   qk  : S16B [B][2C/8][2][N][8]  q = channels 0..C-1, k = channels C..2C-1 (NIN_0 | NIN_1 outputs, one GEMM)
 *   vT  : S16B over tokens [B][N/8][2][C][8]  (NIN_2 output WITHOUT its bias, token-major)
 *   out : S16B [B][C/8][2][N][8]   o[c][q] = sum_key v[c][key] softmax_key(scale * k[:,key].q[:,q]) + bias_v[c]
 * Supported: C == 256, N % 128 == 0 (the 16^3 attention levels of ddpm_res64); other shapes return MD_ERR_UNSUPPORTED
 * and take the GEMM + md_softmax_keys path.
 */
int md_attn_fwd(const void* qk, const void* vT, void* out, const float* bias_v, int32_t batch, int32_t C,
                int32_t N, float scale, void* stream);

/*
 * One DDPM ancestral-sampling update (models/utils.py:191-198 score scaling,
 * sampling.py:222-230 predictor, sampling.py:476-478 mask), all NCDHW fp32:
 *   x_mean = (x - beta/sigma * eps) / sqrt(1-beta);  x = x_mean + sqrt(beta)*z
 *   x, x_mean *= mask[P]   (mask may be NULL)
 * beta/sigma/… are passed per batch element (coef[b][4] = beta, sigma, sqrt(1-beta), sqrt(beta)).
 */
int md_ancestral_step(const float* x, const float* eps, const float* z, const float* mask,
                      const float* coef, float* x_out, float* x_mean_out, int32_t batch,
                      int32_t C, int64_t P, void* stream);
/*
 * One deterministic DDIM update (sde_lib.py:113-140 `discretize_ddim` + the mask / inpainting lines of
 * `ddim_sampler`, sampling.py:556-565).  State in float64 like the reference:
 *   x0s = x - a2*eps ; x0_pred = x0s/a1 ; x_new = r1*x + (r2-r1)*(x - x0s) ; both *= mask[P] (mask may be NULL);
 *   channel `ch`: v = v*(1-pmask) + partial*pmask when partial/pmask ([P] each) are given.
 * coef[b][4] = {a1 = sqrt(abar_t), a2 = sqrt(1-abar_t), r1 = a1_prev/a1, r2 = a2_prev/a2} as doubles;
 * x_f32 receives float(x_new): the next U-Net input.
 */
int md_ddim_step(const double* x, const float* eps, const float* mask, const double* coef, const float* partial,
                 const float* pmask, int32_t ch, double* x_out, double* x0_out, float* x_f32, int32_t batch,
                 int32_t C, int64_t P, void* stream);
/*
 * Inpainting blend of one channel (sampling.py:443-467):
 *   v = (x*(1-m) + src*m) * gm     applied in place to channel `ch` of x [B][C][P];
 *   src has batch stride src_bstride (0 = shared partial grid).
 */
int md_inpaint_blend(float* x, const float* src, const float* pmask, const float* gmask,
                     int32_t batch, int32_t C, int32_t ch, int64_t P, int64_t src_bstride,
                     void* stream);

/*
 * Re-noise the conditioned channel to the current level (sampling.py:460-466), in place:
 *   upd = coef[b][0]*x + coef[b][1]*z[b] ; x[ch] = (x[ch]*(1-m) + upd*m)*gm ; x_mean[ch] = x[ch]
 * coef[b] = (exp(log_mean_coeff(t)), std(t)) of VPSDE.marginal_prob (sde_lib.py:210-214);
 * z is [B][P]; x_mean may be NULL.
 */
int md_inpaint_renoise(float* x, float* x_mean, const float* z, const float* pmask,
                       const float* gmask, const float* coef, int32_t batch, int32_t C,
                       int32_t ch, int64_t P, void* stream);

/*
 * Training-step pieces that do not need the U-Net backward (lib/diffusion/losses.py:38-78, models/ema.py:32-51).
 *   md_ddpm_perturb : out = (coef[b][0]*x0 + coef[b][1]*noise) * mask            (losses.py:63-65)
 *   md_masked_sq_err: sums[b] += sum_i (eps_hat-noise)^2 * mask (fp64; zero `sums` first);
 *                     grad (may be NULL) = gscale * 2 (eps_hat-noise) * mask = dLoss/d eps_hat   (losses.py:68-78)
 *   md_grad_sqnorm  : *out += sum g^2 (fp64; zero first)          (clip_grad_norm_, losses.py:48)
 *   md_adam_ema_step: clip by min(1, max_norm/(sqrt(*grad_sqnorm)+1e-6)) (skipped when grad_sqnorm NULL or
 *                     max_norm < 0), torch.optim.Adam update with bias correction for `step` (1-based),
 *                     then ema -= (1-ema_decay)*(ema - p) (ema may be NULL).  One pass over 5 flat arrays.
 */
int md_ddpm_perturb(const float* x0, const float* noise, const float* mask, const float* coef, float* out,
                    int32_t batch, int32_t C, int64_t P, void* stream);
int md_masked_sq_err(const float* eps_hat, const float* noise, const float* mask, double* sums, float* grad,
                     float gscale, int32_t batch, int32_t C, int64_t P, void* stream);
int md_grad_sqnorm(const float* g, int64_t n, double* out, void* stream);
int md_adam_ema_step(float* p, const float* g, float* m, float* v, float* ema, int64_t n, float lr, float beta1,
                     float beta2, float eps, float weight_decay, int32_t step, float ema_decay,
                     const double* grad_sqnorm, float max_norm, void* stream);

/*
 * Backward-pass support (csrc/backward.hip).  Contractions of the backward re-use md_gemm_conv:
 *   dgrad  = the forward conv kernels on WPK tiles packed from the flipped/transposed weight
 *   wgrad  = split-K GEMM over (position, sample) on PB16 operands, one B-pointer offset per tap
 * PB16: bf16 [guard + (D+2)(H+2)(W+2) + guard][B/8][2][C][8 samples], zero padded.
 *   md_to_pb16      : F32B (mode 0) / S16B (mode 1) -> PB16; up=1 nearest-upsamples a (D/2)^3 source, stuff=1
 *                     places it at odd fine positions (stride-2 conv backward).  Zero-fills `out` first.
 *   md_wgrad_finish : GEMM result [ntap][rows/8][cols_alloc][8] -> dw[row*s_row + col*s_k + (tap0+t)*s_tap] += .
 *   md_gn_bwd_*     : GroupNorm(+SiLU)(+dropout) backward in three streaming passes (see backward.hip); `sums`
 *                     zeroed by the caller; dgamma/dbeta accumulate; (drop_p, drop_seed) as given to md_gn_apply.
 *   md_channel_sums : out[b][c] += sum_p x (bias / FiLM gradients); md_grad_resample: Upsample backward
 *                     (mode 0: sum of the 8 children) and zero-stuffing (mode 1).
 */
int64_t md_pb16_bytes(int32_t batch, int32_t C, int32_t D, int32_t H, int32_t W, int32_t guard, int32_t pad);
int md_to_pb16(const void* src, void* out, int32_t batch, int32_t C, int32_t c_src, int32_t D, int32_t H, int32_t W,
               int32_t guard, int32_t pad, int32_t mode, int32_t up, int32_t stuff, int32_t zsplit, int32_t zhalo,
               void* stream);
               /* channels >= c_src are zero; pad = halo of the grid: 1 (3x3x3, 1x1x1 consumers) or 2 (5x5x5).
                * zsplit in {1,2,4,8}: each sample is cut into zsplit z-slabs of D planes (whole depth D*zsplit) and the
                * slabs become the "samples" of the sample blocks (batch*zsplit virtual samples), so that batches below 8
                * still fill the 8-sample k-groups of md_wgrad; zhalo = 1 (activation operand): z-halo planes hold the
                * neighbouring slab's data, zhalo = 0 (dY operand): they are zero.  md_pb16_bytes / md_wgrad then take
                * batch*zsplit and the slab dims (D, H, W). */
int md_wgrad_finish(const float* g, float* dw, int32_t rows, int32_t cols, int32_t cols_alloc, int32_t ntap,
                    int32_t tap0, int64_t s_row, int64_t s_k, int64_t s_tap, void* stream);
/*
 * md_wgrad: dw[row*s_row + col*s_k + tap*s_tap] += sum_{position, sample} dY[row][pos] * A[col][pos + off(tap)]
 * for the taps of a k^3 convolution, taps = k^3 in {27, 125} (tap = (dz*k+dy)*k+dx, positions on the grid padded by
 * p = k/2), or the single tap of a 1x1x1 layer (taps = 1, p = 1) -- autograd of nn.Conv3d (layers.py:118-124,
 * ddpm_res128.py:90-92,132) / NIN (layers.py:573-582).
 * dy_pb / act_pb: PB16 tensors from md_to_pb16 (pad = p) with a_ch / b_ch channels (multiples of 8) on the same cubic
 * grid (H = W; D may be a z-slab depth, see md_to_pb16) and `guard` >= p((H+2p)^2 + (H+2p) + 1) + 10; rows <= a_ch,
 * cols <= b_ch are the valid co / ci.  bf16x3 MFMA, fp32
 * accumulate; the contraction is split into `ksplit` position ranges whose partial sums live in `workspace`
 * (md_wgrad_workspace_bytes) and are reduced in a fixed order, so results are run-to-run identical.
 */
int64_t md_wgrad_workspace_bytes(int32_t rows, int32_t cols, int32_t taps, int32_t ksplit);
int md_wgrad(const void* dy_pb, const void* act_pb, float* dw, void* workspace, int64_t workspace_bytes, int32_t batch,
             int32_t a_ch, int32_t b_ch, int32_t rows, int32_t cols, int32_t D, int32_t H, int32_t W, int32_t guard,
             int32_t taps, int32_t ksplit, int64_t s_row, int64_t s_k, int64_t s_tap, void* stream);

/*
 * md_wgrad_wino: weight gradient of a 3x3x3 stride-1 convolution in the Winograd F(2,3)-along-w domain (csrc/wgrad_wino.hip;
 * autograd of nn.Conv3d, layers.py:118-124, for the layers whose forward runs through md_conv3_wino):
 *   dw[co*s_row + ci*s_k + ((kd*3+kh)*3+kw)*s_tap] += sum_{sample, position} dY[co][pos] * A[ci][pos + (kd-1, kh-1, kw-1)]
 * from  u_dy  = md_wino_prep_dual's second output for dY            [B][co/8][4][2][D][H][W/2][8 bf16]
 *       t_act = the operand T of the forward convolution (md_wino_prep / _v2 of the activated input, kept from the forward)
 * 4 products per output pair and (kd, kh) instead of 6 (2/3 of md_wgrad's matrix-core work), no PB16 re-layout; bf16x3 MFMA,
 * fp32 accumulate.  co, ci multiples of 128; W in {32, 64} (rows of 16 / 32 pairs), D, H >= 2.  The contraction is split
 * into `ksplit` ranges of (sample, z) planes (1 <= ksplit <= batch * (D - 1)); partial sums live in `workspace`
 * (md_wgrad_wino_workspace_bytes) and are reduced in a fixed order: results are run-to-run identical.
 */
int64_t md_wgrad_wino_workspace_bytes(int32_t co, int32_t ci, int32_t ksplit);
int md_wgrad_wino(const void* u_dy, const void* t_act, float* dw, void* workspace, int64_t workspace_bytes, int32_t batch,
                  int32_t co, int32_t ci, int32_t D, int32_t H, int32_t W, int32_t ksplit, int64_t s_row, int64_t s_k,
                  int64_t s_tap, void* stream);

/*
 * md_wgrad_nin: weight gradient of a 1x1x1 NIN layer (autograd of layers.py:573-582) straight from S16B tensors:
 *   dw[co*s_row + ci*s_k] += sum_{sample, position} dY[co][pos] * x[ci][pos]
 * dy_s16 / x_s16: S16B [B][C/8][2][P][8] (md_gn_apply with norm = 0 of dY / of the layer's input) -- no PB16 re-layout
 * (md_to_pb16 + md_wgrad with taps = 1 remain for the other shapes).  co, ci multiples of 128, P a multiple of 16; the
 * contraction is split into `ksplit` ranges of 16-position elements (1 <= ksplit <= batch * P / 16), reduced in a fixed order.
 */
int64_t md_wgrad_nin_workspace_bytes(int32_t co, int32_t ci, int32_t ksplit);
int md_wgrad_nin(const void* dy_s16, const void* x_s16, float* dw, void* workspace, int64_t workspace_bytes, int32_t batch,
                 int32_t co, int32_t ci, int64_t P, int32_t ksplit, int64_t s_row, int64_t s_k, void* stream);
int md_gn_bwd_stats(const float* x, const float* dy, const float* params, double* sums, int32_t batch, int32_t C,
                    int64_t P, int32_t c_total, int32_t c_off, int32_t dy_ctotal, int32_t silu, float drop_p,
                    uint64_t drop_seed, void* stream);
int md_gn_bwd_finalize(const double* sums, const float* params, const float* gamma, float* coef, float* dgamma,
                       float* dbeta, int32_t batch, int32_t c_total, int32_t groups, int64_t P, void* stream);
int md_gn_bwd_apply(const float* x, const float* dy, const float* params, const float* coef, float* dx, int32_t batch,
                    int32_t C, int64_t P, int32_t c_total, int32_t c_off, int32_t dy_ctotal, int32_t silu,
                    int32_t accumulate, float drop_p, uint64_t drop_seed, const float* residual, float* ch_sums,
                    uint32_t* amax_bits, void* stream);
                    /* amax_bits (ABI 15, may be NULL): zeroed word receiving max |dx| of this call as md_absmax would (the lift of the
                     * f16f6 data-gradient conv that consumes dx: saves that conv's separate md_absmax pass) */
                    /* residual (may be NULL, ignored when accumulate): F32B like dx, dx = residual + gradient (the identity
                     * shortcut of a ResnetBlock without a separate copy); ch_sums (may be NULL): float [B][c_total],
                     * += per-(sample, channel) sum of the gradient written for this part (bias / FiLM gradients). */
int md_channel_sums(const float* x, float* out, int32_t batch, int32_t C, int64_t P, void* stream);
/* S16B blocked transpose: in [B][R/8][2][Cn][8] -> out [B][Cn/8][2][R][8] (attention backward operands). */
int md_s16b_transpose(const void* in, void* out, int32_t batch, int32_t R, int32_t Cn, void* stream);
/* softmax-over-keys backward: ds = split(alpha * P * (dP - sum_keys P*dP)); layouts as md_softmax_keys. */
int md_softmax_keys_bwd(const void* p, const float* dp, void* ds, int32_t batch, int32_t n_keys, int32_t n_q,
                        float alpha, void* stream);
int md_grad_resample(const float* in, float* out, int32_t batch, int32_t C, int32_t Dc, int32_t Hc, int32_t Wc,
                     int32_t mode, int32_t accumulate, void* stream);

/*
 * Marching tetrahedra on a STATIC tet grid (nvdiffrec/lib/geometry/dmtet.py:105-163), n_meshes meshes per call:
 * chunks of 1024 edges / tets per workgroup, four launches on `stream` (chunk totals, per-mesh scan, vertices, faces).
 * Static tables (built once on the host from the tet file, see meshdiffusion_amd/dmtet.py):
 *   tets       int32 [T][4]
 *   edges      int32 [E][2]   lexicographically sorted unique (min,max) vertex pairs of all tets
 *   tet_edges  int32 [T][6]   edge id of each tet's 6 edges in base_tet_edges order (dmtet.py:54)
 * Per mesh:   pos float32 [M][N][3], sdf float32 [M][N]
 * Outputs:    verts float32 [M][E][3] (first counts[m][0] rows valid),
 *             faces int64 [M][2T][3]  (first counts[m][1] rows valid),
 *             face_tet int64 [M][2T]  (tet id of each face = face_to_valid_tet; may be NULL),
 *             counts int32 [M][4] = {n_verts, n_faces, n_tets_1tri, n_tets_2tri}
 * Face order == reference: all 1-triangle tets (tet order), then 2-triangle tets.
 * Workspace:  md_marching_tets_workspace_bytes(M, E, T) bytes (edge -> vertex id table + per-chunk counters).
 * tets must be 16-byte and edges 8-byte aligned (rows are read whole).  n_meshes <= 65535.
 */
int64_t md_marching_tets_workspace_bytes(int32_t n_meshes, int32_t n_edges, int32_t n_tets);
int md_marching_tets(const float* pos, const float* sdf, const int32_t* tets,
                     const int32_t* edges, const int32_t* tet_edges, int32_t n_meshes,
                     int32_t n_verts, int32_t n_edges, int32_t n_tets, float* verts,
                     int64_t* faces, int64_t* face_tet, int32_t* counts, void* workspace,
                     int64_t workspace_bytes, void* stream);

/*
 * Smooth vertex normals of an extracted mesh (nvdiffrec/lib/render/mesh.py:200-229 `auto_normals`, called from
 * nvdiffrec/eval.py:422 on the marching-tets output): face normals cross(v1-v0, v2-v0) accumulated on their three
 * vertices, degenerate sums replaced by (0,0,1), normalised.  verts float32 [V][3], faces int64 [F][3] (as returned by
 * md_marching_tets); v_nrm float32 [V][3] (zeroed by the call), f_nrm float32 [F][3] or NULL (unnormalised face normals).
 */
int md_vertex_normals(const float* verts, const int64_t* faces, int64_t n_verts, int64_t n_faces, float* v_nrm,
                      float* f_nrm, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MESHDIFFUSION_HIP_H */
