/*
 * libspeecht5_hip.so -- C ABI of the MI355X (gfx950) SpeechT5 forward/backward hot path.
 *
 * The reference (microsoft/SpeechT5) has no FFI: its hot path is PyTorch op call sites inside
 * SpeechT5/speecht5/models/modules/[module].py.  Each entry point below replaces
 * the torch op(s) at the cited reference call site; INTEGRATION.md shows the ctypes stub that a
 * reference maintainer would add.  Conventions:
 *   - plain pointers + sizes, caller-owned device buffers, no torch types;
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work on it;
 *   - `dtype` is ST5_F32 (0) or ST5_BF16 (1): the element type of activations/weights;
 *     statistics, softmax, accumulators, biases, LayerNorm affine params and weight gradients are fp32;
 *   - return value 0 = ok, non-zero = ST5_ERR_* (argument / alignment / launch error);
 *   - threading / devices: the library is built for ONE process per GPU driven by ONE host thread (the reference's
 *     execution model).  It keeps process-global mutable state -- the stream-fork event ring, the
 *     A/B switches (st5_gemm_set_*, st5_layernorm_set_max_blocks) and tables of PER-STREAM state: the deferred split-K
 *     queue with its slab arena (st5_gemm_defer_splitk / _flush_splitk; an arena is hipFree'd and re-allocated when it
 *     must grow, which synchronises the device), the split-K slab workspaces, the deferred LayerNorm reductions
 *     (st5_layernorm_defer / _flush) and the workspace of the ordered row scatter.  One host thread may therefore keep
 *     several streams busy at once (the two micro-batches of an update side by side).  Table sizes: 32 streams with
 *     deferred LayerNorm reductions and 32 with deferred split-K reductions (a 33rd, e.g. in a test session that keeps
 *     creating streams, restarts the table after a device synchronisation, provided nothing is queued on any state -- a
 *     state is never handed from one live stream to another); 8 with split-K slab workspaces (taken over after a device
 *     synchronisation).  A flush only folds what was
 *     queued on the stream it is given.  Calling the library from two host threads, or for
 *     two devices from one process, is NOT supported.
 */
#ifndef SPEECHT5_HIP_H
#define SPEECHT5_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ST5_F32 0
#define ST5_BF16 1

#define ST5_ACT_NONE 0
#define ST5_ACT_GELU 1
#define ST5_ACT_RELU 2
#define ST5_ACT_TANH 3
#define ST5_ACT_LRELU_01 4  /* LeakyReLU(0.1), HiFi-GAN */
#define ST5_ACT_LRELU_001 5 /* LeakyReLU(0.01) */

/* gemm flags */
#define ST5_GEMM_A_KSTRIDED 1  /* A(i,k) stored k-outer (i contiguous): "transposed" operand      */
#define ST5_GEMM_B_KSTRIDED 2  /* B(j,k) stored k-outer (j contiguous)                            */
#define ST5_GEMM_OUT_F32 4     /* C / R are fp32 regardless of dtype (weight gradients)           */
#define ST5_GEMM_DACT 8        /* multiply result by act'(P) (P = saved pre-activation, C layout) */
#define ST5_GEMM_DEFERRABLE 16 /* C is not read before st5_gemm_flush_splitk: a split-K reduction may be queued (st5_gemm_defer_splitk) */

/* Generalised operand addressing (elements):
 *   off(outer, inner) = (rpb ? (outer / rpb) * bstride + (outer % rpb) * ld : outer * ld)
 *                     + (seg ? (inner / seg) * seg_stride + inner % seg : inner)
 *                     + (z / zdiv) * zs0 + (z % zdiv) * zs1            (z = batch index)
 * K-major operand: outer = row (i or j), inner = k.  K-strided operand: outer = k, inner = row.
 * This one descriptor covers nn.Linear fwd/dgrad/wgrad, strided / grouped conv1d as implicit GEMM
 * on channels-last activations (overlapping rows: ld = stride*C, inner spans k*C), and the
 * per-(batch,head) attention matmuls. */
typedef struct st5_operand {
  const void* ptr;
  int64_t ld;
  int32_t rpb;
  int32_t seg;
  int64_t bstride;
  int64_t seg_stride;
  int64_t zs0, zs1;
} st5_operand;

typedef struct st5_gemm_params {
  st5_operand A;    /* [M x K] */
  st5_operand B;    /* [N x K] (nn.Linear weight layout when K-major) */
  st5_operand C;    /* [M x N] output; outer = row, inner = col */
  st5_operand R;    /* optional residual added to the output (ptr may be NULL), C layout */
  st5_operand P;    /* optional pre-activation for ST5_GEMM_DACT, C layout */
  st5_operand Cpre; /* optional second output receiving the pre-activation value, C layout */
  const float* bias; /* optional fp32 [N] (+ z * bias_zs) */
  int64_t bias_zs;
  int32_t M, N, K;
  int32_t batch, zdiv;
  int32_t act;      /* ST5_ACT_* applied after bias */
  int32_t flags;
  float alpha;      /* scales A*B before bias */
  float beta;       /* C = result + beta * C_old (0 => C_old not read) */
  float dropout_p;  /* dropout applied last (after act, before residual); 0 => off */
  uint64_t seed;    /* dropout RNG seed; element counter = z*M*N + row*N + col */
  float* asum;      /* optional fp32 [M] (batch == 1): asum[m] += sum_k A(m, k).  With A = dY^T this is the bias gradient,
                     * computed by the weight-gradient GEMM itself (one extra MFMA column in the first column of tiles)
                     * instead of a separate reduction over dY (autograd of F.linear: grad_bias = dY.sum(0)). */
} st5_gemm_params;

/* Generic MFMA GEMM: C = drop(act(alpha * A.B^T + bias)) [* act'(P)] + R + beta*C.
 * Replaces F.linear at multihead_attention.py:213-231,397; transformer_layer.py:127-131,385-389;
 * speech_encoder_prenet.py:177; F.conv1d at speech_encoder_prenet.py:300 (layers 1..6) and :107
 * (pos_conv); torch.bmm at multihead_attention.py:340,389; and their autograd backward passes. */
int st5_gemm(const st5_gemm_params* p, int dtype, void* stream);
/* 1 (default): plain NT GEMMs use the LDS-DMA pipelined kernel; 0: always the register-staged kernel (A/B testing). */
/* Stream ordering for the host side (no torch types): everything enqueued on `to` after this call runs after everything
 * enqueued on `from` before it.  Used to run the weight-gradient GEMMs of a backward pass (off the critical path: only the
 * optimizer / gradient all-reduce needs them) on a second stream beside the data-gradient chain. */
int st5_stream_fork(void* from_stream, void* to_stream);
int st5_gemm_set_glds(int enabled);
/* NT block tile: 0 = chosen per problem (default), 1 = 128x128 always, 2 = first 256x256 kernel always, 3 / 4 = phased 256x256 kernel always
 * with / without the half-phase stagger, 5 = 64x128 kernel always (A/B measurements only). */
int st5_gemm_set_nt_tile(int mode);
/* (mode 5 = the 64x128 kernel always.)  bf16 NT GEMMs of at most `tiles` tiles of 128x128 (x batch) run on 64x128 tiles, three blocks per CU
 * (round 6: the transformer's Linear GEMMs at 8 utterances per GPU -- modules/transformer_layer.py:127-131, multihead_attention.py:213-231 --
 * are 0.25-3 rounds of 128x128 tiles); 0 = never.  Results are bit-identical for every choice. */
int st5_gemm_set_m64_max_tiles(int tiles);
/* (A/B) bf16 NT GEMMs of at least `tiles` tiles of 256x256 with at least `nk` k-tiles of 64 also run on the phased 256x256 kernel (the
 * N = 768 long-reduction shapes: one block on 96 CUs instead of 384 tiles of 128x128 on all of them); tiles = 0: off (default). */
int st5_gemm_set_nt_longk(int tiles, int nk);
/* Block count the split-K choice of the fp32-output (weight-gradient) GEMMs aims for; default 384 (1.5 per CU). */
int st5_gemm_set_splitk_target(int blocks);
/* Weight-gradient (TN form, no row split / segments) GEMMs: 0 (default) = always the 128x128 LDS-DMA kernel; 1 = phased 256x256 kernel with
 * its own split-K count (M, N multiples of 256, K >= 512), 2 = the same without the half-phase stagger (A/B measurements only). */
int st5_gemm_set_tn_phased(int mode);
/* 128x128 NT kernel: grids of at most max_blocks blocks (one per CU) use an nbuf-stage operand ring (2 = never; default 256, 4). */
int st5_gemm_set_deep_ring(int max_blocks, int nbuf);
/* 128x128 NT kernel on grids of more than one block per CU: 5 (default) = operand tiles through a ring of five 16 KB LDS slots (A and B
 * of a k-step are separate ring entries: 2.5 k-steps of loads in flight, 80 KB, two blocks per CU), 4 = two whole stages (64 KB).
 * Bit-identical results; A/B switch. */
int st5_gemm_set_nt_slots(int slots);
/* Batch the slab reductions of split-K GEMMs (weight gradients): while enabled, a split-K st5_gemm only queues its reduction;
 * st5_gemm_flush_splitk launches ONE kernel that folds every queued reduction into its output (the outputs are complete
 * only after the flush; same stream as the GEMMs).  Used by the data-parallel wrapper, which flushes before it reduces a
 * gradient bucket across ranks and at the end of backward.  Disabling flushes. */
/* MX-fp8 NT GEMM for the d = 1024 / FFN 4096 Linears of t5_transformer_large (models/speecht5.py:1402-1425; BASELINE.json
 * configs[4]): C = epilogue(A . B^T) with the epilogue features of st5_gemm, A [M x K] and B [N x K] as OCP fp8 e4m3 bytes
 * (p->A / p->B: ld in bytes, K-major, no row split / segments / batch, K % 128 == 0) and one e8m0 scale byte per 32 consecutive
 * k-elements of a row (a_scale [M x K/32], b_scale [N x K/32], row pitches % 4 == 0): element = fp8 * 2^(scale - 127), the OCP
 * microscaling (MX) format consumed directly by v_mfma_scale_f32_32x32x64_f8f6f4.  C-class operands (C, R, P, Cpre) are bf16.
 * Replaces the same F.linear call sites as st5_gemm when the fp8 compute mode is on (speecht5_amd.functional.set_fp8). */
int st5_gemm_mxfp8(const st5_gemm_params* p, const uint8_t* a_scale, int64_t a_scale_ld, const uint8_t* b_scale, int64_t b_scale_ld,
                   void* stream);
/* st5_gemm_mxfp8 whose bf16 output C [M x N] (N % 32 == 0, plain layout) is ALSO written as its MX-fp8 image -- out_q [M x N] e4m3 bytes,
 * out_s [M x N/32] e8m0 scale bytes: the bytes st5_quant_mxfp8 would produce from C -- by the epilogue, for the two producers whose output is
 * the next fp8 GEMM's A operand: bias + GELU with the pre-activation copy (fc1 forward, transformer_layer.py:127-131) and x act'(P) (the data
 * gradient of fc2).  Any other epilogue combination: ST5_ERR_ARG. */
int st5_gemm_mxfp8_q(const st5_gemm_params* p, const uint8_t* a_scale, int64_t a_scale_ld, const uint8_t* b_scale, int64_t b_scale_ld,
                     void* out_q, int64_t out_q_ld, uint8_t* out_s, int64_t out_s_ld, void* stream);
/* Block tile of st5_gemm_mxfp8: 0 (default) = per problem -- the phased 256x256 kernel (gemm_nt8p_mx8_kernel: st5_gemm's phased bf16
 * schedule on fp8 bytes, twice the reduction depth per LDS byte and matrix-pipe cycle) for problems of several rounds of 256x256 tiles
 * or one nearly full round, else 128x128; 1 = 128x128 always, 2 = phased 256x256 always.  Results are bit-identical for every choice. */
int st5_gemm_set_mx8_tile(int mode);
/* The per-problem choice keeps launches with a heavy epilogue (GELU / its derivative, pre-activation copy, dropout, fp8 image) on 128x128
 * tiles, two blocks per CU, unless the reduction has at least `nk` k-tiles of 128 (default 16; 0 = no such rule).  A/B knob. */
int st5_gemm_set_mx8_heavy_nk(int nk);
/* MX quantisation along rows of a bf16 matrix x [rows x cols] (ld elements, cols % 32 == 0): q = e4m3(x * 2^(127 - s)) bytes
 * (pitch q_ld), s[r][c / 32] = floor(log2(max|finite block elements|)) - 8 + 127 as e8m0 (pitch s_ld); round to nearest even, finite
 * values saturating at +-448.  NaN / Inf propagate: the element becomes the e4m3 NaN code 0x7f and its block's scale the e8m0 NaN 0xff,
 * so a diverged tensor still turns the GEMM output (and the loss / gradient norm behind it) non-finite. */
int st5_quant_mxfp8(const void* x, int64_t ld, void* q, int64_t q_ld, uint8_t* s, int64_t s_ld, int64_t rows, int32_t cols, void* stream);
/* The same for njobs CONTIGUOUS bf16 matrices in one launch -- the fp8 images of every eligible Linear weight and of its transposed copy
 * (forward / data-gradient operand of F.linear, transformer_layer.py:127-131,385-389), refreshed once per optimizer step.  jobs: DEVICE array
 * of njobs records { const void* x; void* q; uint8_t* s; int64_t elems (rows * cols, cols % 32 == 0); int32_t cols; int32_t blk0 }, blk0 = the
 * job's first block at 2048 elements per block (ascending, job 0 at 0); nblocks = the total. */
int st5_multi_quant_mxfp8(const void* jobs, int32_t njobs, int32_t nblocks, void* stream);
/* n <= 8 weight-gradient GEMMs (each as st5_gemm would take it: A, B k-strided bf16, ST5_GEMM_OUT_F32, no epilogue but beta and asum; the
 * four / six weight gradients of a transformer layer: autograd of F.linear at transformer_layer.py:127-131,385-389, multihead_attention.py:
 * 213-231,397) as ONE launch in which every block runs the whole token reduction of its tile -- no split-K slabs, no reduction kernel.  The
 * operands must stay valid until the launch has run (the caller queues the problems and keeps the tensors).  A problem's result is
 * independent of what it is grouped with (bit for bit).  A group with a problem of another form, or with two problems writing the same
 * output, runs through st5_gemm one by one. */
int st5_gemm_tn_group(const st5_gemm_params* list, int32_t n, int dtype, void* stream);
/* Block tile of st5_gemm_tn_group: 0 (default) = per problem -- the phased 256x256 grouped kernel (gemm_tn8p_group_kernel: whole token
 * reductions on st5_gemm's phased schedule; weight gradients bit-identical to the 128x128 group, bias-gradient column summed on the VALU in
 * its own fixed order) when M and N are multiples of 256 and K >= 512, i.e. every Linear of the transformer; 1 = 128x128 always. */
int st5_gemm_set_tn_group_tile(int mode);
/* 1 when st5_gemm_tn_group would run a weight gradient of this shape on the phased kernel (host-side accounting of 256x256 tiles). */
int st5_gemm_tn_group_is_phased(int32_t M, int32_t N, int32_t K);
int st5_gemm_defer_splitk(int enabled, void* stream);
int st5_gemm_flush_splitk(void* stream);

/* ---- row-wise normalisation (encoder.py:226, transformer_layer.py:124,132, speech_encoder_prenet.py:174) */
/* y = LN(x) * gamma + beta over the last dim (cols); saves mean/rstd (fp32 [rows]). */
int st5_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                      float* rstd, int64_t rows, int32_t cols, float eps, int dtype, void* stream);
/* dx; dgamma/dbeta are ACCUMULATED (+=) into fp32 [cols]; ws >= st5_layernorm_bwd_ws_bytes(). */
int st5_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx,
                      float* dgamma, float* dbeta, void* ws, int64_t rows, int32_t cols,
                      void* dx_dropped /* optional 2nd output dx * dropout_mask(seed, row*cols + c): the gradient of the
                                          dropout-epilogue Linear in front of this LayerNorm (cols % 4 == 0) */,
                      float drop_p, uint64_t drop_seed, int dtype, void* stream);
/* Batched dgamma / dbeta reductions: while enabled, st5_layernorm_bwd leaves its block partials in an internal arena and
 * st5_layernorm_flush(stream) folds all pending LayerNorms' partials into their dgamma / dbeta with ONE launch (same stream
 * as the st5_layernorm_bwd calls; the owner of the gradient buffers flushes wherever gradients must be complete). */
/* st5_layernorm_fwd for bf16 rows with cols % 32 == 0 (cols <= 2048) that also writes y's MX-fp8 image (q [rows x cols] e4m3 bytes,
 * sc [rows x cols / 32] e8m0 scale bytes: the bytes st5_quant_mxfp8 produces from y): the LayerNorm in front of the QKV projection / fc1 of
 * a pre-LN layer in fp8 compute mode (transformer_layer.py:103-110,124-126 with `encoder_normalize_before`). */
int st5_layernorm_fwd_q8(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, void* q, uint8_t* sc,
                         int64_t rows, int32_t cols, float eps, void* stream);
/* st5_layernorm_bwd whose dx also receives `addend` (dx = LayerNorm backward + addend; same shape / dtype as dx, cols % 4 == 0, cols <= 2048):
 * the gradient of a pre-LN block's residual connection (transformer_layer.py:90-111 with layer_norm_first: y = x + f(LN(x))) folded into
 * the LayerNorm backward at the block's input instead of an autograd accumulation kernel. */
int st5_layernorm_bwd_add(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx, float* dgamma,
                          float* dbeta, void* ws, int64_t rows, int32_t cols, const void* addend, int dtype, void* stream);
/* y = GELU(LN(x) * gamma + beta) in ONE pass and its backward (dx, dgamma += , dbeta += from the gradient of the activated output; LN(x) is
 * recomputed, nothing but x, mean, rstd is kept): the LayerNorm + GELU behind every convolution of the layer-norm feature extractor
 * (t5_transformer_large: speech_encoder_prenet.py:318-331 with extractor_mode=layer_norm).  cols % 4 == 0, cols <= 512; bf16 uses the GELU
 * polynomials of the GEMM epilogues, fp32 the exact erf form.  ws as st5_layernorm_bwd_ws_bytes(rows, cols). */
int st5_layernorm_gelu_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int64_t rows, int32_t cols,
                           float eps, int dtype, void* stream);
int st5_layernorm_gelu_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean, const float* rstd, void* dx,
                           float* dgamma, float* dbeta, void* ws, int64_t rows, int32_t cols, int dtype, void* stream);
/* The LayerDrop gate folded into a post-LN layer's LAST LayerNorm (modules/encoder.py:251-257, modules/decoder.py:64-67 inside a
 * replayed step, see st5_select): forward  y = *keep ? LN(x) * gamma + beta : skip  (skip = the layer's input, same shape and dtype);
 * backward  as st5_layernorm_bwd with dy counted as zero when *keep == 0 (dx = 0, nothing added to dgamma / dbeta).  The gradient
 * of the layer's OUTPUT then reaches the layer's input through st5_skip_grad.  keep = one float in device memory; cols % 4 == 0. */
int st5_layernorm_gated_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int64_t rows,
                            int32_t cols, float eps, const float* keep, const void* skip, int dtype, void* stream);
int st5_layernorm_gated_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx,
                            float* dgamma, float* dbeta, void* ws, int64_t rows, int32_t cols, void* dx_dropped, float drop_p,
                            uint64_t drop_seed, const float* keep, int dtype, void* stream);
int st5_layernorm_defer(int enabled, void* stream);
int st5_layernorm_flush(void* stream);
int64_t st5_layernorm_bwd_ws_bytes(int64_t rows, int32_t cols);

/* ---- attention probabilities (multihead_attention.py:343-386) ----
 * scores [BH, T, lds] (dtype) hold alpha*q.k^T; qp [BH, T, nb] (dtype) holds alpha*q.pe^T (may be NULL);
 * P[bh,i,j] = softmax_j(scores + qp[bh,i,clip(i-j,-maxrel,maxrel-1)+maxrel] + causal + key-padding(-inf)).
 * Writes probs (dtype, [BH,T,lds], pad cols zeroed) and, when dropout_p > 0, probs_drop (dropout applied).
 * kpm: uint8 [B, S] (1 = padded key) or NULL.  H = heads (bh = b*H + h). */
int st5_softmax_fwd(const void* scores, const void* qp, const uint8_t* kpm, void* probs, void* probs_drop,
                    int32_t BH, int32_t H, int32_t T, int32_t S, int32_t lds, int32_t nb, int32_t maxrel,
                    int32_t causal, float dropout_p, uint64_t seed, int dtype, void* stream);
/* dS = P * (dP' - sum_j dP'_j P_j), dP' = dropout'(dP) (+ dP_extra fp32 on the un-dropped probs);
 * dqp[bh,i,b] (dtype, fully written by the kernel) = sum_{j: bucket(i-j)=b} dS[bh,i,j]. dS overwrites dP. */
int st5_softmax_bwd(void* dP_inout, const void* probs, const float* dP_extra, void* dqp, int32_t BH, int32_t T,
                    int32_t S, int32_t lds, int32_t nb, int32_t maxrel, float dropout_p, uint64_t seed, int dtype,
                    void* stream);

/* ---- fused attention (bf16, head_dim 64): O = dropout(softmax(scale*Q.K^T + relpos-bias + masks)) . V without
 *      materialising scores (multihead_attention.py:340-389).  q/k/v/o are row-major projections with leading dimensions
 *      *_ld: row (b*T + t) (queries) / (b*S + s) (keys), column h*64 + d.  lse fp32 [B*H, T] (saved for backward).
 *      pe: relative key table [nb = 2*maxrel, 64] (dtype) or NULL.  lds: row length used by the dropout counter
 *      (same generator/counters as the unfused st5_softmax path).  Other dtypes / head sizes: use the unfused path. */
int st5_flash_attn_fwd(const void* q, int64_t q_ld, const void* k, int64_t k_ld, const void* v, int64_t v_ld, void* o,
                       int64_t o_ld, float* lse, const void* pe, const uint8_t* kpm, int32_t B, int32_t H, int32_t T,
                       int32_t S, int32_t head_dim, int32_t nb, int32_t maxrel, int32_t causal, int32_t lds, float scale,
                       float dropout_p, uint64_t seed, int dtype, void* stream);

/* Relative-position table of the fused kernels: QP[bh][q][row] (dtype) with row = st5_flash_attn_qp_row(nb) elements:
 * second-generation kernels (default): [8 x QP[0] | scale*log2(e) * q.pe[b]^T, b = 0..nb-1 | 8 x QP[nb-1]] -- the replicated
 * end chunks are what a clipped relative position reads, so clipping is a clamp of a 16-byte chunk index; first generation
 * (st5_flash_attn_set_impl(1)): the plain nb values.  The kernels DMA per-tile windows of it into LDS.
 * st5_flash_attn_qp_table builds it (the forward does this itself when given qp_out); the backward takes it as `qp`. */
int32_t st5_flash_attn_qp_row(int32_t nb);
int st5_flash_attn_qp_table(const void* q, int64_t q_ld, const void* pe, void* qp_out, int32_t B, int32_t H, int32_t T, int32_t nb,
                            float scale, int dtype, void* stream);
/* 2 (default): second-generation kernels (csrc/flash_attn2.hip: LDS-DMA staging, transpose reads, bias windows, two blocks per
 * CU); 1: first generation (csrc/flash_attn.hip).  Same results bit for bit; A/B measurements and tests only. */
int st5_flash_attn_set_impl(int impl);

/* st5_flash_attn_fwd with the relative-position table workspace (qp_out [B*H, T, st5_flash_attn_qp_row(nb)], dtype; required by
 * the second-generation kernels when pe != NULL -- without it the call runs the first-generation kernel, which keeps the table
 * in LDS): built here, returned for the backward's `qp` input. */
int st5_flash_attn_fwd_qp(const void* q, int64_t q_ld, const void* k, int64_t k_ld, const void* v, int64_t v_ld, void* o,
                          int64_t o_ld, float* lse, const void* pe, const uint8_t* kpm, int32_t B, int32_t H, int32_t T,
                          int32_t S, int32_t head_dim, int32_t nb, int32_t maxrel, int32_t causal, int32_t lds, float scale,
                          float dropout_p, uint64_t seed, void* qp_out, int dtype, void* stream);

/* Backward of st5_flash_attn_fwd: recomputes P from (q, k, bias, lse).  Writes dq/dk/dv (dtype, same row layouts as
 * q/k/v with their own leading dimensions); dvec fp32 [B*H*T] scratch (D = rowsum(dO*O)).  With pe: qp = the relative-position table
 * [B*H, T, st5_flash_attn_qp_row(nb)] (from st5_flash_attn_fwd_qp or st5_flash_attn_qp_table) and dqp [B*H, T, nb] (plain) receives the bucket gradients
 * (the caller folds dqp into dq and d(pe) with st5_gemm, exactly as for the unfused path). */
int st5_flash_attn_bwd(const void* q, int64_t q_ld, const void* k, int64_t k_ld, const void* v, int64_t v_ld, const void* o,
                       int64_t o_ld, const void* dout, int64_t do_ld, void* dq, int64_t dq_ld, void* dk, int64_t dk_ld, void* dv,
                       int64_t dv_ld, const float* lse, float* dvec, const void* pe, const void* qp, void* dqp,
                       const uint8_t* kpm, int32_t B, int32_t H, int32_t T, int32_t S, int32_t head_dim, int32_t nb,
                       int32_t maxrel, int32_t causal, int32_t lds, float scale, float dropout_p, uint64_t seed, int dtype,
                       void* stream);
/* The same with the dq and dkv kernels side by side: D on `stream`, then dq on `stream` and dkv on `stream2`
 * (ordered after everything enqueued on `stream`), `stream` joined with `stream2` before returning.  stream2 NULL =
 * st5_flash_attn_bwd.  Results are bit-identical to the single-stream form. */
int st5_flash_attn_bwd_2s(const void* q, int64_t q_ld, const void* k, int64_t k_ld, const void* v, int64_t v_ld, const void* o,
                          int64_t o_ld, const void* dout, int64_t do_ld, void* dq, int64_t dq_ld, void* dk, int64_t dk_ld,
                          void* dv, int64_t dv_ld, const float* lse, float* dvec, const void* pe, const void* qp, void* dqp,
                          const uint8_t* kpm, int32_t B, int32_t H, int32_t T, int32_t S, int32_t head_dim, int32_t nb,
                          int32_t maxrel, int32_t causal, int32_t lds, float scale, float dropout_p, uint64_t seed, int dtype,
                          void* stream, void* stream2);

/* ---- speech pre-net layer 0: Conv1d(1->C,k,stride,no bias) + GroupNorm(C groups) + GELU
 *      (speech_encoder_prenet.py:300,323-324).  wav fp32 [B,S]; out channels-last [B,L,C] (dtype);
 *      stats fp32 [B,C,2] = (mean, rstd) saved for backward.  ws >= st5_conv0_ws_bytes. */
int st5_conv0_gn_gelu_fwd(const float* wav, const float* w, const float* gamma, const float* beta, void* out,
                          float* stats, void* ws, int32_t B, int32_t S, int32_t C, int32_t k, int32_t stride,
                          float eps, int dtype, void* stream);
/* dY channels-last (dtype).  Accumulates (+=) dw [C,k], dgamma [C], dbeta [C] (fp32), scaled by gscale. */
int st5_conv0_gn_gelu_bwd(const float* wav, const float* w, const float* gamma, const float* beta,
                          const float* stats, const void* dY, float* dw, float* dgamma, float* dbeta, void* ws,
                          int32_t B, int32_t S, int32_t C, int32_t k, int32_t stride, float gscale, int dtype,
                          void* stream);
/* The same pair with the waveform moments carried from the forward to the backward: `mom` = B x st5_conv0_mom_count(k) doubles (sum_t
 * x[s t + j] and sum_t x[s t + j] x[s t + j'], j <= j' < k: what the GroupNorm backward needs of the waveform besides dY).  The forward
 * publishes them (NULL: not), the backward given them skips its own pass over the waveform and the fold (two launches of ~17 us for 4 KB
 * of results); NULL in the backward = st5_conv0_gn_gelu_bwd.  Same results bit for bit: the same kernels produce the same doubles. */
int32_t st5_conv0_mom_count(int32_t k);
int st5_conv0_gn_gelu_fwd_m(const float* wav, const float* w, const float* gamma, const float* beta, void* out, float* stats, double* mom,
                            void* ws, int32_t B, int32_t S, int32_t C, int32_t k, int32_t stride, float eps, int dtype, void* stream);
int st5_conv0_gn_gelu_bwd_m(const float* wav, const float* w, const float* gamma, const float* beta, const float* stats, const double* mom,
                            const void* dY, float* dw, float* dgamma, float* dbeta, void* ws, int32_t B, int32_t S, int32_t C, int32_t k,
                            int32_t stride, float gscale, int dtype, void* stream);
/* A/B switch of the bf16 forward apply pass: 1 (default) = convolution on the matrix cores with split-bf16 operands (x = xh + xl,
 * w = wh + wl; wh.xh + wh.xl + wl.xh in two v_mfma_f32_32x32x16_bf16 per 32 channels x 32 steps), 0 = the VALU form. */
int st5_conv0_set_mfma(int on);
/* A/B switch of the matrix-core forward: 1 (default) = the clip's GroupNorm statistics and the weight fragments come from one launch
 * (moments, statistics + fragments, apply: three launches), 0 = separate statistics and fragment launches (four).  `stats` holds the
 * same bits either way. */
int st5_conv0_set_fold(int on);
/* A/B switch of the matrix-core forward's GELU: 1 (default) = a 256-entry chord table of the upper tail Q = 1 - Phi in LDS,
 * gelu(z) = max(z, 0) - |z| Q(|z|) (|error| <= 8.4e-6 |z|, 8.5 VALU issue slots per element), 0 = the transcendental-free
 * polynomial (1.5e-5 |z|, 14.5 slots). */
int st5_conv0_set_gelu_table(int on);
int64_t st5_conv0_ws_bytes(int32_t B, int32_t S, int32_t C, int32_t k, int32_t stride);

/* ---- element-wise / reductions (glue ops fused where the reference has separate torch calls) */
/* dst(dtype) = src(fp32), optionally transposed: src [rows, cols] -> dst [cols, rows] */
int st5_cast_from_f32(const float* src, void* dst, int64_t rows, int64_t cols, int32_t transpose, int dtype,
                      void* stream);
int st5_cast_to_f32(const void* src, float* dst, int64_t n, int dtype, void* stream);
/* out[c] (+)= scale * sum_r x[r, c]   (bias gradients); ws >= st5_colsum_ws_bytes(rows, cols) */
int st5_colsum_ws(const void* x, float* out, void* ws, int64_t rows, int32_t cols, int64_t ld, float scale,
                  int32_t accumulate, int dtype, void* stream);
int64_t st5_colsum_ws_bytes(int64_t rows, int32_t cols);
/* out[0] (+)= scale * sum(x^2)   (features_pen, speech_encoder_prenet.py:172) */
int st5_sumsq(const void* x, float* out, int64_t n, float scale, int32_t accumulate, int dtype, void* stream);
/* y = a*x + b*y */
int st5_axpby(const void* x, void* y, int64_t n, float a, float b, int dtype, void* stream);
/* LayerDrop inside a replayed (HIP-graph) step, where the host's per-layer draw cannot steer control flow
 * (modules/encoder.py:251-257 `if not self.training or (dropout_probability > self.encoder_layerdrop)`, modules/decoder.py:64-67
 * LayerDropModuleList): every layer runs and  y = keep ? b (layer output) : a (layer input),  keep = one float in device memory
 * written from the host draw before the replay; backward (ga, gb) = keep ? (0, g) : (g, 0).  Raw 16-byte chunks, any dtype. */
int st5_select(const float* keep_dev, const void* a, const void* b, void* y, int64_t nbytes, void* stream);
int st5_select_bwd(const float* keep_dev, const void* g, void* ga, void* gb, int64_t nbytes, void* stream);
/* dx = *keep ? dx : g  in place (raw 16-byte chunks): the input gradient of a gated layer (st5_layernorm_gated_fwd) -- a kept layer's
 * own input gradient stands, a dropped layer's (exact zeros) is replaced by the gradient of its output. */
int st5_skip_grad(const float* keep_dev, const void* g, void* dx, int64_t nbytes, void* stream);
/* y = act(x) ; dx = dy * act'(x) */
int st5_act_fwd(const void* x, void* y, int64_t n, int32_t act, int dtype, void* stream);
int st5_act_bwd(const void* dy, const void* x, void* dx, int64_t n, int32_t act, int dtype, void* stream);
/* y[r,c] = act(x[r,c] * a[c] + b[c])   (per-channel affine: spectrogram normalisation, BatchNorm apply) */
int st5_channel_affine(const void* x, const float* a, const float* b, void* y, int64_t rows, int32_t cols, int32_t act,
                       int dtype, void* stream);
/* y = dropout(x) with the library's counter RNG (same generator as the GEMM epilogue) */
int st5_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, int dtype, void* stream);
/* x[r,:] = v (fp32 [cols]) where mask[r] != 0  (apply_hubert_mask, speech_encoder_prenet.py:249) */
int st5_masked_fill_rows(void* x, const uint8_t* mask, const float* v, int64_t rows, int32_t cols, int dtype,
                         void* stream);
/* dv[c] += sum_{r: mask[r]} dx[r,c]; dx[r,:] = 0 where mask[r] */
int st5_masked_fill_rows_bwd(void* dx, const uint8_t* mask, float* dv, int64_t rows, int32_t cols, int dtype,
                             void* stream);
/* y[r,:] = x[r,:] + scale * table[idx[r], :]   (table fp32 [n, cols]); idx int32 [rows] */
int st5_add_table_rows(const void* x, const float* table, const int32_t* idx, void* y, int64_t rows, int32_t cols,
                       float scale, int dtype, void* stream);
/* same with the scale read from device memory (learnable alpha of ScaledPositionalEncoding; no host sync) */
int st5_add_table_rows_dev(const void* x, const float* table, const int32_t* idx, void* y, int64_t rows, int32_t cols,
                           const float* scale_dev, int dtype, void* stream);
/* y[r,:] = emb_scale * table[tok[r],:] + pos_scale * pos[pidx[r],:]   (embedding + positions) */
int st5_embed_rows(const float* table, const int32_t* tok, const float* pos, const int32_t* pidx, void* y,
                   int64_t rows, int32_t cols, float emb_scale, float pos_scale, int dtype, void* stream);
/* dtable[tok[r],:] += scale * dy[r,:]  (fp32 atomics) */
int st5_embed_rows_bwd(const void* dy, const int32_t* tok, float* dtable, int64_t rows, int32_t cols, float scale,
                       int dtype, void* stream);
/* The same, bit-reproducible (no atomics): the rows are rank-sorted by (table row, position) and every table row is summed in
 * position order by one block.  Tokens outside [0, vocab) contribute nothing; cols % 4 == 0.  The sort is quadratic in `rows` (one micro-batch
 * of tokens / frames: ~10 us for 8k rows); rows <= 2^22.  Uses one internal workspace: calls must be stream-ordered. */
int st5_embed_rows_bwd_det(const void* dy, const int32_t* tok, float* dtable, int64_t rows, int32_t cols, int32_t vocab, float scale,
                           int dtype, void* stream);
/* ... with a per-row weight: row r is scaled by row_w[(r / rw_div) % rw_mod] (NULL = 1). */
int st5_embed_rows_bwd_det_w(const void* dy, const int32_t* tok, float* dtable, int64_t rows, int32_t cols, int32_t vocab, float scale,
                             const float* row_w, int32_t rw_div, int32_t rw_mod, int dtype, void* stream);
/* dst[a, b, c] (+)= src[off + a*sa + b*sb + c*sc]  (src fp32, dst `dtype`, element strides, may be negative): one-pass weight
 * re-layout + cast for the implicit-GEMM convolutions, and re-laid-out accumulation of their weight gradients. */
int st5_gather3(const float* src, void* dst, int32_t A, int32_t B, int32_t C, int64_t sa, int64_t sb, int64_t sc, int64_t off,
                int32_t accumulate, int dtype, void* stream);
/* x[b, t, :] = 0 for t < head and for t >= tail_start (x [B, Tp, C]): halo rows + rows no output window covers, one launch. */
int st5_zero_time_edges(void* x, int32_t B, int32_t Tp, int32_t C, int32_t head, int32_t tail_start, int dtype, void* stream);
/* zero-padded copy: dst [B, pad_l + T + pad_r, C] <- src [B, T, C] */
/* out[(b,t), j] = wav[b, t*stride + j] (j < k), 0 (k <= j < kpad); out is [B*L, kpad] (dtype), L = (S-k)/stride + 1.
 * Turns the Cin = 1 first convolution of the extractor_mode=layer_norm feature extractor
 * (speech_encoder_prenet.py:300-318) into rows for st5_gemm. */
int st5_unfold_rows(const float* wav, void* out, int32_t B, int32_t S, int32_t k, int32_t stride, int32_t kpad, int dtype,
                    void* stream);
int st5_pad_time(const void* src, void* dst, int32_t B, int32_t T, int32_t C, int32_t pad_l, int32_t pad_r,
                 int dtype, void* stream);
/* HiFi-GAN generator (HuggingFace SpeechT5HifiGan, modeling_speecht5.py:2887-3066: every convolution is preceded by a LeakyReLU
 * and zero "same" padding): dst [B, pad_l + T + pad_r, C] = zero-padded act(src [B, T, C]) in ONE pass (16-byte vectors when
 * C * sizeof(elem) % 16 == 0), act in {ST5_ACT_NONE, ST5_ACT_LRELU_01, ST5_ACT_LRELU_001, ...} with act(0) == 0. */
int st5_pad_time_act(const void* src, void* dst, int32_t B, int32_t T, int32_t C, int32_t pad_l, int32_t pad_r, int32_t act,
                     int dtype, void* stream);
/* Zero the halo rows [0, pad_l) and [pad_l + T, pad_l + T + pad_r) of every batch element of dst [B, pad_l + T + pad_r, C]: a
 * convolution whose st5_gemm output operand addresses the interior of the padded buffer the NEXT convolution reads. */
int st5_zero_halo(void* dst, int32_t B, int32_t T, int32_t C, int32_t pad_l, int32_t pad_r, int dtype, void* stream);
/* Channels-last Conv1d with 32 or 64 OUTPUT channels, bf16, as an MFMA implicit GEMM whose 32 MFMA rows are the output channels and
 * whose columns are time steps (csrc/conv1d_narrow.hip) -- the 64- and 32-channel stages of the HiFi-GAN generator (HF
 * modeling_speecht5.py:2920-2960 residual blocks, :3010-3030 upsampler), where the 128-wide tiles of st5_gemm waste 1/2 .. 3/4 of
 * their MFMAs:
 *   y[b, t, co] = act(alpha * sum_{tau < taps, c < Cin} w[co, tau * Cin + c] * x[b, t, tau, c] + bias[co]) + residual[b, t, co] + beta * y[b, t, co]
 * with x[b, t, tau, c] = x[b * x_bs + t * x_ts + tau * tap_stride + c] (all strides in ELEMENTS; the caller's buffer holds the halo rows:
 * a dilated convolution is tap_stride = dilation * Cin over the zero-padded input, one phase of a stride-s transposed convolution is
 * taps = 2, tap_stride = Cin), y / residual addressed as base + b * bs + t * ts + co (padded or phase-interleaved layouts without
 * copies).  Cin in {32, 64, 128}, Cout in {32, 64}; x, w 16-byte aligned, strides multiples of 8 (x) / 4 (y, residual) elements,
 * else ST5_ERR_ARG / ST5_ERR_ALIGN.  act in {ST5_ACT_NONE, ST5_ACT_LRELU_01, ST5_ACT_LRELU_001}.  bias fp32 or NULL; residual NULL for none. */
int st5_conv1d_narrow(const void* x, int64_t x_bs, int32_t x_ts, const void* w, const float* bias, void* y, int64_t y_bs, int32_t y_ts,
                      const void* residual, int64_t r_bs, int32_t r_ts, int32_t B, int32_t L, int32_t Cin, int32_t Cout, int32_t taps,
                      int32_t tap_stride, float alpha, float beta, int32_t act, void* stream);
/* The generator's last convolution (HF modeling_speecht5.py:3058-3062: 32 -> 1 channels, k = 7, tanh), bf16:
 *   y[b * L + t] = act(alpha * sum_{tau, c} w[tau * Cin + c] * x[b * x_bs + t * x_ts + tau * tap_stride + c] + bias[0]). */
int st5_conv1d_cout1(const void* x, int64_t x_bs, int32_t x_ts, const void* w, const float* bias, void* y, int32_t B, int32_t L, int32_t Cin,
                     int32_t taps, int32_t tap_stride, float alpha, int32_t act, void* stream);

/* ---- log-mel front end of the input pipeline (data/speech_dataset.py:142-181 `logmelfilterbank`: librosa.stft
 *      n_fft 1024 / hop 256 / periodic Hann / centred with reflect padding, |.|, 80 Slaney mel filters, log10 floor 1e-10).
 *      The two matrix products (windowed DFT basis, mel filterbank) are st5_gemm calls in fp32; these are the pieces around them.
 *      st5_stft_frames: out fp32 [B * L, n_fft], L = 1 + S / hop, out[(b,l), j] = wav[b, reflect(l*hop + j - n_fft/2)].
 *      st5_stft_magnitude: reim fp32 [rows, 2*ldh] (real parts in columns [0,nbins), imaginary in [ldh, ldh+nbins)) ->
 *      mag fp32 [rows, ldh] = sqrt(re^2 + im^2), columns >= nbins zero.  st5_log10_floor: y = log10(max(x, floor)). */
int st5_stft_frames(const float* wav, float* out, int32_t B, int32_t S, int32_t n_fft, int32_t hop, void* stream);
int st5_stft_magnitude(const float* reim, float* mag, int64_t rows, int32_t nbins, int32_t ldh, void* stream);
int st5_log10_floor(const float* x, float* y, int64_t n, float floor_value, void* stream);

/* ---- collation of the speech-pretraining batch on the device (data/speech_dataset.py:302-446 SpeechPretrainDataset.collater,
 *      collater_audio, crop_to_max_size, collater_frm_label; fairseq collate_tokens): every tensor of the batch is a RAGGED GATHER of
 *      the items' rows -- crop at a per-item start, thin out by a stride (the reduction factor), shift by one frame, pad to the
 *      longest.  st5_ragged_rows: out[b, t, 0:w] = src[b][(off[b] + t * step) * w + (0:w)] when t >= tmin and
 *      0 <= off[b] + t * step < hi[b], else the pad pattern; src = device array of B item base pointers, off / hi int32 [B], elements
 *      of es = 1, 4 or 8 bytes (copied as bits: fp32 wave / mel rows, int64 labels), pad_bits = the pad element's bit pattern.
 *      st5_tail_mask: out[b, t] = (t >= n[b]) ? 1 : 0 as uint8 (dtype_f32 = 0: padding_mask, :392-402) or fp32 (dtype_f32 = 1: the
 *      stop-token labels, :347-349).  The crop starts are drawn on the host from numpy's stream exactly as the reference draws them
 *      (one np.random.randint per item longer than the batch's audio size); speecht5_amd/collate.py drives both. */
int st5_ragged_rows(const void* const* src, const int32_t* off, const int32_t* hi, void* out, int32_t B, int32_t T, int32_t w, int32_t step,
                    int32_t tmin, int32_t es, uint64_t pad_bits, void* stream);
int st5_tail_mask(const int32_t* n, void* out, int32_t B, int32_t T, int32_t dtype_f32, void* stream);

/* ---- BatchNorm1d + tanh + dropout (+ residual) on channels-last rows: the non-GEMM part of the mel post-net
 *      (speech_decoder_postnet.py:39-51,65-70 = espnet Tacotron Postnet: 5 x [Conv1d k5 -> BatchNorm1d -> tanh -> dropout], the
 *      last block without tanh; after = before + postnet(before)).  x fp32 [rows, C] = the convolution GEMM's fp32 output
 *      (ST5_GEMM_OUT_F32), C % 4 == 0.  Training: batch statistics (fp64 partial sums), running_mean / running_var /
 *      num_batches_tracked updated like torch.nn.BatchNorm1d (momentum blend, unbiased variance); eval: running statistics.
 *      y = dropout(act((x - mean) * rstd * gamma + beta)) [+ residual], act in {ST5_ACT_NONE, ST5_ACT_TANH}; dropout element
 *      counters = row * C + c.  y: `dtype` elements, or fp32 when y_f32 (the block's fp32 `after` output); residual: `dtype`
 *      [rows, C].  Output row r goes to element offset (r / out_L) * out_bstride + (r % out_L) * C + out_off (out_L = 0: r * C)
 *      and `halo` rows in front of / behind every out_L-row batch block are zeroed: the next convolution's time-padded
 *      operand is written directly.  stats fp32 [2*C] (mean, rstd) is saved for the backward.  ws >= st5_batchnorm_ws_bytes(C). */
int64_t st5_batchnorm_ws_bytes(int32_t C);
int st5_batchnorm_act_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                          int64_t* num_batches_tracked, float momentum, float eps, int32_t training, int32_t act,
                          float dropout_p, uint64_t seed, const void* residual, void* y, int32_t y_f32, float* stats, void* ws,
                          int64_t rows, int32_t C, int64_t out_L, int64_t out_bstride, int64_t out_off, int32_t halo, int dtype,
                          void* stream);
/* Backward: g = dropout_mask * dy * act'(.) with dy fp32 (the fp32 data gradient of the next convolution);
 * dgamma += sum g*xhat, dbeta += sum g (either may be NULL); dx (dtype, mapped / haloed like the forward's y) =
 * gamma * rstd * (g - mean(g) - xhat * mean(g * xhat)) for batch statistics, gamma * rstd * g for running statistics. */
int st5_batchnorm_act_bwd(const float* x, const float* dy, const float* stats, const float* gamma, const float* beta, float* dgamma,
                          float* dbeta, int32_t training, int32_t act, float dropout_p, uint64_t seed, void* dx, void* ws,
                          int64_t rows, int32_t C, int64_t out_L, int64_t out_bstride, int64_t out_off, int32_t halo, int dtype,
                          void* stream);

/* ---- losses ---- */
/* Row-wise (label-smoothed) cross entropy on logits [rows, ld] (dtype, first V cols valid).
 * loss_sum[0] += sum_r w_r * ((1-eps)*nll_r + eps/V... ) following speech_to_text_loss.py:93-110;
 * target < 0 or == ignore_index rows are skipped.  Also writes dlogits = scale * dloss/dlogits. */
int st5_cross_entropy(const void* logits, const int32_t* target, float* loss_sum, float* nll_sum, void* dlogits,
                      int64_t rows, int32_t V, int64_t ld, float label_smoothing, int32_t ignore_index,
                      float grad_scale, int dtype, void* stream);

/* The same with per-row outputs instead of atomically accumulated sums: row_loss[r] (and row_nll[r] if given) = the row's
 * loss (0 for skipped rows); the caller sums them (deterministic).  This is what the criteria use
 * (speech_pretrain_criterion.py:98-141 NCE cross entropy, text_pretrain_criterion.py:56-60, speech_to_text_loss.py:93-110). */
int st5_cross_entropy_rows(const void* logits, const int32_t* target, float* row_loss, float* row_nll, void* dlogits,
                           int64_t rows, int32_t V, int64_t ld, float label_smoothing, int32_t ignore_index,
                           float grad_scale, int dtype, void* stream);

/* ---- optimizer (fairseq `adam`: decoupled weight decay; README.md:107-115 flags) ----
 * One fused pass over the flat fp32 buffers: g' = g * grad_scale * min(1, max_norm / (sqrt(*gnorm_sq) * grad_scale));
 * m,v update; p = p*(1 - lr*wd) - lr*sqrt(bc2)/bc1 * m / (sqrt(v) + eps) (fairseq/optim/adam.py: eps is not bias-corrected;
 * bc_i = 1 - beta_i^step).  gnorm_sq is a DEVICE scalar (may be NULL). */
int st5_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int32_t step, const float* gnorm_sq, float max_norm, float grad_scale,
                  void* bf16_mirror /* optional bf16 [n]: receives the updated parameters in the compute dtype */, void* stream);
/* The same with the learning rate and the step count read from DEVICE memory (hyper_dev = {lr, step} as two floats; NULL =
 * the by-value arguments): the form a captured HIP graph replays -- its kernel arguments are frozen, the host rewrites the two
 * floats before every replay. */
int st5_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                      float weight_decay, int32_t step, const float* gnorm_sq, float max_norm, float grad_scale, void* bf16_mirror,
                      const float* hyper_dev, void* stream);
/* ---- speech-decoder criterion (Tacotron2Loss with masking, text_to_speech_loss.py:263-345 + the label / length fix-ups of
 * :186-204) as one reduction + one gradient pass.  All tensors fp32 on the device; ys [B, >=L, C] and labels [B, >=L] with
 * their own batch strides (elements); olens int64 [B] (device) = the UNTRIMMED target lengths, r = reduction factor:
 * valid frames are t < olens - olens % r, the stop label of the last valid frame is 1.
 * fwd: out4 = {l1, mse, bce, n_frames}.   bwd: gradients of g_l1 * l1 + g_mse * mse + g_bce * bce (device scalars, NULL = 0)
 * w.r.t. after / before / logits (each may be NULL). */
int64_t st5_tacotron_loss_ws_bytes(void);
int st5_tacotron_loss_fwd(const float* after, const float* before, const float* logits, const float* ys, int64_t ys_bstride,
                          const float* labels, int64_t labels_bstride, const int64_t* olens, int32_t B, int32_t L, int32_t C, int32_t r,
                          float pos_weight, float* out4, void* ws, void* stream);
int st5_tacotron_loss_bwd(const float* after, const float* before, const float* logits, const float* ys, int64_t ys_bstride,
                          const float* labels, int64_t labels_bstride, const int64_t* olens, int32_t B, int32_t L, int32_t C, int32_t r,
                          float pos_weight, const float* out4, const float* g_l1, const float* g_mse, const float* g_bce, float* d_after,
                          float* d_before, float* d_logits, void* stream);
/* ---- CTC loss of the ASR criterion (speech_to_text_loss.py:301-337: F.ctc_loss(lprobs, flat targets, input_lengths,
 * target_lengths, blank, reduction="sum", zero_infinity), torch's own recursion since the reference turns cuDNN off) ----
 * lprobs [T, B, V] fp32 log-probabilities; targets int64, sentence b's labels at targets[target_offsets[b] ...
 * + target_lengths[b]) (lengths clamped to max_target_len); all index arrays on the device.
 * fwd: nll [B], loss[0] = sum_b nll_b (an impossible alignment counts 0 with zero_infinity); ws >= st5_ctc_loss_ws_bytes
 * keeps the alpha rows.  bwd: grad [T, B, V] = grad_out[0] * (exp(lprobs) - posterior) for t < input_lengths[b], else 0 --
 * the form torch's ctc_loss backward returns (the gradient w.r.t. the logits in front of the log-softmax; pushing it through
 * the log-softmax backward leaves it unchanged). */
int64_t st5_ctc_loss_ws_bytes(int32_t T, int32_t B, int32_t max_target_len);
int st5_ctc_loss_fwd(const float* lprobs, const int64_t* targets, const int64_t* target_offsets, const int64_t* input_lengths,
                     const int64_t* target_lengths, int32_t T, int32_t B, int32_t V, int32_t max_target_len, int32_t blank,
                     int32_t zero_infinity, float* nll, float* loss, void* ws, void* stream);
int st5_ctc_loss_bwd(const float* lprobs, const int64_t* targets, const int64_t* target_offsets, const int64_t* input_lengths,
                     const int64_t* target_lengths, int32_t T, int32_t B, int32_t V, int32_t max_target_len, int32_t blank,
                     int32_t zero_infinity, const float* nll, const float* grad_out, const void* ws, float* grad, void* stream);
/* ---- guided attention loss of the TTS criterion (text_to_speech_loss.py:370-427) ----
 * att [B, H, To, Ti] fp32 (the selected heads of the selected layers, concatenated), ilens / olens int64 [B] on the device.
 * fwd: out2 = {alpha * mean over (to < olen_b, ti < ilen_b) of (1 - exp(-(ti/ilen_b - to/olen_b)^2 / (2 sigma^2))) att, alpha / count}.
 * bwd: datt = grad_out[0] * out2[1] * weight inside the selection, 0 outside. */
int64_t st5_guided_attn_ws_bytes(void);
int st5_guided_attn_fwd(const float* att, const int64_t* ilens, const int64_t* olens, int32_t B, int32_t H, int32_t To, int32_t Ti,
                        float sigma, float alpha, float* out2, void* ws, void* stream);
int st5_guided_attn_bwd(const int64_t* ilens, const int64_t* olens, int32_t B, int32_t H, int32_t To, int32_t Ti, float sigma,
                        const float* out2, const float* grad_out, float* datt, void* stream);
/* ---- Gumbel vector quantizer + time-wise code / encoder-state mix (speecht5.py:95-107, 858-882; csrc/vq.hip) ----
 * logits, gumbel [N, G*V] fp32 (V <= 128); vars [G*V, Dg] fp32; enc / out [N, G*Dg] (dtype); mix_w [T] fp32 or NULL (rows are
 * (b, t) with t = n % T); tau by value or from tau_dev.  training = 0: hard arg-max of the logits, no noise.
 * fwd: out, idx [N, G] (code-book ROW g*V + v), avg [G][st5_vq_vpad()] (mean softmax, kept for bwd), perp2 = {code_perplexity,
 * prob_perplexity}.  bwd: dlogits [N, G*V] from dsel [N][dsel_ld >= G*vpad] (= dOut . vars_g^T per group, the caller's GEMM; NULL =
 * no code path), the upstream gradient of prob_perplexity (device scalar, NULL = 0); denc = (1 - w) dOut (NULL = skip). */
int64_t st5_vq_ws_bytes(void);
int32_t st5_vq_vpad(void);
int st5_vq_fwd(const float* logits, const float* gumbel, const float* vars, const void* enc, const float* mix_w, float tau, const float* tau_dev,
               int32_t training, void* out, int32_t* idx, float* avg, float* perp2, void* ws, int32_t N, int32_t G, int32_t V, int32_t Dg,
               int32_t T, int dtype, void* stream);
int st5_vq_bwd(const float* logits, const float* gumbel, const float* dsel, int32_t dsel_ld, const float* avg, const float* g_prob_perp,
               const void* dout, const float* mix_w, float tau, const float* tau_dev, int32_t training, float* dlogits, void* denc, int32_t N,
               int32_t G, int32_t V, int32_t Dg, int32_t T, int dtype, void* stream);
/* ---- HuBERT NCE head (speech_encoder_postnet.py:56-76), fp32 (csrc/nce.hip) ----
 * st5_norm_rows: y[r] = x[r] / max(|x[r]|, 1e-8) (x in `dtype`, y fp32), inv[r] = 1 / max(|x[r]|, 1e-8); st5_norm_rows_bwd: its
 * gradient dx (`dtype`; accumulate != 0: dx += ...).  st5_canon_rows: canon[c] = smallest c' whose row equals row c exactly.
 * st5_nce_logits: logits[s] = [sim[s, t_s], sim[s, :]] / temp with -inf where canon[c] == canon[t_s]; st5_nce_logits_bwd: dsim. */
int st5_norm_rows(const void* x, float* y, float* inv, int64_t rows, int32_t cols, int dtype, void* stream);
int st5_norm_rows_bwd(const float* y, const float* inv, const float* dy, void* dx, int64_t rows, int32_t cols, int32_t accumulate, int dtype,
                      void* stream);
int st5_canon_rows(const float* e, int32_t* canon, int32_t V, int32_t D, void* stream);
int st5_nce_logits(const float* sim, const int32_t* target, const int32_t* canon, float* logits, int64_t S, int32_t V, float temp, void* stream);
int st5_nce_logits_bwd(const float* dlogits, const int32_t* target, const int32_t* canon, float* dsim, int64_t S, int32_t V, float temp,
                       void* stream);
/* A/B switch of the LayerNorm backward: upper bound of its block count (default 256). */
int st5_layernorm_set_max_blocks(int n);
/* ---- CTC prefix scoring for joint CTC / attention beam search (sequence_generator.py:273-418 calls espnet's
 * CTCPrefixScore per hypothesis on the host; in-tree copy Speech2C/speech2c/models/modules/ctc_prefix_score.py:10-112) ----
 * x: fp32 CTC log-posteriors [T, V] of ONE utterance (device).  A state is r[T][2] fp32 = log r_t^n, log r_t^b of a prefix.
 * st5_ctc_initial_state: the state of the empty prefix (:27-39).
 * st5_ctc_prefix_score: for nh hypotheses with states r_prev[nh][T][2], last label last[nh] and out_len labels after <sos>,
 * and nc candidate next labels per hypothesis cs[nh][nc] (int64): log_psi[nh][nc] = log prefix probability of each extension
 * (eos: probability that the prefix ends; blank: logzero = -1e10), r_new[nh][nc][T][2] = the extensions' states. */
int st5_ctc_initial_state(const float* x, int32_t T, int32_t V, int32_t blank, float* r, void* stream);
int st5_ctc_prefix_score(const float* x, int32_t T, int32_t V, int32_t blank, int32_t eos, const float* r_prev, const int64_t* last,
                         int32_t out_len, const int64_t* cs, int32_t nh, int32_t nc, float* log_psi, float* r_new, void* stream);
/* ... over TWO gradient buffers: the step uses g + g2 (g2 may be NULL); zero_grads != 0 leaves both buffers zeroed.  With
 * st5_sumsq_pair (sum of squares of x + y) for the clipping norm this replaces "g += g2", two fills and the plain step. */
int st5_adam_step_pair(float* p, float* g, float* g2, int32_t zero_grads, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                       float eps, float weight_decay, int32_t step, const float* gnorm_sq, float max_norm, float grad_scale,
                       void* bf16_mirror, const float* hyper_dev, void* stream);
int st5_sumsq_pair(const float* x, const float* y, float* out, int64_t n, float scale, int32_t accumulate, void* stream);
/* Dropout seeds: every `seed` argument of this library may instead be a device pointer to the 64-bit seed, tagged with bit 63
 * (seed = (1 << 63) | pointer): the kernels then read the seed from memory (csrc/common.h resolve_seed).  Used by captured HIP
 * graphs, whose kernel arguments cannot change between replays. */
/* Batched bf16 matrix transposes in one launch.  jobs_dev: device array of {int64 src_off, int64 dst_off, int32 rows,
 * int32 cols, int32 tile0, int32 pad} sorted by tile0 (first 64x64 tile index of the job), offsets in elements into
 * src_flat / dst_flat; ntiles = total tile count.  Used for the transposed weight copies of the data-gradient GEMMs
 * (dX = dY.W as an NT product), refreshed once per optimizer step. */
int st5_multi_transpose_bf16(const void* src_flat, void* dst_flat, const void* jobs_dev, int32_t njobs, int32_t ntiles,
                             void* stream);

const char* st5_version(void);

#ifdef __cplusplus
}
#endif
#endif
