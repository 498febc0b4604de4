/*
 * timhip.h — C ABI of libtimhip.so: the MI355X (gfx950) implementation of the
 * TIM (time_interval_machine) encoder hot path, forward and backward.
 *
 * Drop-in boundary.  Each stage entry point replaces the
 * torch.nn call chain the reference runs for that stage:
 *
 *   timhip_layer_{fwd,bwd}     <- TransformerEncoderLayer.forward over nn.MultiheadAttention
 *                                 .../models/helpers/transformers.py:92-111 with the mask of
 *                                 recognition/.../models/tim.py:161-166 (det tim.py:320-325,384-389)
 *   timhip_gemm_nt, timhip_wgrad <- nn.Linear forward / input-gradient / weight-gradient:
 *                                 tim.py:66-74 (time MLP), encodings.py:21-26,140-153 (embedders),
 *                                 head.py:8-15 (CLS heads), det head.py:99-114 (regression heads)
 *   timhip_layernorm_{fwd,bwd} <- nn.LayerNorm (+ the ReLU/GELU in front of it) tim.py:72-73,
 *                                 encodings.py:23-25
 *   timhip_time_l1_{fwd,bwd}   <- the K=2 first Linear+ReLU of TIM.time_mlp, tim.py:67-68
 *   timhip_assemble_{fwd,bwd}  <- *FeatureEncoding.forward concat/add/dropout, encodings.py:190-250
 *   timhip_gather_rows / _scatter_rows_add <- the tail slicing of *CLSHead.forward, head.py:18-36
 *   timhip_cast_weight         <- (no reference counterpart) operand-dtype working copies of the
 *                                 fp32 master weights, plain and transposed, once per optimizer step.
 *
 * Conventions
 *   - Plain C: pointers and sizes only.  Every device buffer is allocated and owned by the caller;
 *     the library never allocates device memory and keeps no pointer after the call returns
 *     (re-entrant; forward is called from the Python main thread and backward from PyTorch's
 *     autograd thread).  Exceptions, all opt-in: two process-wide switches - the dropout salt
 *     pointer registered by timhip_dropout_salt() (HIP-graph replay) and the measurement hooks
 *     timhip_gemm_timing_start/stop() (bench.py) - and timhip_softnms_1d(), which synchronises its
 *     stream once (it returns the number of kept segments to the host).
 *   - All work is enqueued on the `stream` argument (a hipStream_t passed as void*).
 *   - Return value: 0 on success, a negative TIMHIP_E* code otherwise (timhip_strerror()).
 *     No exception crosses the ABI.
 *   - Layout: batch-first.  A window is S = F + Q token rows (F feature tokens first, then Q query
 *     tokens); activations are row-major [B*S, E].  "Operand dtype" T is bf16 for
 *     TIMHIP_PREC_BF16, fp16 for TIMHIP_PREC_F16 and fp32 for TIMHIP_PREC_FP32 / _BF16X3 (which splits on the fly).  Operand matrices are K-contiguous
 *     with a leading dimension that is a multiple of 64 elements, zero padded.
 */
#ifndef TIMHIP_H
#define TIMHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TIMHIP_VERSION 6   /* 6 (round 6): timhip_timing_stop_families; 5 (round 5): timhip_assemble_{fwd,bwd}_p (token / modality vectors by pointer), timhip_dx_init_slabs, timhip_det_side_loss_{fwd,bwd}, timhip_sigmoid_bwd_rows, timhip_time_l1_fwd_split3, timhip_gather_split3_ranges, TIMHIP_EPI_RELU_SPLIT3_T, timhip_layernorm_{fwd,bwd}2, timhip_cast_rows_pair; 4 (round 4): 8-word timhip_grad_scale block + non-finite flag, TIMHIP_DESC_STREAM16*, timhip_dx_init, timhip_reload_env */

enum {
  TIMHIP_OK = 0,
  TIMHIP_EINVAL = -1,       /* null pointer / bad descriptor field */
  TIMHIP_EUNSUPPORTED = -2, /* shape outside what the kernels handle */
  TIMHIP_EWORKSPACE = -3,   /* workspace too small */
  TIMHIP_ELAUNCH = -4,      /* hipGetLastError() reported a launch failure */
  TIMHIP_EALIGN = -5        /* pointer or leading dimension not aligned */
};

/* TIMHIP_PREC_F16: fp16 MFMA operands (v_mfma_f32_32x32x16_f16: the bf16 rate, 3 more mantissa bits), fp32 accumulation -
 * the arithmetic of the reference's own GPU recipe (torch.cuda.amp fp16 autocast, recognition/scripts/train.py:82,197).
 * Gradient operands are stored multiplied by a power of two (TimDesc.grad_scale; the weight-gradient outputs undo it). */
enum { TIMHIP_PREC_BF16 = 0, TIMHIP_PREC_BF16X3 = 1, TIMHIP_PREC_FP32 = 2, TIMHIP_PREC_F16 = 3 };

/* epilogue selector of the generic GEMM entry point (unit tests, heads) */
enum {
  TIMHIP_EPI_STORE_T = 0,   /* out0(T)   = acc + bias */
  TIMHIP_EPI_RELU_T = 1,    /* out0(T)   = relu(acc + bias) */
  TIMHIP_EPI_STORE_F32 = 2, /* out0(f32) = acc + bias */
  TIMHIP_EPI_GELU_DROP_T2 = 3, /* out1(T) = acc+bias ; out0(T) = drop(gelu(out1)) */
  TIMHIP_EPI_DROP_RES_F32 = 4, /* out0(f32) = res + drop(acc + bias) */
  TIMHIP_EPI_ADD_F32 = 5,   /* out0(f32) = acc + res (res may be null) */
  TIMHIP_EPI_DGELU_T = 6,   /* out0(T)   = acc * dropmask * gelu'(aux(T)) */
  TIMHIP_EPI_DRELU_T = 7,   /* out0(T)   = acc * (aux(T) > 0) */
  TIMHIP_EPI_ATOMIC_F32 = 8,/* out0(f32) += acc (split-K weight gradients) */
  TIMHIP_EPI_SIGMOID_F32 = 9,/* out0(f32) = sigmoid(acc + bias) */
  TIMHIP_EPI_DRELU_F32IN_T = 10,/* out0(T) = acc * (aux(f32) > 0) */
  TIMHIP_EPI_GELU_DROP_G2 = 11, /* u = acc+bias ; out0(T) = dropmask * gelu(u) ; out1(T) = dropmask * gelu'(u): the factor the
                                   backward multiplies with, so that its epilogue (TIMHIP_EPI_MULAUX_T) needs no erf / exp and
                                   no second look at the dropout mask */
  TIMHIP_EPI_MULAUX_T = 12,     /* out0(T) = acc * aux(T) */
  TIMHIP_EPI_RELU_SPLIT3_T = 13 /* v = relu(acc + bias) as the split operand [hi | lo | hi]: out0(T)[m, n] = T(v), [m, ld1 + n] = T(v - hi),
                                   [m, 2 ld1 + n] = T(v); ld1 = the block width (a multiple of 64), ld0 = the row stride (3 ld1) - what
                                   timhip_split3_many (mode 0, relu) makes of the fp32 output, without the round trip (time MLP) */
};

/* Shape of one call.  M = B*S rows flow through the encoder. */
typedef struct TimDesc {
  int32_t B;         /* windows in this batch */
  int32_t S;         /* tokens per window = F + Q */
  int32_t F;         /* feature tokens per window (keys every token may attend to) */
  int32_t d;         /* d_model */
  int32_t E;         /* transformer width = 2*d_model */
  int32_t H;         /* heads */
  int32_t FF;        /* feed-forward width */
  int32_t precision; /* TIMHIP_PREC_* */
  float p_drop;      /* encoder dropout probability; 0 => evaluation mode */
  uint64_t seed;     /* Philox key for this step */
  int32_t layer;     /* layer index (part of the Philox stream id) */
  int32_t reserved;  /* flags: TIMHIP_DESC_* (0 by default) */
  const float* grad_scale; /* backward of TIMHIP_PREC_F16: device {S, 1/S} as timhip_grad_scale writes them (NULL: S = 1).
                              fp16 gradient operands (the T copies LayerNorm-backward writes, and everything the data chain
                              derives from them) are stored multiplied by S; the fp32 gradient stream, the weight, bias
                              and LayerNorm-parameter gradients are true-scale.  Ignored by the other precisions' forward */
} TimDesc;
/* TimDesc.reserved flags */
#define TIMHIP_DESC_ATTN_FP32 1        /* run the attention in fp32 arithmetic (reference kernels) */
#define TIMHIP_DESC_ATTN_BWD_ONE_KERNEL 2 /* single-kernel MFMA attention backward */
#define TIMHIP_DESC_WGRAD_OVERWRITE 4  /* bf16 only: timhip_layer_bwd[_weights] WRITES the weight and bias gradients of the four
                                          Linears (dW = ..., not +=): those buffers need no zero fill and are not read */
#define TIMHIP_DESC_WGRAD_SEPARATE 8   /* bf16 only: four timhip_wgrad launches per layer instead of the grouped one (A/B knob) */
#define TIMHIP_DESC_OUTPROJ_SPLIT 16   /* forward, 16-bit precisions: TimLayerParams.out_w is a [E, ld >= 2E] SPLIT copy of the out-projection
                                          weight, column blocks [hi | lo | ...] as timhip_split3_many(mode 0) writes them; the
                                          out-projection runs over K = 2E with the attention output read twice (TimEpi.a_wrap_k):
                                          its weight's fp16 rounding error - the same for every token, hence not averaged out by
                                          the attention - was the largest single term of the fp16 mode's logit error (DESIGN.md
                                          section 6).  out_wt (the backward's operand) stays the plain transposed copy. */
#define TIMHIP_DESC_INPROJ_SPLIT 32    /* the same for in_w ([3E, ld >= 2E]), l1_w ([FF, ld >= 2E]) and l2_w ([E, ld >= 2FF]): with all four */
#define TIMHIP_DESC_L1_SPLIT 64        /* set every forward GEMM of the layer carries its weight to ~22 bits at twice the matrix */
#define TIMHIP_DESC_L2_SPLIT 128       /* work (an opt-in margin mode, tim_amd: TIM_AMD_SPLIT_LAYER_WEIGHTS=all); default: out only */
/* Backward of TIMHIP_PREC_F16 (grad_scale set), timhip_layer_bwd_split / _data_split: the RESIDUAL part of the gradient stream in
 * the operand dtype, times the gradient scale - like its branch parts (dx_*_add) and the gradient operands - instead of fp32.
 *   STREAM16      inside the layer (between the two LayerNorm-backward launches);
 *   STREAM16_IN   dx_out points at a T [M, E] matrix (what the layer above wrote under STREAM16_OUT), not at floats;
 *   STREAM16_OUT  dx_in is written as such a T matrix (requires dx_in_add: the layer below adds the two as it reads).
 * 120 instead of 160 MB per LayerNorm-backward launch at C2a; parameter gradients move by <= 1.5e-3 of their largest element. */
#define TIMHIP_DESC_STREAM16 0x10000
#define TIMHIP_DESC_STREAM16_IN 0x20000
#define TIMHIP_DESC_STREAM16_OUT 0x40000
/* ABI 6: the layer's saved block carries the keep-bits of its attention dropout (TIMHIP_SAVED_ATTN_KEEP_BITS), drawn ahead of the
 * layer by timhip_attn_keep_bits(): the attention forward and the fused attention backward read them instead of running Philox
 * for every (row, key) in both directions.  Set on the forward AND the backward desc of a step, or on neither. */
#define TIMHIP_DESC_ATTN_KEEP_BITS 0x80000

/* One encoder layer.  *_op are operand-dtype working copies made by timhip_prepare_weights:
 * w (as stored, [N,K]) and wt (transposed, [K,N]).  Biases and LayerNorm parameters are the
 * fp32 master tensors. */
typedef struct TimLayerParams {
  const void *in_w, *in_wt;   /* [3E,E] / [E,3E]   self_attn.in_proj_weight */
  const void *out_w, *out_wt; /* [E,E]             self_attn.out_proj.weight */
  const void *l1_w, *l1_wt;   /* [FF,E] / [E,FF]   linear1.weight */
  const void *l2_w, *l2_wt;   /* [E,FF] / [FF,E]   linear2.weight */
  const float *in_b, *out_b, *l1_b, *l2_b;
  const float *n1_w, *n1_b, *n2_w, *n2_b;
} TimLayerParams;

/* fp32 gradient accumulators, same shapes as the master parameters (+=). */
typedef struct TimLayerGrads {
  float *in_w, *in_b, *out_w, *out_b, *l1_w, *l1_b, *l2_w, *l2_b;
  float *n1_w, *n1_b, *n2_w, *n2_b;
  /* optional (NULL = reduce inside the layer call): timhip_layer_ln_partial_bytes() bytes that receive the per-block partial
   * sums of the two LayerNorm parameter gradients (norm2 first, then norm1) INSTEAD of adding them into n*_w / n*_b; the
   * caller reduces the partials of all its layers with one timhip_ln_partials_reduce launch (12 small launches -> 1) */
  float* ln_partials;
} TimLayerGrads;
size_t timhip_layer_ln_partial_bytes(const TimDesc* d);
/* dgamma[i] += column sums of set i's gamma partials, dbeta[i] likewise: nsets sets laid out back to back, each as written by one
 * LayerNorm backward over `rows` rows of `cols` columns; dgamma / dbeta: HOST arrays of nsets device pointers (<= 16) */
int timhip_ln_partials_reduce(const float* partials, int nsets, int rows, int cols, float* const* dgamma,
                              float* const* dbeta, void* stream);

int timhip_version(void);
/* The launchers' A/B knobs (TIMHIP_* environment variables, tim_amd/csrc/common.h: TimKnobs) are read once, at the first
 * launch.  Test hook: read them again after changing the environment inside a process. */
void timhip_reload_env(void);
/* ABI 6 (tests / tools): the row tile of the eight-phase NT kernel (gemm_nt_p8_kernel: 256 x 256 / 320 x 256 tiles for the shapes
 * that run more than one round of 160 x 256 tiles) a plain timhip_gemm_nt launch of this epilogue and shape takes - 8 (256 rows),
 * 10 (320 rows) or 0 (another kernel); follows TIMHIP_GEMM_P8 (0 off, 1 by shape, 8 / 10 forced). */
int timhip_gemm_p8_choice(int epi, int M, int N, int K);
/* bit 0: the library was built with TUNING=1 (carries the measured-slower kernel variants and the ablation hooks) */
int timhip_build_flags(void);
const char* timhip_strerror(int code);

/* bytes of the per-layer saved-for-backward block and of the scratch workspace */
size_t timhip_layer_saved_bytes(const TimDesc* d);
size_t timhip_layer_workspace_bytes(const TimDesc* d);
/* Test hook: byte offset and size of one field of the saved block (opaque to the product's host code).  Fields: the packed
 * in-projection output qkv [M,3E] (T), the attention output o [M,E] (T), the pre-norm sums y1 / y2 [M,E] fp32, the operand copy
 * of norm1's output x1 [M,E] (T), the FFN hidden activations h [M,FF] (T: dropmask * gelu(linear1)), and the FFN dropout
 * keep-bits [M, FF/8] bytes that LayerNorm-1 draws (bit c%8 of byte [r*FF/8 + c/8] = element (r,c) kept; only written when
 * p_drop > 0). */
enum { TIMHIP_SAVED_QKV = 0, TIMHIP_SAVED_O = 1, TIMHIP_SAVED_Y1 = 2, TIMHIP_SAVED_X1T = 3, TIMHIP_SAVED_H = 4,
       TIMHIP_SAVED_Y2 = 5, TIMHIP_SAVED_FFN_KEEP_BITS = 6, TIMHIP_SAVED_ATTN_KEEP_BITS = 7 };
int timhip_layer_saved_field(const TimDesc* d, int field, size_t* offset, size_t* bytes);

/* ---------------------------------------------------------------- weights ---- */
/* dst[rows, ld] (T) = cast(src[rows, cols] fp32), zero padded to ld columns.
 * transpose != 0: dst[cols, ld] = src^T.  ld % 64 == 0. */
int timhip_cast_weight(int precision, const float* src, int rows, int cols, void* dst, int ld,
                       int transpose, void* stream);

/* both operand copies of one fp32 weight [rows, cols] in a single pass:
 * plain[rows, ldp] = cast(src), tr[cols, ldt] = cast(src^T), zero padded; ldp, ldt multiples of 64 */
int timhip_cast_weight_both(int precision, const float* src, int rows, int cols, void* plain, int ldp,
                            void* tr, int ldt, void* stream);

/* the same for n weights in one launch (the refresh of every operand copy after an optimizer step);
 * items is a HOST array, copied into the kernel arguments */
typedef struct TimCastItem {
  const float* src; /* fp32 master [rows, cols], 16-byte aligned */
  void* plain;      /* [rows, ldp] */
  void* tr;         /* [cols, ldt] */
  int32_t rows, cols, ldp, ldt;
} TimCastItem;
int timhip_cast_weights(int precision, const TimCastItem* items, int n, void* stream);

/* ---------------------------------------------------------------- generic ops (also unit-test hooks) */
typedef struct TimEpi {
  void* out0;
  void* out1;
  const float* bias;
  const float* res;
  const void* aux;
  int32_t ld0, ld1, ldres, ldaux;
  float p_drop;
  uint32_t site;   /* Philox stream id of the dropout site */
  uint64_t seed;
  const void* mask; /* optional: the site's keep-bits drawn ahead of time (bit c%8 of byte [r*ldmask + c/8] = element (r,c)
                       kept; what timhip_layer_fwd has LayerNorm-1 write for the FFN dropout).  NULL: drawn in the epilogue */
  int32_t ldmask;   /* row stride of mask in bytes */
  int32_t reserved; /* 0, or the operand replication factor of a split-operand product (3 for timhip_split3_many operands):
                       only divides the FLOP count the timing hooks attribute to the launch */
  /* TIMHIP_EPI_DROP_RES_F32 only, optional: the residual is LayerNorm(res).  res then holds the PRE-norm fp32 rows, ln_stats
   * the (mean, rstd) pair of every row as timhip_layernorm_fwd writes them, ln_w / ln_b the affine parameters; the epilogue
   * normalises what it reads, so the normalised fp32 rows need not exist in memory.  NULL: res is used as it is. */
  const float* ln_stats;
  const float* ln_w;
  const float* ln_b;
  /* optional device scalar: the accumulators are multiplied by *acc_scale before the epilogue (NULL = 1).  TIMHIP_PREC_F16
   * keeps its gradient OPERANDS multiplied by a power of two S (timhip_grad_scale); an input-gradient GEMM whose result joins
   * the fp32 gradient stream (TIMHIP_EPI_ADD_F32) passes 1/S here. */
  const float* acc_scale;
  /* optional (0 = off): the A operand has only a_wrap_k columns and is read TWICE along the contraction, K = 2 * a_wrap_k
   * (a multiple of 64; lda >= a_wrap_k): C = [A | A] B^T.  With B = [w_hi | w_lo] (the first two column blocks of a
   * timhip_split3_many mode-0 copy of an fp32 weight) the product carries the weight to ~22 bits at twice the matrix work and
   * no extra activation traffic - what TIMHIP_DESC_OUTPROJ_SPLIT uses.  16-bit precisions only. */
  int32_t a_wrap_k;
  int32_t reserved2;
} TimEpi;

/* n <= 6 independent small problems with the same epilogue as ONE launch (bf16; TIMHIP_EPI_STORE_F32, _ADD_F32, _STORE_T,
 * _RELU_T): what the classification heads use - four under-filled GEMMs each way (head.py:17-38).  items is a HOST array. */
typedef struct TimGemmItem {
  const void* A; const void* B;
  int32_t lda, ldb, M, N, K, reserved;   /* reserved: as TimEpi.reserved */
  TimEpi e;
} TimGemmItem;
int timhip_gemm_nt_group(int precision, int epi, const TimGemmItem* items, int n, void* stream);

/* C[M,N] = A[M,K] * B[N,K]^T through epilogue `epi` (TIMHIP_EPI_*).  A, B operand dtype,
 * lda/ldb multiples of 64 and >= K.  splitk > 1 only with TIMHIP_EPI_ATOMIC_F32. */
int timhip_gemm_nt(int precision, int epi, const void* A, int lda, const void* B, int ldb, int M,
                   int N, int K, const TimEpi* e, int splitk, void* stream);

/* dst[cols, ld] (T) = src[rows, lds]^T (T); pad columns rows..ld are zeroed.  */
int timhip_transpose(int precision, const void* src, int rows, int cols, int lds, void* dst, int ld,
                     void* stream);

/* out[n] += sum_m src[m, n]  (T source, fp32 accumulate): bias gradients */
int timhip_colsum(int precision, const void* src, int rows, int cols, int ld, float* out,
                  void* stream);

/* Split operands: src_i [rows_i, cols_i] fp32 (row stride lds_i) -> dst_i [rows_i, ldd_i] (T = bf16 / fp16), ldd_i = 3 * block
 * width, block width = cols_i rounded up to a multiple of 64 (zero padded).  With hi = T(x), lo = T(x - hi) the three column
 * blocks are [hi | lo | hi] (mode 0: activations; relu != 0 applies max(x, 0) first) or [hi | hi | lo] (mode 1: weights), so a
 * 16-bit NT GEMM of a mode-0 matrix with a mode-1 matrix over K = ldd computes x w^T to ~22 bits (x_hi w_hi + x_lo w_hi +
 * x_hi w_lo).  count <= 6 matrices per launch; HOST arrays.  What TIMHIP_PREC_F16 models use for the time MLP and the
 * classification heads (the two small sites that dominate the 16-bit error budget). */
int timhip_split3_many(int precision, int count, const float* const* src, const int* rows, const int* cols, const int* lds,
                       void* const* dst, const int* ldd, int mode, int relu, void* stream);

/* Data-parallel gradient exchange (tim_amd/dp.py; replaces the DistributedDataParallel wrap of models/build.py:58-63): after
 * the all-to-all, recv [world][per] (bf16 if wire_bf16 else fp32) holds this rank's chunk of every rank's bucket;
 * out[per] (same dtype) = scale * sum over ranks, accumulated in fp32.  per % 4 == 0, 16-byte aligned buffers. */
int timhip_dp_reduce(int wire_bf16, const void* recv, int world, long long per, float scale, void* out, void* stream);

/* Gradient scale of the fp16 mode.  fp16 has 5 exponent bits: gradient operands would underflow (the reference's GPU recipe
 * wraps its step in a GradScaler for that reason, recognition/scripts/train.py:82,355-363).  Here the scale is chosen per
 * backward pass ON THE DEVICE from the cotangents entering it: S = 2^floor(log2(target / max|cot|)), out[0] = S,
 * out[1] = 1/S (S = 1 when every cotangent is 0).  cot / counts: HOST arrays of n <= 8 device pointers / element counts;
 * out: 8 floats in device memory, out[2..7] zero before the first call (out[2..3] scratch, left zero).  One launch, no host sync.
 *
 * NON-FINITE WATCH (out[4], read as uint32): every entry point below that takes `out_scale` expects NULL or &out[1] of such a
 * block, multiplies what it writes by out[1] = 1/S, and ORs 1 into out[4] when a FINAL fp32 gradient it writes is inf or nan
 * (timhip_wgrad, timhip_wgrad_group, timhip_time_l1_bwd - every weight / bias gradient of the fp16 backward).  An overflow of
 * a 16-bit gradient operand anywhere in the chain reaches every weight gradient upstream of it, so the flag is what the
 * reference's GradScaler inf check computes (recognition/scripts/train.py:351,357-363: skip the step, halve the scale)
 * without a pass over the gradients and without a host synchronisation until somebody reads the word. */
int timhip_grad_scale(const float* const* cot, const long long* counts, int n, float target, float* out, void* stream);

/* fp32 -> T with optional dropout (p_drop > 0) and zero padding to ld; scale: optional device scalar multiplied in */
int timhip_cast_rows(int precision, const float* src, int rows, int cols, int lds, void* dst, int ld,
                     float p_drop, uint64_t seed, uint32_t site, const float* scale, void* stream);
/* timhip_cast_rows for TWO contiguous fp32 matrices with the same row count in one launch (the two embedders' inputs: widths
 * cols[i], operand rows of stride ld[i], dropout site sites[i]; same masks as two timhip_cast_rows calls) */
int timhip_cast_rows_pair(int precision, const float* const* src, const int* cols, void* const* dst, const int* ld, int rows,
                          float p_drop, uint64_t seed, const uint32_t* sites, void* stream);

/* LayerNorm over the last dim of act(y): x = LN(act(y)) * w + b.
 * act: 0 none, 1 relu, 2 gelu(erf).  Writes x_f32 (may be null, row stride ldx, column
 * offset already applied by the caller), x_T (may be null, ld ldt) and stats[rows,2]=(mean,rstd).
 * p_drop>0 applies (sequence) dropout to the outputs. */
int timhip_layernorm_fwd(int precision, const float* y, int rows, int cols, int ldy, int act,
                         const float* w, const float* b, float* x_f32, int ldx, void* x_T, int ldt,
                         float* stats, void* stream);
/* dy = LN'(dx) (through act'); dgamma += , dbeta +=.  dy_f32 (may be null) and
 * dy_T = dropmask(site) * dy (may be null). */
int timhip_layernorm_bwd(int precision, const float* dx, int lddx, const float* y, int ldy,
                         const float* stats, int rows, int cols, int act, const float* w,
                         float* dy_f32, int lddy, void* dy_T, int ldt, float p_drop, uint64_t seed,
                         uint32_t site, float* dgamma, float* dbeta, const float* t_scale, void* stream);
/* Two LayerNorms of the same width and activation over STACKED rows in one launch (round 5: the two modality embedders,
 * encodings.py:21-26): rows [0, split_row) use (w, b), rows from split_row on (w2, b2); in the backward the parameter gradients of
 * the two halves go to (dgamma, dbeta) and (dgamma2, dbeta2) - accumulated, the caller zeroes them - and split_row must be a
 * multiple of 16 (a block of the backward owns 16 consecutive rows).  Everything else as timhip_layernorm_fwd / _bwd. */
int timhip_layernorm_fwd2(int precision, const float* y, int rows, int cols, int ldy, int act, const float* w, const float* b,
                          int split_row, const float* w2, const float* b2, float* x_f32, int ldx, void* x_T, int ldt, float* stats,
                          void* stream);
int timhip_layernorm_bwd2(int precision, const float* dx, int lddx, const float* y, int ldy, const float* stats, int rows, int cols,
                          int act, const float* w, int split_row, const float* w2, float* dy_f32, int lddy, void* dy_T, int ldt,
                          float* dgamma, float* dbeta, float* dgamma2, float* dbeta2, const float* t_scale, void* stream);

/* structured attention over qkv[B*S, 3E] (T): token i attends to the F feature tokens and to
 * itself.  o[B*S,E] (T), lse[B,H,S] fp32. */
int timhip_attention_fwd(const TimDesc* d, const void* qkv, void* o, float* lse, void* stream);
/* ABI 6: keep-bits of the attention dropout of `nlayers` encoder layers in ONE launch, written into the layers' saved blocks
 * (saved[l] = the block timhip_layer_fwd of layer l will fill; field TIMHIP_SAVED_ATTN_KEEP_BITS: per (window, head, token row)
 * two 64-bit words, word g bit 4 c + t = key 8 c + 4 g + t kept, keys 0 .. 127 - the order in which the MFMA attention
 * kernels' lanes own keys).  The same Philox stream as the kernels' own draws (seed d->seed, site of layer l, element
 * ((b H + h) S + s) LP + key, LP = F + 1 rounded up to 8): results are bit-identical with and without the flag.
 * TIMHIP_EUNSUPPORTED (the caller leaves TIMHIP_DESC_ATTN_KEEP_BITS off): everything but 16-bit precisions with 128-wide heads and
 * 97 .. 128 feature keys (the geometry whose kernels read the bits), and evaluation mode (p_drop = 0). */
int timhip_attn_keep_bits(const TimDesc* d, int nlayers, void* const* saved, void* stream);
int timhip_attention_bwd(const TimDesc* d, const void* qkv, const void* o, const float* lse,
                         const void* d_o, void* dqkv, void* workspace, size_t workspace_bytes,
                         void* stream);
size_t timhip_attention_bwd_workspace_bytes(const TimDesc* d);

/* dW[Nout,Kout] += dY[M,Nout]^T X[M,Kout] and (db != NULL) db[Nout] += colsum(dY): weight/bias
 * gradients of one nn.Linear.  workspace (timhip_wgrad_workspace_bytes) holds the transposed
 * operand copies and the split-K fp32 partial slabs. */
size_t timhip_wgrad_workspace_bytes(int precision, int Nout, int Kout, int M);
int timhip_wgrad(int precision, const void* dY, int ldy, int Nout, const void* X, int ldx, int Kout,
                 int M, float* dW, float* db, void* workspace, size_t workspace_bytes, const float* out_scale, void* stream);

/* The same for n <= 8 Linear layers that share M (the four of one encoder layer), as ONE launch: their tile lists are
 * concatenated and the number of splits of the contraction is chosen for the total - at E = 1024, FF = 2048 the layer has
 * exactly 512 tiles = the chip's 512 block slots, the contraction is not split and dW / db are written directly, with no
 * partial slabs and no reduce.  accumulate = 0 writes dW / db instead of adding to them.  bf16 operands only (the fp32 and
 * bf16x3 modes go through timhip_wgrad); every Nout*Kout must be a multiple of 4.  items is a HOST array, copied into the
 * kernel arguments.  workspace may be NULL when timhip_wgrad_group_workspace_bytes returns 0. */
typedef struct TimWgradItem {
  const void* dY;   /* [M, ldy] */
  const void* X;    /* [M, ldx] */
  float* dW;        /* [Nout, Kout] */
  float* db;        /* [Nout] or NULL */
  int32_t ldy, ldx, Nout, Kout;
} TimWgradItem;
size_t timhip_wgrad_group_workspace_bytes(int precision, const TimWgradItem* items, int n, int M);
int timhip_wgrad_group(int precision, const TimWgradItem* items, int n, int M, int accumulate, void* workspace,
                       size_t workspace_bytes, const float* out_scale, void* stream);

/* dx[r,c] = g[r,c] * keep-mask/(1-p): backward of the feature dropout applied by timhip_cast_rows */
int timhip_dropout_rows_bwd(const float* g, int rows, int cols, int ldg, float* dx, int ldx,
                            float p_drop, uint64_t seed, uint32_t site, void* stream);

/* time MLP layer 1 (K = 2, tim.py:67): h[r,j] = relu(t[r,0] w[j,0] + t[r,1] w[j,1] + b[j]) (T, ld) */
int timhip_time_l1_fwd(int precision, const float* times, int rows, int d, const float* w,
                       const float* b, void* h, int ld, void* stream);
/* The same with the row written as the split operand [hi | lo | hi] of the fp32 value (three 16-bit column blocks of width ld, a
 * multiple of 64; row stride 3 ld; 16-bit modes): what timhip_split3_many (mode 0) makes of the fp32 output, without the fp32
 * round trip and the second launch. */
int timhip_time_l1_fwd_split3(int precision, const float* times, int rows, int d, const float* w, const float* b, void* h3,
                              int ld, void* stream);
/* dh: gradient w.r.t. the pre-activation of layer 1 (T).  dw[d,2] +=, db[d] +=, dt[rows,2] = (may be NULL) */
int timhip_time_l1_bwd(int precision, const float* times, int rows, int d, const float* w,
                       const void* dh, int ld, float* dw, float* db, float* dt, const float* out_scale, void* stream);

/* keep-mask (1/0 bytes) of a dropout site, as the kernels generate it: test hook.  Element (r, c) has linear index
 * r * cols + c; one Philox-4x32-7 call (key = seed, stream = site, counter = index / 8) decides 8 consecutive elements by its
 * eight 16-bit halves: kept iff halfword >= floor(p * 65536). */
int timhip_dropout_mask(uint64_t seed, uint32_t site, float p, int rows, int cols, uint8_t* out,
                        void* stream);

/* Graph-safe dropout.  Every entry point that takes a `seed` passes it to its kernels by value, so a step captured in a
 * HIP graph would replay the masks of the captured step.  After timhip_dropout_salt(p) (p: one 64-bit word in device
 * memory, owned by the caller, alive while registered) every dropout kernel launched from this library draws with
 * seed + *p, read on the device when the kernel runs: bump the word between replays (or with a node inside the graph) and
 * forward and backward of one replay still agree.  NULL restores plain launch-time seeds.  Process-wide (one process per
 * GPU).  The reference has no counterpart: torch's CUDA-graph-safe Philox offsets play this role there. */
int timhip_dropout_salt(const unsigned long long* dev_salt);

/* ---------------------------------------------------------------- stages ---- */
/* One post-norm encoder layer.  x_in (fp32 [M,E]) and x_in_T (T [M,E]) are the layer input,
 * x_out / x_out_T the output.  `saved` (timhip_layer_saved_bytes) is read back by the backward.
 * x_out may be NULL when nobody reads the fp32 output rows (the next layer then runs timhip_layer_fwd_chained).
 * workspace / workspace_bytes are not used any more (kept for binary compatibility). */
int timhip_layer_fwd(const TimDesc* d, const TimLayerParams* w, const float* x_in,
                     const void* x_in_T, float* x_out, void* x_out_T, void* saved, void* workspace,
                     size_t workspace_bytes, void* stream);
/* The same for a layer that follows another one of the same shape: its fp32 input rows are not read from memory but
 * recomputed where they are needed (the residual of the out-projection epilogue) as LayerNorm-2 of the previous layer's
 * pre-norm rows, taken from prev_saved with prev_w's norm2 parameters; x_in_T is still the previous layer's x_out_T.
 * Together with x_out = NULL on the inner layers the fp32 stream between layers never exists in memory. */
int timhip_layer_fwd_chained(const TimDesc* d, const TimLayerParams* w, const TimLayerParams* prev_w,
                             const void* prev_saved, const void* x_in_T, float* x_out, void* x_out_T, void* saved,
                             void* stream);
/* dx_out: gradient w.r.t. the layer output (fp32 [M,E], clobbered).  dx_in: gradient w.r.t. the
 * layer input (fp32 [M,E]).  Parameter gradients are accumulated (+=) into *g. */
int timhip_layer_bwd(const TimDesc* d, const TimLayerParams* w, const void* x_in_T,
                     const void* saved, float* dx_out, float* dx_in, const TimLayerGrads* g,
                     void* workspace, size_t workspace_bytes, void* stream);

/* Two-stream form of the layer backward.  The data chain (LayerNorm / GELU / attention backward and
 * the input-gradient GEMMs) produces dx_in and leaves the four gradient operands df, du, da, dqkv in
 * `dy` (timhip_layer_dy_bytes); the weight-gradient part consumes `dy` and the saved activations and
 * may be enqueued on a second stream so that it overlaps the data chain of the next layer
 * (the caller orders the two streams with events).  LayerNorm gradients are written by the data chain,
 * all other parameter gradients by the weights part. */
size_t timhip_layer_dy_bytes(const TimDesc* d);
size_t timhip_layer_data_workspace_bytes(const TimDesc* d);
size_t timhip_layer_wgrad_workspace_bytes(const TimDesc* d);
int timhip_layer_bwd_data(const TimDesc* d, const TimLayerParams* w, const void* saved, float* dx_out,
                          float* dx_in, void* dy, const TimLayerGrads* g, void* workspace,
                          size_t workspace_bytes, void* stream);
int timhip_layer_bwd_weights(const TimDesc* d, const void* x_in_T, const void* saved, const void* dy,
                             const TimLayerGrads* g, void* workspace, size_t workspace_bytes,
                             void* stream);
/* Round 6: the weight gradients of TWO layers (a, b: the arguments of timhip_layer_bwd_weights for each) as one grouped launch.
 * A C2a layer's four products (transformers.py:73,102,107 under autograd) are 128 tiles of 256 x 256: two layers fill the 256
 * CUs with one tile each, every block running the whole contraction.  timhip_layer_wgrad_pair_wins(d) = 1 when that is the
 * case for this descriptor - then a host defers a layer's weight gradients until its neighbour's data chain has run (each
 * layer's `dy` block and saved activations stay untouched until the pair call) - 0 otherwise (pairing gains nothing; the pair
 * call may then return TIMHIP_EWORKSPACE, its workspace being sized for one layer). */
int timhip_layer_bwd_weights_pair(const TimDesc* d, const void* x_in_T_a, const void* saved_a, const void* dy_a,
                                  const TimLayerGrads* ga, const void* x_in_T_b, const void* saved_b, const void* dy_b,
                                  const TimLayerGrads* gb, void* workspace, size_t workspace_bytes, void* stream);
int timhip_layer_wgrad_pair_wins(const TimDesc* d);

/* Split gradient stream between layers (optional, what tim_amd/functional.py uses inside the stack).  The gradient of a layer
 * boundary travels as an fp32 part plus an operand-dtype part: dx = dx_f32 + (1/S) * dx_add, S the fp16 gradient scale of
 * TimDesc.grad_scale (1 in the other modes).  dx_out_add (may be NULL: the gradient entering the stack is plain fp32) is added
 * by the layer's norm2 backward as it reads its input; with dx_in_add != NULL the layer leaves its input gradient in the same
 * split form - dx_in receives norm1's fp32 output gradient, dx_in_add [M,E] (T) the in-projection's input-gradient product -
 * for the next call's dx_out / dx_out_add; with dx_in_add == NULL dx_in is the complete fp32 gradient (as timhip_layer_bwd).
 * Inside the layer the FFN branch's input gradient is handed to norm1's backward the same way.  The "+ residual" epilogues of
 * the two N = E input-gradient GEMMs disappear: they store 2 bytes per element instead of reading 4 and writing 4.  dx_out is
 * NOT clobbered by these forms. */
int timhip_layer_bwd_split(const TimDesc* d, const TimLayerParams* w, const void* x_in_T, const void* saved, const float* dx_out,
                           const void* dx_out_add, float* dx_in, void* dx_in_add, const TimLayerGrads* g, void* workspace,
                           size_t workspace_bytes, void* stream);
int timhip_layer_bwd_data_split(const TimDesc* d, const TimLayerParams* w, const void* saved, const float* dx_out,
                                const void* dx_out_add, float* dx_in, void* dx_in_add, void* dy, const TimLayerGrads* g,
                                void* workspace, size_t workspace_bytes, void* stream);

/* ---- front end --------------------------------------------------------------
 * The time MLP (tim.py:66-74), the modality embedders and the sequence assembly
 * (encodings.py:41-75,102-121,181-251) are sequenced by the host mirror
 * (tim_amd/functional.py) from the generic entry points above plus the two below. */

/* Sequence assembly (encodings.py:190-250): token row s of every window is described by
 * rows[s]: x[b,s,:d] = (kind 0: e0[b,src] | kind 2: e1[b,src] | kind 1: cls[src]),
 * x[b,s,d:] = te[b,te_row], + mod[mod] over all 2d columns (mod < 0: none), then sequence
 * dropout; writes the fp32 residual stream and its operand-dtype copy. */
typedef struct TimSeqRow {
  int32_t kind;
  int32_t src;
  int32_t te_row;
  int32_t mod;
} TimSeqRow;
int timhip_assemble_fwd(int precision, const TimSeqRow* rows /*device, [S]*/, int B, int S, int d,
                        const float* e0, const float* e1, int n_e_rows, const float* cls,
                        const float* te, int T, const float* mod, float p_seq_drop, uint64_t seed,
                        uint32_t site, float* x, void* x_T, void* stream);
/* d_e0/d_e1 are written; d_cls, d_te, d_mod are accumulated (+=, must be zeroed by the caller) */
int timhip_assemble_bwd(const TimSeqRow* rows, int B, int S, int d, const float* dx, int n_e_rows,
                        int T, float p_seq_drop, uint64_t seed, uint32_t site, float* d_e0,
                        float* d_e1, float* d_cls, float* d_te, float* d_mod, void* stream);

/* The same with the CLS token vectors (ncls <= 8, [d] each) and the modality vectors (nmod <= 4, [2 d] each) given by pointer:
 * host arrays of device pointers, 16-byte aligned - the separate nn.Parameters of encodings.py:29-35,158-175 as they lie (and, in
 * the backward, their gradient views: accumulated into, zeroed by the caller).  Saves the two concatenations of a forward and
 * the copy back of a backward. */
int timhip_assemble_fwd_p(int precision, const TimSeqRow* rows, int B, int S, int d, const float* e0, const float* e1,
                          int n_e_rows, const float* const* cls, int ncls, const float* te, int T, const float* const* mod,
                          int nmod, float p_seq_drop, uint64_t seed, uint32_t site, float* x, void* x_T, void* stream);
int timhip_assemble_bwd_p(const TimSeqRow* rows, int B, int S, int d, const float* dx, int n_e_rows, int T,
                          float p_seq_drop, uint64_t seed, uint32_t site, float* d_e0, float* d_e1, float* const* d_cls, int ncls,
                          float* d_te, float* const* d_mod, int nmod, void* stream);

/* ---- heads ------------------------------------------------------------------ */
/* rows_T[B*n, E] (T) = x_T[b, s0 + i, :] for i < n : gathers the query rows a head reads */
int timhip_gather_rows(int precision, const void* x_T, int B, int S, int E, int s0, int n,
                       void* rows_T, void* stream);
/* dx[b, s0+i, :] += d_rows[b*n+i, :]  (fp32) */
int timhip_scatter_rows_add(const float* d_rows, int B, int S, int E, int s0, int n, float* dx,
                            void* stream);
/* The same moves for up to 6 token ranges in one launch (the four classification heads; the ranges of a scatter must be
 * DISJOINT - heads that share token rows, as in the detection model, go through timhip_scatter_rows_add one by one), and the
 * fp32 -> operand-dtype
 * cast of several cotangent matrices (dst[i] is [rows[i], ld[i]], zero padded beyond cols[i]) in one launch. */
int timhip_gather_ranges(int precision, const void* x_T, int B, int S, int E, int count, const int* s0, const int* n,
                         void* const* rows_T, void* stream);
/* rows3_T[i][b * n[i] + j, 0 .. 3 E) = the fp32 row x[b, s0[i] + j, :] as the split operand [hi | lo | hi] (three 16-bit column
 * blocks of width E, a multiple of 64; 16-bit modes): timhip_gather_ranges on the fp32 rows + timhip_split3_many (mode 0) in one
 * launch - how the fp16 mode feeds its classification heads (count 1 .. 6 ranges) */
int timhip_gather_split3_ranges(int precision, const float* x, int B, int S, int E, int count, const int* s0, const int* n,
                                void* const* rows3_T, void* stream);
/* dx[B,S,E] (fp32, the gradient entering the last encoder layer) written in one pass: rows s < F <- feats_cot[b,s,:] (NULL: 0),
 * rows of the count <= 6 DISJOINT token ranges [s0[i], s0[i] + n[i]) (s0[i] >= F) <- d_rows[i][b*n[i] + j,:], every other row 0. */
int timhip_dx_init(int B, int S, int F, int E, const float* feats_cot, int count, const int* s0, const int* n,
                   const float* const* d_rows, float* dx, void* stream);
/* dst[rows, ld] (operand dtype, zero padded beyond cols) = scale * grad_out * y * (1 - y): the backward of a sigmoid output layer
 * (the regression heads of det head.py:95-163) written as the operand rows of the gradient GEMMs; scale: device scalar or NULL */
int timhip_sigmoid_bwd_rows(int precision, const float* grad_out, const float* y, int rows, int cols, void* dst, int ld,
                            const float* scale, void* stream);
/* ... with range i given as nslab[i] (1 .. 16; NULL: 1 each) consecutive [B n[i], E] slabs at d_rows[i] that are added up on the
 * way: the input-gradient product of a head with a long contraction (3806 action classes) runs as several column chunks of the
 * contraction side by side, each into its own slab (tim_amd/functional.py: the 240 blocks of that product ran 60 contraction
 * steps each while the other heads' blocks had finished after 2 - 5). */
int timhip_dx_init_slabs(int B, int S, int F, int E, const float* feats_cot, int count, const int* s0, const int* n,
                         const float* const* d_rows, const int* nslab, float* dx, void* stream);
int timhip_scatter_ranges_add(int B, int S, int E, int count, const int* s0, const int* n,
                              const float* const* d_rows, float* dx, void* stream);
int timhip_cast_rows_many(int precision, int count, const float* const* src, const int* rows, const int* cols,
                          void* const* dst, const int* ld, const float* scale, void* stream);

/* ---------------------------------------------------------------- detection query labelling */
/* TIM.label_queries of the detection model (detection/time_interval_machine/models/tim.py:214-270, with get_query_ious
 * :186-212): every query [b, q] is matched to the ground-truth segment of window b with the largest 1-D IoU (first maximum,
 * as torch.argmax; all intervals of a window are first shifted by |min(0, earliest segment start)|).  queries [B,Nq,2] and
 * segs [B,Ng,2] fp32, labels [B,Ng,NL] int64.  Writes targets [B*Nq,2] (the matched - shifted - segment, +inf for queries whose
 * best IoU is < iou_threshold), ious [B*Nq] and qlabels [B*Nq,NL] (-1 for those negatives).  fp32 results are the
 * reference's bit for bit. */
int timhip_label_queries(const float* queries, const float* segs, const int64_t* labels, int B, int Nq, int Ng, int NL,
                         float iou_threshold, float* targets, float* ious, int64_t* qlabels, void* stream);
/* The label-smoothed one-hot classification targets of TIM.assign_positive_labels (tim.py:157-184) for label column `col`
 * of qlabels [rows, ld]: out[r, c] = (c == label) ? on : base for c < n (a label of -1 leaves the row at `base`).  The caller
 * passes on = fp32(fp32(smoothing) + base), base = fp32((1 - smoothing) / (n + 1)), the two values the reference's expression
 * takes.  out [rows, n] fp32, 16-byte aligned. */
int timhip_smooth_one_hot(const int64_t* qlabels, int ld, int col, int64_t rows, int n, float on, float base, float* out,
                          void* stream);

/* ---------------------------------------------------------------- loss tail of the training step */
/* Label-smoothed cross entropy under mixup, as recognition/scripts/train.py:46-49,218-316 applies it through
 * utils/mixup.py:24-39:  loss = lam * mean_{r: ta[r] != -1} CE(logits[r], ta[r]) + (1-lam) * mean_{r: tb[r] != -1}
 * CE(logits[r], tb[r]),  CE with label smoothing `smoothing` (torch.nn.CrossEntropyLoss(label_smoothing, ignore_index=-1)).
 * target_b may be NULL (plain criterion, lam = 1).  stats [rows,4] and accum [4] are scratch kept for the backward;
 * loss is a device scalar.  dlogits = grad_out[0] * d loss / d logits (grad_out: device scalar or NULL = 1). */
int timhip_ce_mixup_fwd(const float* logits, int rows, int C, int ld, const int64_t* target_a, const int64_t* target_b,
                        float lam, float smoothing, float* stats, float* accum, float* loss, void* stream);
int timhip_ce_mixup_bwd(const float* logits, int rows, int C, int ld, const int64_t* target_a, const int64_t* target_b,
                        float lam, float smoothing, const float* stats, const float* accum, const float* grad_out,
                        float* dlogits, int ldd, void* stream);

/* DRLoc sample collection (models/helpers/losses/drloc.py:11-15,24-26,37-39): out[(b*m+i), 0:D] = x1[b, pos1[b,i], :],
 * out[.., D:2D] = x2[b, pos2[b,i], :], cast to the operand dtype ([n*m, ld] GEMM operand of drloc_mlp.0).
 * x1, x2: fp32 [n, l, D] views with element strides (batch_stride, row_stride, 1).  The scatter adds the gradient
 * of that operand back: dx1[b, pos1[b,i], :] += d_pts[(b*m+i), 0:D] (fp32 atomics; positions repeat). */
int timhip_drloc_gather(int precision, const float* x1, const float* x2, int64_t batch_stride, int64_t row_stride,
                        int n, int l, int D, const int64_t* pos1, const int64_t* pos2, int m, void* out, int ld,
                        void* stream);
int timhip_drloc_scatter_add(const float* d_pts, int ldg, float* dx1, float* dx2, int64_t batch_stride,
                             int64_t row_stride, int n, int l, int D, const int64_t* pos1, const int64_t* pos2, int m,
                             void* stream);

/* ---------------------------------------------------------------- detection losses */
/* sigmoid focal loss (detection models/helpers/losses/sigmoid.py:5-52) under get_loss(.., weights, reduction="sum")
 * (losses/loss.py:5-14):  loss_sum = sum_{r: valid[r]} w[r] * sum_c focal(logits[r,c], targets[r,c]);  targets are the
 * smoothed one-hot floats of det tim.py:157-184; row_weights / row_valid may be NULL; loss_elem (optional, [rows,C])
 * receives the weighted per-element terms (reduction "none").  bwd: dlogits = grad_out[0] * d loss_sum / d logits. */
int timhip_focal_loss_fwd(const float* logits, const float* targets, int rows, int C, const float* row_weights,
                          const uint8_t* row_valid, float alpha, float gamma, float* loss_sum, float* loss_elem,
                          void* stream);
int timhip_focal_loss_bwd(const float* logits, const float* targets, int rows, int C, const float* row_weights,
                          const uint8_t* row_valid, float alpha, float gamma, const float* grad_out, float* dlogits,
                          void* stream);
/* 1-D centre-offset DIoU loss (losses/iou.py:4-65), summed over the valid rows; offsets are [n,2] = (left, right).
 * loss_sum and/or dpred (= grad_out[0] * d loss / d pred) are produced when non-NULL. */
int timhip_diou_1d(const float* pred_offsets, const float* target_offsets, int n, const uint8_t* row_valid, float eps,
                   const float* grad_out, float* loss_sum, float* dpred, void* stream);
/* One modality side of the detection training loss (det scripts/train.py:222-349) with the row flags derived where they are used:
 * valid_cls = iou >= 0, row weight = iou < iou_threshold ? 1 : iou, positive = offsets[r, 0] != inf; the nheads <= 4 classification
 * heads' focal sums, the DIoU sum of the positive rows and their count in one pass each, then
 *   normaliser <- momentum * normaliser + (1 - momentum) * max(positives, 1)      (device scalar, in / out)
 *   loss = focal / (nheads * normaliser) + (positives > 0 ? lambda_reg * DIoU / normaliser : 0)
 * block (device float[8], out) = {loss, focal sum, DIoU sum, positives, normaliser used, 0, 0, 0}; the backward reads it.
 * logits / targets / dlogits: host arrays of device pointers, [rows, C[k]] each; dlogits[k] may be NULL, dreg may be NULL. */
int timhip_det_side_loss_fwd(const float* const* logits, const float* const* targets, const int* C, int nheads, int rows,
                             const float* iou, const float* offsets, const float* reg_pred, float iou_threshold, float alpha,
                             float gamma, float eps, float lambda_reg, float momentum, float* normaliser, float* block,
                             void* stream);
int timhip_det_side_loss_bwd(const float* const* logits, const float* const* targets, const int* C, int nheads, int rows,
                             const float* iou, const float* offsets, const float* reg_pred, float iou_threshold, float alpha,
                             float gamma, float eps, float lambda_reg, const float* block, const float* grad_out,
                             float* const* dlogits, float* dreg, void* stream);


/* ---------------------------------------------------------------- 1-D segment NMS */
/* Batched soft-NMS over independent groups (one per video x class), bit-identical in its selection to
 * detection/eval_detection/csrc/nms_cpu.cpp:69-170 (softnms_1d_cpu) run on each group:
 *   segs [N,2], scores [N] grouped contiguously; group g = rows [group_offsets[g], group_offsets[g+1]) (int32, device copy
 *   and host copy of the same G+1 values); method 0 vanilla / 1 linear / 2 gaussian.
 *   dets [N,3]: for group g rows group_offsets[g] .. +count[g]-1 hold (start, end, score at selection) in selection order,
 *   inds [N]: the selected segments' indices relative to the group's first row; count [G].
 * workspace: timhip_softnms_1d_workspace_bytes(N, G) bytes of device memory.  Synchronises `stream` before returning
 * (it stages per-size-class group lists from pageable host memory). */
size_t timhip_softnms_1d_workspace_bytes(int64_t n_total, int n_groups);
int timhip_softnms_1d(const float* segs, const float* scores, const int32_t* group_offsets,
                      const int32_t* group_offsets_host, int n_groups, float iou_threshold, float sigma, float min_score,
                      int method, float* dets, int32_t* inds, int32_t* count, void* workspace, size_t workspace_bytes,
                      void* stream);
/* Batched vanilla NMS (nms_cpu.cpp:19-60): order [N] = for each group the positions (relative to the group) sorted by
 * descending score; keep [N] receives per group the kept positions in that order, count [G] their number;
 * removed_scratch: N bytes. */
int timhip_nms_1d(const float* segs, const int32_t* order, const int32_t* group_offsets, int n_groups, float iou_threshold,
                  uint8_t* removed_scratch, int32_t* keep, int32_t* count, void* stream);

/* ---------------------------------------------------------------- sliding-window batch assembly */
/* recognition datasets/sliding_window.py:341-421 (__getitem__) for a batch, on feature stores resident in HBM.
 * feats [sum_v N_feat(v) * num_aug, C] fp32: every video's [N_feat, num_aug, C] array, concatenated;
 * video_row0[w] = first feature row (before the num_aug factor) of window w's video; feat_indices [W, num_feats] (int32);
 * windows [B] = the window ids of the batch; aug_indices [B*num_feats] in [0, num_aug) or NULL (0).
 * out [B, num_feats, C] = feats[(video_row0[w] + feat_indices[w, j]) * num_aug + aug[b, j]]            (:352-358, :364-370) */
int timhip_window_gather(const float* feats, int C, int num_aug, const int64_t* video_row0, const int32_t* feat_indices,
                         int num_feats, const int32_t* windows, int B, const int32_t* aug_indices, float* out,
                         void* stream);
/* times [B, T, 2], T = (v ? nf : 0) + (a ? nf : 0) + max_v + max_a, rows ordered vis feats | aud feats | v queries | a queries:
 * clamp((t - start_sec[w]) / window_size, min=0)   (:359-360, :371-372, :402-404).  *_feat_times [rows, ld >= 2] per-video
 * feature (start, end) tables concatenated like feats (without the num_aug factor); *_queries [W, max, 2] zero padded. */
int timhip_window_times(const float* v_feat_times, int v_ld, const int64_t* v_row0, const float* a_feat_times, int a_ld,
                        const int64_t* a_row0, const int32_t* feat_indices, int num_feats, const int32_t* windows, int B,
                        const float* v_queries, int max_v, const float* a_queries, int max_a, const float* start_sec,
                        float window_size, float* times, void* stream);

/* ---------------------------------------------------------------- measurement hook (bench.py roofline) */
/* While armed, every NT / TN GEMM launch of at least min_flops algorithmic FLOPs (2*M*N*K) is bracketed by two HIP events
 * recorded on the stream it is launched on (up to `capacity` launches).  stop() waits for them and returns the summed
 * durations [ms], the summed FLOPs and the number of launches.  Not thread-safe against concurrent start/stop. */
int timhip_gemm_timing_start(int capacity, double min_flops);
int timhip_gemm_timing_stop(double* total_ms, double* total_flops, int* launches);
/* ABI 6: the same stop, per kernel family - [0] the GEMM launches (work = FLOPs), [1] the attention launches, [2] the LayerNorm
 * launches (work = their algorithmic bytes: operand rows read + rows written); each array has three entries.  While armed the
 * attention / LayerNorm launches are bracketed like the GEMMs (no min_flops filter). */
int timhip_timing_stop_families(double* ms3, double* work3, int* launches3);

#ifdef __cplusplus
}
#endif
#endif /* TIMHIP_H */
