// extern "C" stage entry points: one post-norm encoder layer forward/backward
// (transformers.py:92-111), plus small utilities.  Pure launch sequencing: no
// allocation, no synchronisation, no retained state.
#include <atomic>
#include "common.h"

namespace {

struct SavedLayout {
  size_t qkv, o, lse, y1, st1, x1t, u, h, y2, st2, ffn_mask, attn_keep, total;
};

SavedLayout saved_layout(const TimDesc& d) {
  const size_t M = (size_t)d.B * d.S, ts = opsize(d.precision);
  SavedLayout L;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  L.qkv = take(M * 3 * d.E * ts);
  L.o = take(M * d.E * ts);
  L.lse = take((size_t)d.B * d.H * d.S * 4);
  L.y1 = take(M * d.E * 4);
  L.st1 = take(M * 2 * 4);
  L.x1t = take(M * d.E * ts);
  L.u = take(M * d.FF * ts);
  L.h = take(M * d.FF * ts);
  L.y2 = take(M * d.E * 4);
  L.st2 = take(M * 2 * 4);
  L.ffn_mask = take(M * d.FF / 8);   // keep-bits of the FFN dropout (FF % 64 == 0): written by norm1, read by the linear1 epilogue
  L.attn_keep = take((size_t)d.B * d.H * d.S * 16);   // keep-bits of the attention dropout (timhip_attn_keep_bits, TIMHIP_DESC_ATTN_KEEP_BITS)
  L.total = off;
  return L;
}

int splitk_for(int Mout, int Nout, int Kp) {
  const int tiles = ((Mout + 127) / 128) * ((Nout + 127) / 128);
  int sk = (512 + tiles - 1) / tiles;
  const int maxk = Kp / 256 > 0 ? Kp / 256 : 1;
  if (sk > maxk) sk = maxk;
  if (sk > 8) sk = 8;
  if (sk < 1) sk = 1;
  return sk;
}

struct WgradWs { size_t tA, tB, slab, total; };
WgradWs wgrad_ws(int prec, int Nout, int Kout, int M) {
  const size_t Mp = round_up(M, 64), ts = opsize(prec);
  WgradWs w;
  if (h16_storage(prec)) {  // transposing-read kernel: no operand copies
    w.tA = w.tB = w.slab = 0;
    w.total = tim_wgrad_tn_ws(Nout, Kout, M);
    return w;
  }
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  w.tA = take((size_t)Nout * Mp * ts);
  w.tB = take((size_t)Kout * Mp * ts);
  const int sk = splitk_for(Nout, Kout, (int)Mp);
  w.slab = take(sk > 1 ? (size_t)sk * Nout * Kout * 4 : 0);
  w.total = off;
  return w;
}

struct WsLayout {
  size_t f32a, f32b, Ta, Tb, Tc, tA, tB, attn, lnp, total, wg_bytes;
};

WsLayout ws_layout(const TimDesc& d) {
  const size_t M = (size_t)d.B * d.S, ts = opsize(d.precision);
  const size_t Mp = round_up((int)M, 64);
  const size_t wide = (size_t)(3 * d.E > d.FF ? 3 * d.E : d.FF);
  const size_t mid = (size_t)(d.FF > d.E ? d.FF : d.E);
  WsLayout L;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  L.f32a = take(M * d.E * 4);
  L.f32b = take(M * d.E * 4);
  L.Ta = take(M * wide * ts);
  L.Tb = take(M * d.E * ts);
  L.Tc = take(M * d.E * ts);
  {
    size_t wg = wgrad_ws(d.precision, 3 * d.E, d.E, (int)M).total;
    size_t w2 = wgrad_ws(d.precision, d.E, d.FF, (int)M).total;
    size_t w3 = wgrad_ws(d.precision, d.FF, d.E, (int)M).total;
    if (w2 > wg) wg = w2;
    if (w3 > wg) wg = w3;
    if (h16_storage(d.precision)) {   // the grouped launch of timhip_layer_bwd_weights (slabs only when it splits)
      const TimWgradItem it[4] = {{nullptr, nullptr, nullptr, nullptr, d.E, d.FF, d.E, d.FF},
                                  {nullptr, nullptr, nullptr, nullptr, d.FF, d.E, d.FF, d.E},
                                  {nullptr, nullptr, nullptr, nullptr, d.E, d.E, d.E, d.E},
                                  {nullptr, nullptr, nullptr, nullptr, 3 * d.E, d.E, 3 * d.E, d.E}};
      const size_t wgp = tim_wgrad_group_ws(it, 4, (int)M);
      if (wgp > wg) wg = wgp;
    }
    L.tA = take(wg);
    L.tB = L.tA;
    L.wg_bytes = wg;
  }
  (void)Mp; (void)wide; (void)mid;
  L.attn = take(tim_attention_bwd_ws(d));
  L.lnp = take(tim_layernorm_bwd_ws((int)M, d.E));
  L.total = off;
  return L;
}

int check_layer_desc(const TimDesc& d) {
  if (d.B <= 0 || d.S <= 0 || d.F <= 0 || d.F > d.S || d.E <= 0 || d.H <= 0 || d.FF <= 0) return TIMHIP_EINVAL;
  if (d.E % 64 || d.FF % 64 || d.E % d.H) return TIMHIP_EUNSUPPORTED;
  if (!valid_precision(d.precision)) return TIMHIP_EUNSUPPORTED;
  if (d.p_drop < 0.f || d.p_drop >= 1.f) return TIMHIP_EINVAL;
  return TIMHIP_OK;
}

TimEpi epi0() {
  TimEpi e;
  e.out0 = e.out1 = nullptr; e.bias = e.res = nullptr; e.aux = nullptr;
  e.ld0 = e.ld1 = e.ldres = e.ldaux = 0; e.p_drop = 0.f; e.site = 0; e.seed = 0;
  e.mask = nullptr; e.ldmask = 0; e.reserved = 0; e.a_wrap_k = 0; e.reserved2 = 0;
  e.ln_stats = e.ln_w = e.ln_b = nullptr;
  e.acc_scale = nullptr;
  return e;
}

// dW[Nout, Kout] += dY[M, Nout]^T X[M, Kout]   (+ db[Nout] += colsum dY)
// Both operands are transposed into K(=M)-contiguous copies, the product runs split-K into fp32
// slabs (plain coalesced stores, no atomics) and one reduce kernel adds the slabs into dW.
int wgrad(int prec, const void* dY, int ldy, int Nout, const void* X, int ldx, int Kout, int M, float* dW, float* db,
          void* ws, size_t ws_bytes, hipStream_t s, int accumulate = 1, const float* out_scale = nullptr) {
  const int Mp = round_up(M, 64);
  const WgradWs W = wgrad_ws(prec, Nout, Kout, M);
  if (ws_bytes < W.total) return TIMHIP_EWORKSPACE;
  if (h16_storage(prec)) return tim_wgrad_tn_h16(prec, dY, ldy, Nout, X, ldx, Kout, M, dW, db, ws, ws_bytes, s, accumulate, out_scale);
  if (!accumulate || out_scale) return TIMHIP_EUNSUPPORTED;   // the fp32 / bf16x3 route accumulates into dW, unscaled
  char* w = (char*)ws;
  void* tA = w + W.tA; void* tB = w + W.tB; float* slab = (float*)(w + W.slab);
  int rc;
  if ((rc = tim_transpose(prec, dY, M, Nout, ldy, tA, Mp, db, s))) return rc;
  if ((rc = tim_transpose(prec, X, M, Kout, ldx, tB, Mp, nullptr, s))) return rc;
  const int sk = splitk_for(Nout, Kout, Mp);
  TimEpi e = epi0();
  if (sk == 1 || (Kout % 4) != 0) {
    e.out0 = dW; e.ld0 = Kout;
    return tim_gemm_nt(prec, TIMHIP_EPI_ATOMIC_F32, tA, Mp, tB, Mp, Nout, Kout, Mp, e, sk, s);
  }
  e.out0 = slab; e.ld0 = Kout;
  if ((rc = tim_gemm_nt(prec, TIMHIP_EPI_STORE_F32, tA, Mp, tB, Mp, Nout, Kout, Mp, e, sk, s))) return rc;
  return tim_slab_reduce(slab, (long long)Nout * Kout, sk, dW, s);
}

}  // namespace

const unsigned long long* tim_salt_ptr = nullptr;

// ---- environment knobs, cached (common.h: TimKnobs) ----------------------------------------------------------------
namespace {
TimKnobs g_knobs;
std::atomic<int> g_knobs_state{0};   // 0: not read, 1: valid
int env_int(const char* name, int dflt) { const char* v = getenv(name); return (v && v[0]) ? atoi(v) : dflt; }
void read_knobs(TimKnobs& k) {
  k.gemm_pp = env_int("TIMHIP_GEMM_PP", 1); k.gemm_ld = env_int("TIMHIP_GEMM_LD", 1); k.gemm_pf = env_int("TIMHIP_GEMM_PF", 4);
  k.gemm_pf_mode = env_int("TIMHIP_GEMM_PF_MODE", 1); k.gemm_pf_mr = env_int("TIMHIP_GEMM_PF_MR", 0);
  k.gemm_ldp = env_int("TIMHIP_GEMM_LDP", 1); k.gemm_ld1 = env_int("TIMHIP_GEMM_LD1", 0); k.gemm_pt = env_int("TIMHIP_GEMM_PT", 0);
  k.gemm_dg = env_int("TIMHIP_GEMM_DG", 0); k.gemm_dg_offset = env_int("TIMHIP_GEMM_DG_OFFSET", 9);
  k.fuse_ln = env_int("TIMHIP_FUSE_LN", 0); k.fuse_ln_spin = env_int("TIMHIP_FUSE_LN_SPIN", 100000);
  k.wgrad_pp = env_int("TIMHIP_WGRAD_PP", 1); k.wgrad_ld = env_int("TIMHIP_WGRAD_LD", 1); k.wgrad_pf = env_int("TIMHIP_WGRAD_PF", 4);
  k.wgrad_p8 = env_int("TIMHIP_WGRAD_P8", 1); k.wgrad_p8_ph = env_int("TIMHIP_WGRAD_P8_PH", 2);
  k.attn_waves = env_int("TIMHIP_ATTN_WAVES", 0); k.attn_fused = env_int("TIMHIP_ATTN_FUSED", 1);
  k.ln_rpb = env_int("TIMHIP_LN_RPB", 0);
  k.gemm_tmw = env_int("TIMHIP_GEMM_TMW", 0);
  k.ln_rpb_small = env_int("TIMHIP_LN_RPB_SMALL", 0); k.ln_fwd_rpb = env_int("TIMHIP_LN_FWD_RPB", 0);
  k.gemm_small_nst = env_int("TIMHIP_GEMM_SMALL_NST", 0); k.gemm_small_w8 = env_int("TIMHIP_GEMM_SMALL_W8", 1); k.gemm_p8_ph = env_int("TIMHIP_GEMM_P8_PH", 2);
  k.gemm_pp_min = env_int("TIMHIP_GEMM_PP_MIN_TILES", 192);
  k.attn_ks = env_int("TIMHIP_ATTN_KS", 1);
  k.gemm_p8 = env_int("TIMHIP_GEMM_P8", 1);
  k.ln_pair = env_int("TIMHIP_LN_PAIR", 1);
  k.epi_pair = env_int("TIMHIP_EPI_PAIR", 1);
  k.attn_split_min = env_int("TIMHIP_ATTN_SPLIT_MIN", 4);
  if (k.attn_split_min < 1) k.attn_split_min = 1;
}
}  // namespace
const TimKnobs& tim_knobs() {
  if (g_knobs_state.load(std::memory_order_acquire) == 0) {   // (a benign race reads the same environment twice)
    TimKnobs k;
    read_knobs(k);
    g_knobs = k;
    g_knobs_state.store(1, std::memory_order_release);
  }
  return g_knobs;
}

extern "C" {

int timhip_version(void) { return TIMHIP_VERSION; }
void timhip_reload_env(void) { g_knobs_state.store(0, std::memory_order_release); (void)tim_knobs(); }
int timhip_gemm_p8_choice(int epi, int M, int N, int K) { return tim_gemm_p8_choice(epi, M, N, K); }
int timhip_build_flags(void) {
#ifdef TIMHIP_TUNING
  return 1;
#else
  return 0;
#endif
}

int timhip_dropout_salt(const unsigned long long* dev_salt) {
  tim_salt_ptr = dev_salt;
  return TIMHIP_OK;
}

const char* timhip_strerror(int code) {
  switch (code) {
    case TIMHIP_OK: return "ok";
    case TIMHIP_EINVAL: return "invalid argument (null pointer or bad descriptor field)";
    case TIMHIP_EUNSUPPORTED: return "shape or precision not supported by the gfx950 kernels";
    case TIMHIP_EWORKSPACE: return "workspace too small";
    case TIMHIP_ELAUNCH: return "HIP kernel launch failed";
    case TIMHIP_EALIGN: return "pointer or leading dimension not aligned";
    default: return "unknown timhip error";
  }
}

int timhip_attn_keep_bits(const TimDesc* d, int nlayers, void* const* saved, void* stream) {
  if (!d || !saved || nlayers <= 0) return TIMHIP_EINVAL;
  // (the geometry whose kernels read them: 128-wide heads, 97 .. 128 feature keys - attn_fwd_mfma<.., 128, 4, true> and the fused
  //  key-split backward; every other shape draws its masks in the kernels)
  if (!h16_storage(d->precision) || !(d->p_drop > 0.f) || d->B <= 0 || d->S <= 0 || d->H <= 0 || d->E / d->H != 128 ||
      (d->F + 31) / 32 != 4)
    return TIMHIP_EUNSUPPORTED;
  const SavedLayout L = saved_layout(*d);
  for (int l0 = 0; l0 < nlayers; l0 += 8) {
    unsigned long long* out[8] = {};
    const int n = nlayers - l0 < 8 ? nlayers - l0 : 8;
    for (int i = 0; i < n; ++i) {
      if (!saved[l0 + i]) return TIMHIP_EINVAL;
      out[i] = reinterpret_cast<unsigned long long*>((char*)saved[l0 + i] + L.attn_keep);
    }
    const int rc = tim_attn_keep_bits(*d, l0, n, out, (hipStream_t)stream);
    if (rc) return rc;
  }
  return TIMHIP_OK;
}

size_t timhip_layer_saved_bytes(const TimDesc* d) { return d ? saved_layout(*d).total : 0; }

// test hook: where a field of the (otherwise opaque) saved block lives
int timhip_layer_saved_field(const TimDesc* d, int field, size_t* offset, size_t* bytes) {
  if (!d || !offset || !bytes) return TIMHIP_EINVAL;
  const SavedLayout L = saved_layout(*d);
  const size_t M = (size_t)d->B * d->S, ts = opsize(d->precision);
  switch (field) {
    case TIMHIP_SAVED_QKV: *offset = L.qkv; *bytes = M * 3 * d->E * ts; break;
    case TIMHIP_SAVED_O: *offset = L.o; *bytes = M * d->E * ts; break;
    case TIMHIP_SAVED_Y1: *offset = L.y1; *bytes = M * d->E * 4; break;
    case TIMHIP_SAVED_X1T: *offset = L.x1t; *bytes = M * d->E * ts; break;
    case TIMHIP_SAVED_H: *offset = L.h; *bytes = M * d->FF * ts; break;
    case TIMHIP_SAVED_Y2: *offset = L.y2; *bytes = M * d->E * 4; break;
    case TIMHIP_SAVED_FFN_KEEP_BITS: *offset = L.ffn_mask; *bytes = M * d->FF / 8; break;
    case TIMHIP_SAVED_ATTN_KEEP_BITS: *offset = L.attn_keep; *bytes = (size_t)d->B * d->H * d->S * 16; break;
    default: return TIMHIP_EINVAL;
  }
  return TIMHIP_OK;
}
size_t timhip_layer_workspace_bytes(const TimDesc* d);

int timhip_gemm_nt(int precision, int epi, const void* A, int lda, const void* B, int ldb, int M, int N, int K,
                   const TimEpi* e, int splitk, void* stream) {
  if (!e) return TIMHIP_EINVAL;
  return tim_gemm_nt(precision, epi, A, lda, B, ldb, M, N, K, *e, splitk, (hipStream_t)stream);
}

size_t timhip_wgrad_workspace_bytes(int precision, int Nout, int Kout, int M) {
  return wgrad_ws(precision, Nout, Kout, M).total;
}

int timhip_gemm_nt_group(int precision, int epi, const TimGemmItem* items, int n, void* stream) {
  return tim_gemm_nt_group(precision, epi, items, n, (hipStream_t)stream);
}

size_t timhip_wgrad_group_workspace_bytes(int precision, const TimWgradItem* items, int n, int M) {
  return (h16_storage(precision) && items && n > 0) ? tim_wgrad_group_ws(items, n, M) : 0;
}

int timhip_wgrad_group(int precision, const TimWgradItem* items, int n, int M, int accumulate, void* workspace,
                       size_t workspace_bytes, const float* out_scale, void* stream) {
  return tim_wgrad_group_h16(precision, items, n, M, accumulate, workspace, workspace_bytes, out_scale, (hipStream_t)stream);
}

int timhip_wgrad(int precision, const void* dY, int ldy, int Nout, const void* X, int ldx, int Kout, int M,
                 float* dW, float* db, void* workspace, size_t workspace_bytes, const float* out_scale, void* stream) {
  if (!dY || !X || !dW || !workspace) return TIMHIP_EINVAL;
  return wgrad(precision, dY, ldy, Nout, X, ldx, Kout, M, dW, db, workspace, workspace_bytes, (hipStream_t)stream, 1, out_scale);
}

// The fp32 residual stream is only ever READ by the "+ residual" epilogues, and each of them can normalise on the fly
// (TimEpi.ln_*): a layer's input rows are LayerNorm-2 of the previous layer's y2, its inner residual LayerNorm-1 of its own
// y1.  So the normalised fp32 rows are written only where someone else needs them (x_out of the last layer -> feats);
// LayerNorm writes its bf16 operand copy and the statistics, 60 instead of 100 MB per launch.
static int layer_fwd_impl(const TimDesc& d, const TimLayerParams* w, const float* x_in, const float* x_in_prenorm,
                          const float* x_in_stats, const float* x_in_lnw, const float* x_in_lnb, const void* x_in_T,
                          float* x_out, void* x_out_T, void* saved, hipStream_t s) {
  int rc;
  const int M = d.B * d.S, E = d.E, FF = d.FF, prec = d.precision;
  const SavedLayout L = saved_layout(d);
  char* sv = (char*)saved;
  void* qkv = sv + L.qkv; void* o = sv + L.o; float* lse = (float*)(sv + L.lse);
  float* y1 = (float*)(sv + L.y1); float* st1 = (float*)(sv + L.st1); void* x1t = sv + L.x1t;
  void* u = sv + L.u; void* h = sv + L.h; float* y2 = (float*)(sv + L.y2); float* st2 = (float*)(sv + L.st2);

  // 1. packed in-projection (F._in_projection_packed)
  // (a *_SPLIT flag: that weight pointer is a split copy [hi | lo | ..] with row stride 3 K; the product runs over 2 K with the
  //  activation operand read twice - TimEpi.a_wrap_k)
  auto split = [&](int flag) { return (d.reserved & flag) != 0 && h16_storage(prec); };
  TimEpi e = epi0();
  e.out0 = qkv; e.ld0 = 3 * E; e.bias = w->in_b;
  if (split(TIMHIP_DESC_INPROJ_SPLIT)) {
    e.a_wrap_k = E; e.reserved = 2;
    if ((rc = tim_gemm_nt(prec, TIMHIP_EPI_STORE_T, x_in_T, E, w->in_w, 3 * E, M, 3 * E, 2 * E, e, 1, s))) return rc;
  } else if ((rc = tim_gemm_nt(prec, TIMHIP_EPI_STORE_T, x_in_T, E, w->in_w, E, M, 3 * E, E, e, 1, s))) return rc;
  // 2. structured attention
  const unsigned long long* akeep = ((d.reserved & TIMHIP_DESC_ATTN_KEEP_BITS) && d.p_drop > 0.f)
                                        ? reinterpret_cast<const unsigned long long*>(sv + L.attn_keep) : nullptr;
  if ((rc = tim_attention_fwd(d, qkv, o, lse, s, akeep))) return rc;
  // 3. out-projection + dropout1 + residual
  e = epi0();
  e.out0 = y1; e.ld0 = E; e.bias = w->out_b; e.ldres = E;
  if (x_in) {
    e.res = x_in;
  } else {   // the input rows as LayerNorm-2 of the previous layer's y2
    e.res = x_in_prenorm; e.ln_stats = x_in_stats; e.ln_w = x_in_lnw; e.ln_b = x_in_lnb;
  }
  e.p_drop = d.p_drop; e.seed = d.seed; e.site = layer_site(d.layer, SITE_L_DROP1);
  uint8_t* fmask = d.p_drop > 0.f ? (uint8_t*)(sv + L.ffn_mask) : nullptr;
  // TIMHIP_FUSE_LN=1 (round 3, opt-in): the LayerNorm that follows the out-projection / linear2 inside the GEMM's epilogue
  // (gemm_nt_ldln_kernel: the column tiles of a row panel exchange row statistics); the stand-alone LayerNorm stays behind it as
  // a launch that exits at once unless a tile's wait timed out.  Falls back to the two kernels wherever the shape does not fit.
  const bool fuse_ln = tim_knobs().fuse_ln == 1;
  const uint32_t* ln_run_if = nullptr;
  bool fused1 = false;
  if (fuse_ln && h16_storage(prec) && !split(TIMHIP_DESC_OUTPROJ_SPLIT)) {
    TimLnFuse lf{x1t, E, nullptr, 0, st1, w->n1_w, w->n1_b, fmask, FF, d.p_drop, d.seed, layer_site(d.layer, SITE_L_FFN)};
    rc = tim_gemm_nt_fuse_ln(prec, o, E, w->out_w, E, M, E, E, e, lf, &ln_run_if, s);
    if (rc == TIMHIP_OK) fused1 = true;
    else if (rc != TIMHIP_EUNSUPPORTED) return rc;
  }
  if (!fused1) {
    if (split(TIMHIP_DESC_OUTPROJ_SPLIT)) {
      // out_w = [w_hi | w_lo | ..] (row stride 3E): o [w_hi | w_lo]^T over K = 2E, o read twice - the weight to ~22 bits
      e.a_wrap_k = E; e.reserved = 2;
      if ((rc = tim_gemm_nt(prec, TIMHIP_EPI_DROP_RES_F32, o, E, w->out_w, 3 * E, M, E, 2 * E, e, 1, s))) return rc;
    } else if ((rc = tim_gemm_nt(prec, TIMHIP_EPI_DROP_RES_F32, o, E, w->out_w, E, M, E, E, e, 1, s))) return rc;
  }
  // 4. norm1.  The kernel is HBM-bound with idle VALU: it also draws the keep-bits of the FFN dropout (same Philox
  //    stream as the epilogues would use), which the linear1 epilogue and, in the backward, the gelu' epilogue read
  if ((rc = tim_layernorm_fwd(prec, y1, M, E, E, 0, w->n1_w, w->n1_b, nullptr, 0, x1t, E, st1, s, fmask, FF, d.p_drop, d.seed,
                              layer_site(d.layer, SITE_L_FFN), fused1 ? ln_run_if : nullptr))) return rc;
  // 5. linear1 + GELU(erf) + dropout
  e = epi0();
  e.out0 = h; e.ld0 = FF; e.out1 = u; e.ld1 = FF; e.bias = w->l1_b;
  e.p_drop = d.p_drop; e.seed = d.seed; e.site = layer_site(d.layer, SITE_L_FFN);
  e.mask = fmask; e.ldmask = FF / 8;
  // (out1 = `u` holds dropmask * gelu'(linear1 output): the factor the backward multiplies with - it never needs the
  //  pre-activations themselves, so its epilogue is a plain multiply)
  if (split(TIMHIP_DESC_L1_SPLIT)) {
    e.a_wrap_k = E; e.reserved = 2;
    if ((rc = tim_gemm_nt(prec, TIMHIP_EPI_GELU_DROP_G2, x1t, E, w->l1_w, 3 * E, M, FF, 2 * E, e, 1, s))) return rc;
  } else if ((rc = tim_gemm_nt(prec, TIMHIP_EPI_GELU_DROP_G2, x1t, E, w->l1_w, E, M, FF, E, e, 1, s))) return rc;
  // 6. linear2 + dropout2 + residual
  e = epi0();
  e.out0 = y2; e.ld0 = E; e.bias = w->l2_b; e.ldres = E;
  e.res = y1; e.ln_stats = st1; e.ln_w = w->n1_w; e.ln_b = w->n1_b;   // residual = norm1(y1), normalised by the epilogue
  e.p_drop = d.p_drop; e.seed = d.seed; e.site = layer_site(d.layer, SITE_L_DROP2);
  bool fused2 = false;
  if (fuse_ln && h16_storage(prec) && !split(TIMHIP_DESC_L2_SPLIT)) {
    TimLnFuse lf{x_out_T, E, x_out, E, st2, w->n2_w, w->n2_b, nullptr, 0, 0.f, 0, 0};
    rc = tim_gemm_nt_fuse_ln(prec, h, FF, w->l2_w, FF, M, E, FF, e, lf, &ln_run_if, s);
    if (rc == TIMHIP_OK) fused2 = true;
    else if (rc != TIMHIP_EUNSUPPORTED) return rc;
  }
  if (!fused2) {
    if (split(TIMHIP_DESC_L2_SPLIT)) {
      e.a_wrap_k = FF; e.reserved = 2;
      if ((rc = tim_gemm_nt(prec, TIMHIP_EPI_DROP_RES_F32, h, FF, w->l2_w, 3 * FF, M, E, 2 * FF, e, 1, s))) return rc;
    } else if ((rc = tim_gemm_nt(prec, TIMHIP_EPI_DROP_RES_F32, h, FF, w->l2_w, FF, M, E, FF, e, 1, s))) return rc;
  }
  // 7. norm2 (x_out == NULL: only the operand copy and the statistics)
  return tim_layernorm_fwd(prec, y2, M, E, E, 0, w->n2_w, w->n2_b, x_out, E, x_out_T, E, st2, s, nullptr, 0, 0.f, 0, 0,
                           fused2 ? ln_run_if : nullptr);
}

int timhip_layer_fwd(const TimDesc* dp, const TimLayerParams* w, const float* x_in, const void* x_in_T, float* x_out,
                     void* x_out_T, void* saved, void* workspace, size_t workspace_bytes, void* stream) {
  (void)workspace; (void)workspace_bytes;   // kept in the signature: earlier versions staged norm1's fp32 rows there
  if (!dp || !w || !x_in || !x_in_T || !x_out_T || !saved) return TIMHIP_EINVAL;
  int rc = check_layer_desc(*dp);
  if (rc) return rc;
  return layer_fwd_impl(*dp, w, x_in, nullptr, nullptr, nullptr, nullptr, x_in_T, x_out, x_out_T, saved, (hipStream_t)stream);
}

int timhip_layer_fwd_chained(const TimDesc* dp, const TimLayerParams* w, const TimLayerParams* prev_w, const void* prev_saved,
                             const void* x_in_T, float* x_out, void* x_out_T, void* saved, void* stream) {
  if (!dp || !w || !prev_w || !prev_saved || !x_in_T || !x_out_T || !saved) return TIMHIP_EINVAL;
  int rc = check_layer_desc(*dp);
  if (rc) return rc;
  const SavedLayout L = saved_layout(*dp);   // the previous layer has the same shape
  const char* ps = (const char*)prev_saved;
  return layer_fwd_impl(*dp, w, nullptr, (const float*)(ps + L.y2), (const float*)(ps + L.st2), prev_w->n2_w, prev_w->n2_b,
                        x_in_T, x_out, x_out_T, saved, (hipStream_t)stream);
}

// gradient operands handed from the data chain to the weight-gradient part: df[M,E] | du[M,FF] | da[M,E] | dqkv[M,3E]
struct DyLayout { size_t df, du, da, dqkv, total; };
static DyLayout dy_layout(const TimDesc& d) {
  const size_t M = (size_t)d.B * d.S, ts = opsize(d.precision);
  DyLayout L;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  L.df = take(M * d.E * ts);
  L.du = take(M * d.FF * ts);
  L.da = take(M * d.E * ts);
  L.dqkv = take(M * 3 * d.E * ts);
  L.total = off;
  return L;
}

size_t timhip_layer_ln_partial_bytes(const TimDesc* d) { return d ? 2 * tim_layernorm_bwd_ws(d->B * d->S, d->E) : 0; }
size_t timhip_layer_dy_bytes(const TimDesc* d) { return d ? dy_layout(*d).total : 0; }
size_t timhip_layer_workspace_bytes(const TimDesc* d) { return d ? ws_layout(*d).total + dy_layout(*d).total : 0; }
size_t timhip_layer_data_workspace_bytes(const TimDesc* d) { return d ? ws_layout(*d).total : 0; }
size_t timhip_layer_wgrad_workspace_bytes(const TimDesc* d) { return d ? ws_layout(*d).wg_bytes : 0; }

// The data chain.  dx_out_add / dx_in_add (both optional): the SPLIT form of the gradient stream between layers - the
// gradient of a layer boundary travels as an fp32 part (what LayerNorm-backward wrote) plus a 16-bit part (the input-gradient
// product of the GEMM in front of it, still carrying the fp16 gradient scale), and the next LayerNorm-backward adds the two
// as it reads them.  The two "+ residual" input-gradient GEMMs of a layer then store 20 MB instead of reading 40 and writing
// 40 (C2a), and the fp32 sum is never written: 360 instead of 440 MB of gradient-stream traffic per layer.
static int layer_bwd_data_impl(const TimDesc& d, const TimLayerParams* w, const void* saved, const float* dx_out,
                               const void* dx_out_add, float* dx_in, void* dx_in_add, void* dy, const TimLayerGrads* g,
                               void* workspace, size_t workspace_bytes, hipStream_t s) {
  int rc = check_layer_desc(d);
  if (rc) return rc;
  const WsLayout W = ws_layout(d);
  if (workspace_bytes < W.total) return TIMHIP_EWORKSPACE;
  const int M = d.B * d.S, E = d.E, FF = d.FF, prec = d.precision;
  const SavedLayout L = saved_layout(d);
  const DyLayout Y = dy_layout(d);
  const char* sv = (const char*)saved;
  const void* qkv = sv + L.qkv; const void* o = sv + L.o; const float* lse = (const float*)(sv + L.lse);
  const float* y1 = (const float*)(sv + L.y1); const float* st1 = (const float*)(sv + L.st1);
  const void* u = sv + L.u;
  const float* y2 = (const float*)(sv + L.y2); const float* st2 = (const float*)(sv + L.st2);
  char* ws = (char*)workspace;
  float* f32a = (float*)(ws + W.f32a); float* f32b = (float*)(ws + W.f32b);
  void* Tb = ws + W.Tb; void* Tc = ws + W.Tc;
  char* yb = (char*)dy;
  void* df = yb + Y.df; void* du = yb + Y.du; void* da = yb + Y.da; void* dqkv = yb + Y.dqkv;
  // fp16: the gradient OPERANDS (df, du, da, Tb, Tc, dqkv, the 16-bit parts of the stream and the attention scratch) carry the
  // factor S = grad_scale[0]; it enters with the T copies LayerNorm-backward writes and leaves where a product joins fp32 values
  const float* gs_in = (prec == TIMHIP_PREC_F16 && d.grad_scale) ? d.grad_scale : nullptr;
  const float* gs_out = gs_in ? gs_in + 1 : nullptr;

  // the residual part of the stream as 16-bit (TIMHIP_DESC_STREAM16*, fp16 mode only: same scale as the gradient operands)
  const bool s16 = gs_in != nullptr && (d.reserved & TIMHIP_DESC_STREAM16) != 0;
  const bool s16_in = s16 && (d.reserved & TIMHIP_DESC_STREAM16_IN) != 0;
  const bool s16_out = s16 && (d.reserved & TIMHIP_DESC_STREAM16_OUT) != 0;
  if (((d.reserved & (TIMHIP_DESC_STREAM16_IN | TIMHIP_DESC_STREAM16_OUT)) != 0 && !s16) || (s16_out && !dx_in_add)) return TIMHIP_EINVAL;
  // norm2 backward -> dy2 (fp32; STREAM16: T times S, in f32a's space) and df = dropout2-mask * dy2 (T)
  if ((rc = tim_layernorm_bwd(prec, dx_out, E, y2, E, st2, M, E, 0, w->n2_w, f32a, E, df, E, d.p_drop, d.seed,
                              layer_site(d.layer, SITE_L_DROP2), g->n2_w, g->n2_b,
                              g->ln_partials ? g->ln_partials : (float*)(ws + W.lnp), s, g->ln_partials != nullptr, gs_in,
                              dx_out_add, E, gs_out, (s16_in ? 1 : 0) | (s16 ? 2 : 0)))) return rc;
  // du = (df W2) * [dropout-mask * gelu'(pre-activation)]
  TimEpi e = epi0();
  e.out0 = du; e.ld0 = FF; e.aux = u; e.ldaux = FF;   // u = dropmask * gelu'(pre-activation), written by the forward
  if ((rc = tim_gemm_nt(prec, TIMHIP_EPI_MULAUX_T, df, E, w->l2_wt, E, M, FF, E, e, 1, s))) return rc;
  // the FFN branch's input gradient du W1 as an operand-dtype product; norm1 backward adds it to dy2 (the residual branch) as it
  // reads.  fp16 (11 bits under the gradient scale) and the fp32-storage modes (nothing is rounded) take this form; plain bf16
  // would round the branch to 8 bits per layer, so there the product joins the fp32 stream directly (dy2 += du W1, fp32)
  const bool branch_split = prec != TIMHIP_PREC_BF16;
  e = epi0();
  if (branch_split) {
    e.out0 = Tb; e.ld0 = E;
    if ((rc = tim_gemm_nt(prec, TIMHIP_EPI_STORE_T, du, FF, w->l1_wt, FF, M, E, FF, e, 1, s))) return rc;
  } else {
    e.out0 = f32b; e.ld0 = E; e.res = f32a; e.ldres = E;
    if ((rc = tim_gemm_nt(prec, TIMHIP_EPI_ADD_F32, du, FF, w->l1_wt, FF, M, E, FF, e, 1, s))) return rc;
  }
  const float* ln1_in = branch_split ? f32a : f32b;
  // norm1 backward -> dy1 (fp32) and da = dropout1-mask * dy1 (T)
  float* dy1 = dx_in_add ? dx_in : (branch_split ? f32b : f32a);
  if ((rc = tim_layernorm_bwd(prec, ln1_in, E, y1, E, st1, M, E, 0, w->n1_w, dy1, E, da, E, d.p_drop, d.seed,
                              layer_site(d.layer, SITE_L_DROP1), g->n1_w, g->n1_b,
                              g->ln_partials ? g->ln_partials + tim_layernorm_bwd_ws(M, E) / sizeof(float) : (float*)(ws + W.lnp), s,
                              g->ln_partials != nullptr, gs_in, branch_split ? Tb : nullptr, E, gs_out,
                              (s16 ? 1 : 0) | (s16_out ? 2 : 0)))) return rc;
  // do = da Wo
  e = epi0();
  e.out0 = Tc; e.ld0 = E;
  if ((rc = tim_gemm_nt(prec, TIMHIP_EPI_STORE_T, da, E, w->out_wt, E, M, E, E, e, 1, s))) return rc;
  // attention backward -> dqkv
  const unsigned long long* akeep = ((d.reserved & TIMHIP_DESC_ATTN_KEEP_BITS) && d.p_drop > 0.f)
                                        ? reinterpret_cast<const unsigned long long*>(sv + L.attn_keep) : nullptr;
  if ((rc = tim_attention_bwd(d, qkv, o, lse, Tc, dqkv, ws + W.attn, W.lnp - W.attn, s, akeep))) return rc;
  e = epi0();
  if (dx_in_add) {   // split form: the product stays 16-bit (and scaled); dy1 is already in dx_in
    e.out0 = dx_in_add; e.ld0 = E;
    return tim_gemm_nt(prec, TIMHIP_EPI_STORE_T, dqkv, 3 * E, w->in_wt, 3 * E, M, E, 3 * E, e, 1, s);
  }
  // dx_in = dqkv Win + dy1: the complete fp32 gradient (first layer of the stack, or a caller that wants one tensor)
  e.out0 = dx_in; e.ld0 = E; e.res = dy1; e.ldres = E; e.acc_scale = gs_out;
  return tim_gemm_nt(prec, TIMHIP_EPI_ADD_F32, dqkv, 3 * E, w->in_wt, 3 * E, M, E, 3 * E, e, 1, s);
}

int timhip_layer_bwd_data(const TimDesc* dp, const TimLayerParams* w, const void* saved, float* dx_out, float* dx_in,
                          void* dy, const TimLayerGrads* g, void* workspace, size_t workspace_bytes, void* stream) {
  if (!dp || !w || !saved || !dx_out || !dx_in || !dy || !g || !workspace) return TIMHIP_EINVAL;
  return layer_bwd_data_impl(*dp, w, saved, dx_out, nullptr, dx_in, nullptr, dy, g, workspace, workspace_bytes, (hipStream_t)stream);
}

int timhip_layer_bwd_data_split(const TimDesc* dp, const TimLayerParams* w, const void* saved, const float* dx_out,
                                const void* dx_out_add, float* dx_in, void* dx_in_add, void* dy, const TimLayerGrads* g,
                                void* workspace, size_t workspace_bytes, void* stream) {
  if (!dp || !w || !saved || !dx_out || !dx_in || !dy || !g || !workspace) return TIMHIP_EINVAL;
  return layer_bwd_data_impl(*dp, w, saved, dx_out, dx_out_add, dx_in, dx_in_add, dy, g, workspace, workspace_bytes,
                             (hipStream_t)stream);
}

int timhip_layer_bwd_weights(const TimDesc* dp, const void* x_in_T, const void* saved, const void* dy,
                             const TimLayerGrads* g, void* workspace, size_t workspace_bytes, void* stream) {
  if (!dp || !x_in_T || !saved || !dy || !g || !workspace) return TIMHIP_EINVAL;
  const TimDesc& d = *dp;
  int rc = check_layer_desc(d);
  if (rc) return rc;
  const int M = d.B * d.S, E = d.E, FF = d.FF, prec = d.precision;
  hipStream_t s = (hipStream_t)stream;
  const SavedLayout L = saved_layout(d);
  const DyLayout Y = dy_layout(d);
  const char* sv = (const char*)saved;
  const char* yb = (const char*)dy;
  // linear2: dW2 += df^T h ; linear1: dW1 += du^T x1 ; out-proj: dWo += da^T o ; in-proj: dWin += dqkv^T x_in
  // (TIMHIP_DESC_WGRAD_OVERWRITE: "=" instead of "+=": the gradient buffers are neither zero-filled nor read)
  const int acc = (d.reserved & TIMHIP_DESC_WGRAD_OVERWRITE) ? 0 : 1;
  const float* gs_out = (prec == TIMHIP_PREC_F16 && d.grad_scale) ? d.grad_scale + 1 : nullptr;
  if (h16_storage(prec) && !(d.reserved & TIMHIP_DESC_WGRAD_SEPARATE) && ((size_t)E * E) % 4 == 0 && ((size_t)E * FF) % 4 == 0) {
    // one grouped launch (wgrad.hip): 12 E^2 / 128^2 tiles with FF = 2E, i.e. 512 at E = 1024 - the contraction is not split
    const TimWgradItem it[4] = {
        {yb + Y.df, sv + L.h, g->l2_w, g->l2_b, E, FF, E, FF},
        {yb + Y.du, sv + L.x1t, g->l1_w, g->l1_b, FF, E, FF, E},
        {yb + Y.da, sv + L.o, g->out_w, g->out_b, E, E, E, E},
        {yb + Y.dqkv, x_in_T, g->in_w, g->in_b, 3 * E, E, 3 * E, E}};
    return tim_wgrad_group_h16(prec, it, 4, M, acc, workspace, workspace_bytes, gs_out, s);
  }
  if ((rc = wgrad(prec, yb + Y.df, E, E, sv + L.h, FF, FF, M, g->l2_w, g->l2_b, workspace, workspace_bytes, s, acc, gs_out))) return rc;
  if ((rc = wgrad(prec, yb + Y.du, FF, FF, sv + L.x1t, E, E, M, g->l1_w, g->l1_b, workspace, workspace_bytes, s, acc, gs_out))) return rc;
  if ((rc = wgrad(prec, yb + Y.da, E, E, sv + L.o, E, E, M, g->out_w, g->out_b, workspace, workspace_bytes, s, acc, gs_out))) return rc;
  return wgrad(prec, yb + Y.dqkv, 3 * E, 3 * E, x_in_T, E, E, M, g->in_w, g->in_b, workspace, workspace_bytes, s, acc, gs_out);
}

// The weight gradients of TWO layers in one grouped launch (round 6): at production batch sizes the eight products are 256 tiles of
// 256 x 256 for the eight-phase kernel (wgrad_pp.hip: wgrad_p8_kernel), one per CU; any other shape takes the grouped kernels the
// single-layer call takes (then as two rounds).  a / b: the arguments of timhip_layer_bwd_weights for the two layers (same
// descriptor but for `layer`, which only names the dropout sites and is not read here).
int timhip_layer_bwd_weights_pair(const TimDesc* dp, const void* x_in_T_a, const void* saved_a, const void* dy_a, const TimLayerGrads* ga,
                                  const void* x_in_T_b, const void* saved_b, const void* dy_b, const TimLayerGrads* gb,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  if (!dp || !x_in_T_a || !saved_a || !dy_a || !ga || !x_in_T_b || !saved_b || !dy_b || !gb || !workspace) return TIMHIP_EINVAL;
  const TimDesc& d = *dp;
  int rc = check_layer_desc(d);
  if (rc) return rc;
  const int M = d.B * d.S, E = d.E, FF = d.FF, prec = d.precision;
  if (!h16_storage(prec) || (d.reserved & TIMHIP_DESC_WGRAD_SEPARATE) || ((size_t)E * E) % 4 || ((size_t)E * FF) % 4) return TIMHIP_EUNSUPPORTED;
  const SavedLayout L = saved_layout(d);
  const DyLayout Y = dy_layout(d);
  const int acc = (d.reserved & TIMHIP_DESC_WGRAD_OVERWRITE) ? 0 : 1;
  const float* gs_out = (prec == TIMHIP_PREC_F16 && d.grad_scale) ? d.grad_scale + 1 : nullptr;
  TimWgradItem it[8];
  const void* xs[2] = {x_in_T_a, x_in_T_b};
  const char* svs[2] = {(const char*)saved_a, (const char*)saved_b};
  const char* ybs[2] = {(const char*)dy_a, (const char*)dy_b};
  const TimLayerGrads* gs[2] = {ga, gb};
  for (int h = 0; h < 2; ++h) {
    const char* sv = svs[h]; const char* yb = ybs[h]; const TimLayerGrads* g = gs[h];
    it[4 * h + 0] = TimWgradItem{yb + Y.df, sv + L.h, g->l2_w, g->l2_b, E, FF, E, FF};
    it[4 * h + 1] = TimWgradItem{yb + Y.du, sv + L.x1t, g->l1_w, g->l1_b, FF, E, FF, E};
    it[4 * h + 2] = TimWgradItem{yb + Y.da, sv + L.o, g->out_w, g->out_b, E, E, E, E};
    it[4 * h + 3] = TimWgradItem{yb + Y.dqkv, xs[h], g->in_w, g->in_b, 3 * E, E, 3 * E, E};
  }
  return tim_wgrad_group_h16(prec, it, 8, M, acc, workspace, workspace_bytes, gs_out, (hipStream_t)stream);
}

// 1: timhip_layer_bwd_weights_pair runs this descriptor's two layers as ONE round of eight-phase tiles (hosts defer a layer's
// weight gradients to its neighbour's only then); 0: no gain from pairing
int timhip_layer_wgrad_pair_wins(const TimDesc* dp) {
  if (!dp || check_layer_desc(*dp)) return 0;
  const TimDesc& d = *dp;
  if (!h16_storage(d.precision) || (d.reserved & TIMHIP_DESC_WGRAD_SEPARATE)) return 0;
  const int E = d.E, FF = d.FF;
  TimWgradItem it[8];
  for (int h = 0; h < 2; ++h) {
    it[4 * h + 0] = TimWgradItem{nullptr, nullptr, nullptr, nullptr, E, FF, E, FF};
    it[4 * h + 1] = TimWgradItem{nullptr, nullptr, nullptr, nullptr, FF, E, FF, E};
    it[4 * h + 2] = TimWgradItem{nullptr, nullptr, nullptr, nullptr, E, E, E, E};
    it[4 * h + 3] = TimWgradItem{nullptr, nullptr, nullptr, nullptr, 3 * E, E, 3 * E, E};
  }
  return tim_wgrad_p8_wins(it, 8, d.B * d.S) ? 1 : 0;
}

// single-stream form: data chain followed by the weight gradients
int timhip_layer_bwd(const TimDesc* dp, const TimLayerParams* w, const void* x_in_T, const void* saved, float* dx_out,
                     float* dx_in, const TimLayerGrads* g, void* workspace, size_t workspace_bytes, void* stream) {
  if (!dp) return TIMHIP_EINVAL;
  const WsLayout W = ws_layout(*dp);
  const size_t need = W.total + dy_layout(*dp).total;
  if (workspace_bytes < need) return TIMHIP_EWORKSPACE;
  char* ws = (char*)workspace;
  void* dy = ws + W.total;
  int rc = timhip_layer_bwd_data(dp, w, saved, dx_out, dx_in, dy, g, ws, W.total, stream);
  if (rc) return rc;
  return timhip_layer_bwd_weights(dp, x_in_T, saved, dy, g, ws + W.tA, W.wg_bytes, stream);
}


int timhip_layer_bwd_split(const TimDesc* dp, const TimLayerParams* w, const void* x_in_T, const void* saved, const float* dx_out,
                           const void* dx_out_add, float* dx_in, void* dx_in_add, const TimLayerGrads* g, void* workspace,
                           size_t workspace_bytes, void* stream) {
  if (!dp) return TIMHIP_EINVAL;
  const WsLayout W = ws_layout(*dp);
  const size_t need = W.total + dy_layout(*dp).total;
  if (workspace_bytes < need) return TIMHIP_EWORKSPACE;
  char* ws = (char*)workspace;
  void* dy = ws + W.total;
  int rc = timhip_layer_bwd_data_split(dp, w, saved, dx_out, dx_out_add, dx_in, dx_in_add, dy, g, ws, W.total, stream);
  if (rc) return rc;
  return timhip_layer_bwd_weights(dp, x_in_T, saved, dy, g, ws + W.tA, W.wg_bytes, stream);
}

}  // extern "C"
