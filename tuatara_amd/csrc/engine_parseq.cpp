// PARSeq forward (the TorchScript recogniser run at tuatara.cpp:307): ViT-S encoder, KV-cached AR decode with upstream's early exit, refinement pass.
#include "engine.h"

namespace ttr {

unsigned long long* g_dec_dbg = nullptr;

void Engine::sgemm(const Linear& L, const void* in_planes, int M, void* out, int out_ld, int act, int out_planes,
           float* out_f32, int out_f32_ld, const float* resid, int resid_ld, int np, int resid_mod, int out_full_cols,
           const char* kind, int x_tiled, int out_tiled) {
  if (!L.ws.p) throw std::runtime_error("split GEMM: the layer has no weight planes");
  range_tag("parseq." + range_scope + (kind ? kind : "linear"));
  ConvParams p{};
  p.out_full_cols = out_full_cols;
  p.in0 = in_planes; p.C0 = L.k; p.B = 1; p.H = 1; p.W = M; p.ks = 1; p.dil = 1;
  p.wgt = L.ws.p; p.bias = L.b.as<float>(); p.split = np; p.out_scale = L.inv_scale; p.out_planes = out_planes == 1 ? 3 : out_planes;   // (1 = triples)
  p.wgt_tiled = tn.sp_tiled_w ? L.wst.p : nullptr;
  p.x_tiled = x_tiled; p.out_tiled = out_tiled;
  p.store_policy = out_tiled == 2 && tn.sp_hidden16 == 2 ? 1 : 0;      // (16-row pieces: whole lines per store instruction, so the plane stores may stream)
  p.out = out; p.out_ld = out_ld; p.out_f32 = out_f32; p.out_f32_ld = out_f32_ld; p.resid = resid; p.resid_ld = resid_ld; p.resid_mod = resid_mod;
  p.Cout = L.cout_valid ? L.cout_valid : L.cout; p.M = M; p.act = act;
  p.skip = cur_skip; p.skip_n = cur_skip_n;
  const bool skinny = tn.skinny_split && np == 4 && M <= tn.skinny_max_rows && !x_tiled && !out_tiled && gemm_skx_eligible(p);
  if (!skinny) { if (const char* e = gemm2_check(p)) throw std::runtime_error(e); }
  // few rows (the AR steps: one row per crop): one memory round trip per small workgroup instead of a ring of K steps on a handful of 128-row tiles
  if (skinny) {
    timed(kind ? kind : "split linear (triples, skinny)", 2.0 * M * p.Cout * L.k, 2.0 * M * p.Cout * L.k * np, [&] { launch_gemm_skx(p, stream); });
    return;
  }
  timed(kind ? kind : (np == 3 ? "split linear (pairs)" : "split linear (triples)"), 2.0 * M * L.cout * L.k, 2.0 * M * L.cout * L.k * np, [&] { launch_gemm2(p, 0, stream); });
}

void Engine::ln_gemm(const float* x, const std::string& ln_name, float eps, void* scratch, const Linear& L, int M, void* out, int out_ld, int act,
             float* out_f32, int out_f32_ld) {
  if (tn.ln_fuse && prec == kBF16 && L.k == 384 && M <= skinny_max_rows()) {
    ConvParams p{};
    p.ln_in = x; p.ln_ld = 384; p.ln_gamma = pqf.at(ln_name + ".weight").as<float>(); p.ln_beta = pqf.at(ln_name + ".bias").as<float>(); p.ln_eps = eps;
    p.C0 = L.k; p.B = 1; p.H = 1; p.W = M; p.ks = 1; p.dil = 1;
    p.wgt = L.w.p; p.bias = L.b.as<float>();
    p.out = out; p.out_ld = out_ld; p.out_f32 = out_f32; p.out_f32_ld = out_f32_ld;
    p.Cout = L.cout; p.M = M; p.act = act;
    p.skip = cur_skip; p.skip_n = cur_skip_n;
    igemm(p, 2.0 * M * L.cout * L.k);
    return;
  }
  ln(x, ln_name, eps, scratch, M);
  gemm(L, scratch, M, out, out_ld, act, out_f32, out_f32_ld);
}

void Engine::gemm(const Linear& L, const void* in, int M, void* out, int out_ld, int act, float* out_f32, int out_f32_ld,
          const float* resid, int resid_ld, int resid_mod) {
  ConvParams p{};
  p.in0 = in; p.C0 = L.k; p.B = 1; p.H = 1; p.W = M; p.ks = 1; p.dil = 1;
  p.wgt = L.w.p; p.bias = L.b.as<float>();
  p.out = out; p.out_ld = out_ld; p.out_f32 = out_f32; p.out_f32_ld = out_f32_ld;
  p.resid = resid; p.resid_ld = resid_ld; p.resid_mod = resid_mod;
  p.Cout = L.cout_valid ? L.cout_valid : L.cout; p.M = M; p.act = act;
  p.skip = cur_skip; p.skip_n = cur_skip_n;
  igemm(p, 2.0 * M * p.Cout * L.k);
}

void Engine::ln(const float* x, const std::string& name, float eps, void* out, int M) {
  launch_layernorm(prec, x, 384, pqf.at(name + ".weight").as<float>(), pqf.at(name + ".bias").as<float>(), eps, out, 384, M, 384, stream, cur_skip, cur_skip_n);
}

void Engine::decoder_tail_split(const void* sa, int N, int R, const float* resid_pos, int resid_mod, float* tgt, void* pa, void* pb, void* p1536, float* q384,
                        void* t384, const void* kvmem, float* logits_out, int logits_ld, const int* done_tok, int done_col) {
  const int rows = N * R;
  const std::string d = "decoder.layers.0.";
  auto lnp = [&](const std::string& nm, void* out) {
    range_tag("parseq." + nm);
    launch_layernorm_planes(tgt, 384, pqf.at(nm + ".weight").as<float>(), pqf.at(nm + ".bias").as<float>(), 1e-5f, out, rows, stream, 3, cur_skip, cur_skip_n);
  };
  // LayerNorm + linear: two launches, or - a page's worth of rows (the AR steps of the latency regime) - the skinny kernel with the LayerNorm
  // as its prologue (gemm_skx.hip, LNP): the same arithmetic, one dependent launch less
  auto ln_lin = [&](const std::string& nm, const Linear& L, void* out, int out_ld, int act, int out_planes, float* out_f32, int out_f32_ld, const char* kind) {
    range_tag("parseq." + range_scope + kind);
    ConvParams p{};
    p.ln_in = tgt; p.ln_ld = 384; p.ln_gamma = pqf.at(nm + ".weight").as<float>(); p.ln_beta = pqf.at(nm + ".bias").as<float>(); p.ln_eps = 1e-5f;
    p.C0 = L.k; p.B = 1; p.H = 1; p.W = rows; p.ks = 1; p.dil = 1;
    p.wgt = L.ws.p; p.bias = L.b.as<float>(); p.split = 4; p.out_scale = L.inv_scale; p.out_planes = out_planes == 1 ? 3 : out_planes;
    p.out = out; p.out_ld = out_ld; p.out_f32 = out_f32; p.out_f32_ld = out_f32_ld;
    p.Cout = L.cout_valid ? L.cout_valid : L.cout; p.M = rows; p.act = act;
    p.skip = cur_skip; p.skip_n = cur_skip_n;
    if (tn.skx_ln_fuse && tn.skinny_split && L.ws.p && gemm_skx_ln_eligible(p)) {
      timed(kind, 2.0 * rows * p.Cout * L.k, 2.0 * rows * p.Cout * L.k * 4, [&] { launch_gemm_skx(p, stream); });
      return;
    }
    lnp(nm, pa);
    sgemm(L, pa, rows, out, out_ld, act, out_planes, out_f32, out_f32_ld, nullptr, 0, 4, 0, 0, kind);
  };
  range_tag("parseq.decoder.self_attn.out_proj");
  sgemm(pq.at("self_out"), sa, rows, nullptr, 0, kActNone, 0, tgt, 384, resid_pos, 384, 4, resid_mod);      // tgt = query + self_attn
  ln_lin(d + "norm1", pq.at("cross_q"), q384, 384, kActNone, 0, nullptr, 0, "dec.norm1 + cross_q");           // fp32 queries for the attention kernel
  range_tag("parseq.decoder.cross_attn");
  launch_dec_cross_attn(kF32, q384, kvmem, pb, N, R, stream, cur_skip, cur_skip_n, done_tok, done_col, 3);    // planes out
  sgemm(pq.at("cross_out"), pb, rows, nullptr, 0, kActNone, 0, tgt, 384, tgt, 384);                            // tgt += cross_attn
  ln_lin(d + "norm2", pq.at("ffn1"), p1536, 1536, kActGelu, 3, nullptr, 0, "dec.norm2 + ffn1");
  sgemm(pq.at("ffn2"), p1536, rows, nullptr, 0, kActNone, 0, tgt, 384, tgt, 384);                              // tgt += ffn
  const Linear& head = pq.at("head");
  if (head.ws.p && tn.skinny_split && rows <= tn.skinny_max_rows)   // few rows: the head as a skinny split linear on the final norm's planes
    ln_lin("decoder.norm", head, nullptr, 0, kActNone, 0, logits_out, logits_ld, "dec.norm + head (skinny)");
  else ln_gemm(tgt, "decoder.norm", 1e-5f, t384, head, rows, nullptr, 0, kActNone, logits_out, logits_ld);
}

void Engine::decoder_tail(const void* sa, int N, int R, const float* resid_pos, int resid_mod, float* tgt, void* t384, void* t384b, void* t1536,
                  const void* kvmem, float* logits_out, int logits_ld, const int* done_tok, int done_col) {
  const int rows = N * R;
  const std::string d = "decoder.layers.0.";
  gemm(pq.at("self_out"), sa, rows, nullptr, 0, kActNone, tgt, 384, resid_pos, 384, resid_mod);      // tgt = query + self_attn
  ln_gemm(tgt, d + "norm1", 1e-5f, t384, pq.at("cross_q"), rows, t384b, 384, kActNone);
  launch_dec_cross_attn(prec, t384b, kvmem, t384, N, R, stream, cur_skip, cur_skip_n, done_tok, done_col);
  // (the fused block kernel is one persistent workgroup per CU over 128-row panels: when the panels fill the last round of CUs badly -
  // 1280 crops x 26 rows = 260 panels on 256 CUs: two rounds for 1.02 - the separate GEMMs are faster: 12.17 vs 12.27 ms per forward)
  const int dec_panels = (rows + 127) / 128, dec_cus = device_cu_count(256), dec_rounds = (dec_panels + dec_cus - 1) / dec_cus;
  const bool dec_fill = tn.dec_mlp_fused == 2 || dec_panels * 100 >= 65 * dec_rounds * dec_cus;
  if (R > 1 && prec == kBF16 && gemm_config() >= 0 && tn.dec_mlp_fused && rows >= tn.dec_mlp_min_rows && dec_fill) {
    // refinement pass (26 rows per crop): the block behind the cross-attention is an encoder block's second half with other weights —
    // out projection + residual, norm2, linear1, GELU, linear2, residual, and the final norm as the "next LayerNorm" — one launch
    MlpParams q{};
    q.x = tgt; q.x_out = tgt; q.M = rows;
    q.ln_g = pqf.at(d + "norm2.weight").as<float>(); q.ln_b = pqf.at(d + "norm2.bias").as<float>(); q.ln_eps = 1e-5f;
    q.w1p = dec_ffn1_packed.as<bf16>(); q.b1 = pq.at("ffn1").b.as<float>();
    q.w2p = dec_ffn2_packed.as<bf16>(); q.b2 = pq.at("ffn2").b.as<float>();
    q.nln_g = pqf.at("decoder.norm.weight").as<float>(); q.nln_b = pqf.at("decoder.norm.bias").as<float>(); q.nln_eps = 1e-5f; q.nln_out = (bf16*)t384b;
    q.att = (const bf16*)t384; q.wpp = dec_co_packed.as<bf16>(); q.bp = pq.at("cross_out").b.as<float>();
    timed("mlp_fused (refinement block)", 2.0 * rows * 384 * 1536 * 2 + 2.0 * rows * 384 * 384, 2.0 * rows * 384 * 1536 * 2 + 2.0 * rows * 384 * 384, [&] { launch_mlp_fused(q, stream); });
    gemm(pq.at("head"), t384b, rows, nullptr, 0, kActNone, logits_out, logits_ld);
    return;
  }
  gemm(pq.at("cross_out"), t384, rows, nullptr, 0, kActNone, tgt, 384, tgt, 384, 0);                 // tgt += cross_attn
  ln_gemm(tgt, d + "norm2", 1e-5f, t384, pq.at("ffn1"), rows, t1536, 1536, kActGelu);
  gemm(pq.at("ffn2"), t1536, rows, nullptr, 0, kActNone, tgt, 384, tgt, 384, 0);                     // tgt += ffn
  ln_gemm(tgt, "decoder.norm", 1e-5f, t384, pq.at("head"), rows, nullptr, 0, kActNone, logits_out, logits_ld);
}

void Engine::parseq_forward(const uint8_t* d_crops, int N, float* d_logits, float* d_ar, int* d_ids) {
  if (N <= 0) return;
  // A very large crop batch (64 pages of ~150 boxes) goes through in even groups: the refinement pass's widest planes tensor (26 rows per crop x 1536 x 6 bytes)
  // must stay inside the 2 GiB window of 32-bit buffer offsets (8962 crops), and the workspaces stay bounded.  Crops are independent (batch-invariant logits,
  // tests): grouping changes nothing but the kernels' shapes; the AR loop's early exit then applies per group.
  constexpr int kMaxCrops = 4096;
  if (N > kMaxCrops) {
    const int groups = (N + kMaxCrops - 1) / kMaxCrops, per = (N + groups - 1) / groups;
    for (int g0 = 0; g0 < N; g0 += per) {
      const int n = std::min(per, N - g0);
      parseq_forward(d_crops + (size_t)g0 * 32 * 128 * 3, n, d_logits + (size_t)g0 * 26 * 95, d_ar ? d_ar + (size_t)g0 * 26 * 95 : nullptr, d_ids + (size_t)g0 * 26);
    }
    return;
  }
  prof_stage = 1;
  const int M = N * 128, E = 384;
  const int patch_ld = pq.at("patch").k;   // 96, or 128 in bf16 mode (zero-padded)
  range_tag("parseq.encoder.patch_embed");
  void* patches = (pq_ws[0].ensure((size_t)M * patch_ld * es), pq_ws[0].p);
  float* x = (float*)(pq_ws[1].ensure((size_t)M * E * 4), pq_ws[1].p);
  void* t384 = (pq_ws[2].ensure((size_t)std::max(M, N * 26) * E * es), pq_ws[2].p);
  void* tbig = (pq_ws[3].ensure((size_t)M * 1536 * es), pq_ws[3].p);
  void* att = (pq_ws[4].ensure((size_t)std::max(M, N * 26) * E * es), pq_ws[4].p);
  launch_patchify(prec, d_crops, patches, N, patch_ld, stream);
  gemm(pq.at("patch"), patches, M, nullptr, 0, kActNone, x, E, pqf.at("encoder.pos_embed").as<float>(), E, 128);
  const bool enc_split = prec == kSplit && tn.split_gemm && tn.split_planes;
  if (enc_split) {
    // split-operand engines: LayerNorm, GEMM epilogues and the attention kernel hand each other planes (split.h); only the residual
    // stream x is fp32.  Crop groups keep the widest planes tensor (the MLP hidden: 1536 x 6 bytes per row) inside the 2 GiB window.
    int CHS = std::max(1, std::min(N, (int)((((size_t)1 << 31) - 1) / ((size_t)128 * 1536 * 6))));
    if (tn.enc_chunk > 0) CHS = std::min(CHS, tn.enc_chunk);   // (experiment knob: crop groups whose residual stream and LayerNorm planes stay in the Infinity Cache)
    if (N > CHS) { const int groups = (N + CHS - 1) / CHS; CHS = (N + groups - 1) / groups; }   // even groups: 2560 crops = 1280 + 1280, not 1820 + 740
    void* lnp = (pq_ws[11].ensure((size_t)M * E * 6), pq_ws[11].p);                       // LayerNorm output planes (whole batch: the memory at the end)
    void* bigp = (pq_ws[12].ensure((size_t)std::min(N, CHS) * 128 * 1536 * 6), pq_ws[12].p);   // qkv / MLP hidden planes
    void* attp = (pq_ws[13].ensure((size_t)std::min(N, CHS) * 128 * E * 6), pq_ws[13].p);      // attention output planes
    auto lnp_at = [&](int c0) { return (char*)lnp + (size_t)c0 * 128 * E * 6; };
    const int lnpl = tn.enc_ln_pairs ? 2 : 3;        // planes of the LayerNorm outputs that feed qkv / fc1 (pairs: three MFMAs per product there)
    // the planes the encoder's GEMMs hand each other as their loaders' 1-KiB pieces (ConvParams::x_tiled): a wave instruction of the LDS-DMA fetches one
    // contiguous KiB instead of eight 128-byte rows 1.5 - 6 KB apart - 71 against 31 GB/s per CU for a lone four-wave workgroup (tools/micro/dma_depth.hip)
    // (measured: the recogniser pass at 1280 crops 34.96 -> 34.22 ms; a single page's 40 crops unchanged.  Groups small enough for the skinny projection keep the rows.)
    const int xt = tn.sp_tiled_x && tn.qkv_attn_split && lnpl == 2 && tn.enc_fc2_pairs && std::min(CHS, N) * 128 > tn.skinny_max_rows ? 1 : 0;
    for (int c0 = 0; c0 < N; c0 += CHS) {
      const int nc = std::min(CHS, N - c0), Mc = nc * 128;
      float* xc = x + (size_t)c0 * 128 * E;
      for (int l = 0; l < 12; ++l) {
        const std::string p = "encoder.blocks." + std::to_string(l) + ".";
        range_scope = p;
        range_tag("parseq." + p + "norm1");
        launch_layernorm_planes(xc, E, pqf.at(p + "norm1.weight").as<float>(), pqf.at(p + "norm1.bias").as<float>(), 1e-6f, lnp_at(c0), Mc, stream, lnpl, nullptr, 0, xt);
        if (tn.qkv_attn_split && lnpl == 2) {   // one launch: the attention of a (crop, head) is the epilogue of its 128 x 192 qkv tile
          const Linear& L = pq.at(p + "qkv_hm");
          range_tag("parseq." + p + "attn (qkv + attention)");
          // executed flops: qkv on pairs (x 3), Q K^T on a triple and a pair (x 4), P V on pairs (x 3)
          const double qa = 2.0 * Mc * 3 * E * E, aa = 2.0 * 2 * nc * 6 * 128.0 * 128 * 64;
          timed("enc.qkv+attention: gemm_sp_kernel<128,192,NP=3,EPI=1>", qa + aa, qa * 3 + aa * 3.5,
                [&] { launch_qkv_attn_split(lnp_at(c0), L.ws.p, L.b.as<float>(), L.inv_scale, attp, nc, stream, tn.sp_tiled_w ? L.wst.p : nullptr, xt, xt); });
        } else {
        sgemm(pq.at(p + "qkv"), lnp_at(c0), Mc, bigp, 3 * E, kActNone, 1, nullptr, 0, nullptr, 0, lnpl + 1, 0, tn.qkv_kv_pairs ? E : 0, "enc.qkv");   // (K, V: read as pairs)
        range_tag("parseq." + p + "attn.qkv"), launch_attn_enc_split(bigp, attp, nc, stream);
        }
        sgemm(pq.at(p + "proj"), attp, Mc, nullptr, 0, kActNone, 0, xc, E, xc, E, 4, 0, 0, "enc.proj", xt, 0);
        range_tag("parseq." + p + "norm2");
        launch_layernorm_planes(xc, E, pqf.at(p + "norm2.weight").as<float>(), pqf.at(p + "norm2.bias").as<float>(), 1e-6f, lnp_at(c0), Mc, stream, lnpl, nullptr, 0, xt);
        const int hpl = tn.enc_fc2_pairs ? 2 : 3;                                      // planes of the MLP's hidden activation
        const int hx = xt && tn.sp_hidden16 && hpl == 2 ? 2 : xt;                       // layout of the hidden planes (ConvParams::out_tiled / x_tiled)
        sgemm(pq.at(p + "fc1"), lnp_at(c0), Mc, bigp, 4 * E, kActGelu, hpl, nullptr, 0, nullptr, 0, lnpl + 1, 0, 0, "enc.fc1 + GELU", xt, hx);
        sgemm(pq.at(p + "fc2"), bigp, Mc, nullptr, 0, kActNone, 0, xc, E, xc, E, hpl + 1, 0, 0, "enc.fc2", hx, 0);
      }
      range_scope.clear();
      range_tag("parseq.encoder.norm");
      launch_layernorm_planes(xc, E, pqf.at("encoder.norm.weight").as<float>(), pqf.at("encoder.norm.bias").as<float>(), 1e-6f, lnp_at(c0), Mc, stream);
    }
  }
  // The 12 encoder blocks run over groups of crops so that a group's widest intermediates (qkv, the MLP hidden) are
  // re-read from the 256 MiB Infinity Cache rather than from HBM (tn.enc_chunk crops per group; 0 = one group).
  // the fused MLP block needs a panel of 128 rows per CU to fill the chip: below ~2 panels per CU the separate GEMMs win
  const bool mlp_fused = prec == kBF16 && gemm_config() >= 0 && (tn.mlp_fused == 2 || (tn.mlp_fused == 1 && M >= tn.mlp_min_rows));
  const int CH = (tn.enc_chunk > 0 && !mlp_fused) ? tn.enc_chunk : N;
  for (int c0 = 0; c0 < N && !enc_split; c0 += CH) {
    const int nc = std::min(CH, N - c0), Mc = nc * 128;
    float* xc = x + (size_t)c0 * 128 * E;
    if (mlp_fused) ln(xc, "encoder.blocks.0.norm1", 1e-6f, t384, Mc);
    for (int l = 0; l < 12; ++l) {
      std::string p = "encoder.blocks." + std::to_string(l) + ".";
      if (!mlp_fused) ln(xc, p + "norm1", 1e-6f, t384, Mc);
      if (prec == kBF16 && gemm_config() >= 0 && (tn.qkv_attn == 2 || (tn.qkv_attn == 1 && nc >= tn.qkv_attn_min)) && (size_t)Mc * E * 2 < ((size_t)1 << 31)) {   // (32-bit buffer offsets)
        const Linear& L = pq.at(p + "qkv");
        timed("qkv_attn_kernel", 2.0 * Mc * E * 3 * E, 2.0 * Mc * E * 3 * E, [&] { launch_qkv_attn((const bf16*)t384, L.w.as<bf16>(), L.b.as<float>(), (bf16*)att, nc, stream); });
      } else {
        gemm(pq.at(p + "qkv"), t384, Mc, tbig, 3 * E, kActNone);
        launch_attn_enc(prec, tbig, att, nc, stream);
      }
      const bool proj_in = mlp_fused && tn.mlp_proj;   // the projection runs inside the fused block kernel
      if (!proj_in) gemm(pq.at(p + "proj"), att, Mc, nullptr, 0, kActNone, xc, E, xc, E, 0);
      if (mlp_fused) {
        // norm2 + fc1 + GELU + fc2 + residual in one kernel; it also leaves the next LayerNorm (the next block's norm1, or
        // the encoder's final norm = the decoder's memory) in t384
        const std::string nx = l < 11 ? "encoder.blocks." + std::to_string(l + 1) + ".norm1" : std::string("encoder.norm");
        MlpParams q{};
        q.x = xc; q.x_out = xc; q.M = Mc;
        q.ln_g = pqf.at(p + "norm2.weight").as<float>(); q.ln_b = pqf.at(p + "norm2.bias").as<float>(); q.ln_eps = 1e-6f;
        q.w1p = fc1_packed[l].as<bf16>(); q.b1 = pq.at(p + "fc1").b.as<float>();
        q.w2p = fc2_packed[l].as<bf16>(); q.b2 = pq.at(p + "fc2").b.as<float>();
        q.nln_g = pqf.at(nx + ".weight").as<float>(); q.nln_b = pqf.at(nx + ".bias").as<float>(); q.nln_eps = 1e-6f; q.nln_out = (bf16*)t384;
        if (proj_in) { q.att = (const bf16*)att; q.wpp = proj_packed[l].as<bf16>(); q.bp = pq.at(p + "proj").b.as<float>(); }
        q.no_x_store = l == 11;   // behind the last block only the final norm (the decoder's memory) is read
        const double mf = 2.0 * Mc * E * 4 * E * 2 + (proj_in ? 2.0 * Mc * E * E : 0.0);
        timed("mlp_fused_kernel", mf, mf, [&] { launch_mlp_fused(q, stream); });
        continue;
      }
      ln(xc, p + "norm2", 1e-6f, t384, Mc);
      gemm(pq.at(p + "fc1"), t384, Mc, tbig, 4 * E, kActGelu);
      gemm(pq.at(p + "fc2"), tbig, Mc, nullptr, 0, kActNone, xc, E, xc, E, 0);
    }
  }
  if (!mlp_fused && !enc_split) ln(x, "encoder.norm", 1e-6f, t384, M);       // memory
  void* kvmem = (pq_ws[5].ensure((size_t)M * 768 * es), pq_ws[5].p);
  if (enc_split) {
    const int rows_max = (int)((((size_t)1 << 31) - 1) / ((size_t)E * 6));
    for (int r0 = 0; r0 < M; r0 += rows_max) {
      const int rr = std::min(rows_max, M - r0);
      sgemm(pq.at("cross_kv"), (char*)pq_ws[11].p + (size_t)r0 * E * 6, rr, (char*)kvmem + (size_t)r0 * 768 * 4, 768, kActNone, 0);
    }
  } else
  gemm(pq.at("cross_kv"), t384, M, kvmem, 768, kActNone);

  // ---- decoder
  range_scope = "decoder.";
  struct ScopeReset { std::string& s; ~ScopeReset() { s.clear(); } } scope_reset{range_scope};
  void* kvcache = (pq_ws[6].ensure((size_t)N * 26 * 768 * es), pq_ws[6].p);
  if (kvcache_zeroed != pq_ws[6].cap) {   // slots behind an early exit keep older (finite) rows; they are masked, but 0 x NaN is not 0
    TTR_HIP_CHECK(hipMemsetAsync(kvcache, 0, pq_ws[6].cap, stream));
    kvcache_zeroed = pq_ws[6].cap;
  }
  float* tgt = (float*)(pq_ws[7].ensure((size_t)N * 26 * E * 4), pq_ws[7].p);
  void* d384b = (pq_ws[8].ensure((size_t)N * 26 * E * es), pq_ws[8].p);
  void* d1536 = (pq_ws[9].ensure((size_t)N * 26 * 1536 * es), pq_ws[9].p);
  float* step_logits = (float*)(pq_ws[10].ensure((size_t)N * 26 * 95 * 4), pq_ws[10].p);
  // split-operand engines: the decoder's layers hand each other planes (decoder_tail_split)
  const bool dec_split = prec == kSplit && tn.split_gemm && tn.split_planes && tn.dec_planes;
  void *dpa = nullptr, *dpb = nullptr, *dp1536 = nullptr, *dsa = nullptr;
  if (dec_split) {
    dpa = (pq_ws[14].ensure((size_t)N * 26 * E * 6), pq_ws[14].p); dpb = (pq_ws[15].ensure((size_t)N * 26 * E * 6), pq_ws[15].p);
    dp1536 = (pq_ws[16].ensure((size_t)N * 26 * 1536 * 6), pq_ws[16].p); dsa = (pq_ws[17].ensure((size_t)N * 26 * E * 6), pq_ws[17].p);
  }
  tokens.ensure((size_t)N * 26 * 4);
  int* tk = tokens.as<int>();
  launch_fill_i32(tk, 96, N * 26, 1, stream);   // PAD
  launch_fill_i32(tk, 95, N, 26, stream);       // BOS at position 0
  const float* emb = pqf.at("text_embed.embedding.weight").as<float>();
  const float* posq = pqf.at("pos_queries").as<float>();
  const std::string d = "decoder.layers.0.";
  const float* gc = pqf.at(d + "norm_c.weight").as<float>();
  const float* bc = pqf.at(d + "norm_c.bias").as<float>();
  float* ar = d_ar ? d_ar : step_logits;
  const int nsteps = d_ar ? 26 : 25;  // the 26th AR step only feeds logits the refinement pass discards
  // Fused persistent AR kernel (dec_fused.hip): ~150 us per step whatever N is (every workgroup is bound by its own
  // ~12 B/clk fetch rate on the weight and K/V streams).  With the skinny per-step GEMMs (gemm_sk.hip) the kernel-per-op
  // loop is faster up to ~1200 crops (measured at 40 / 320 / 614 crops), so the fused kernel is only picked beyond that.
  const bool fused_ar = prec == kBF16 && tn.decoder_mode != 0 && (tn.decoder_mode == 4 || tn.decoder_mode == 8 || tn.decoder_mode == 16 || N > 2048);
  auto dec_params = [&]() {
    DecArParams q{};
    auto W = [&](const char* k) { return pq.at(k).w.as<bf16>(); };
    auto Bv = [&](const char* k) { return pq.at(k).b.as<float>(); };
    auto V = [&](const std::string& k) { return pqf.at(k).as<float>(); };
    q.w_selfkv = W("self_kv"); q.w_selfout = W("self_out"); q.w_crossq = W("cross_q"); q.w_crossout = W("cross_out");
    q.w_ffn1 = W("ffn1"); q.w_ffn2 = W("ffn2"); q.w_head = W("head");
    q.b_selfkv = Bv("self_kv"); q.b_selfout = Bv("self_out"); q.b_crossq = Bv("cross_q"); q.b_crossout = Bv("cross_out");
    q.b_ffn1 = Bv("ffn1"); q.b_ffn2 = Bv("ffn2"); q.b_head = Bv("head");
    q.emb = emb; q.posq = posq; q.qself = qself.as<float>();
    q.g_c = gc; q.b_c = bc;
    q.g_1 = V(d + "norm1.weight"); q.b_1 = V(d + "norm1.bias"); q.g_2 = V(d + "norm2.weight"); q.b_2 = V(d + "norm2.bias");
    q.g_f = V("decoder.norm.weight"); q.b_f = V("decoder.norm.bias");
    q.kvmem = (const bf16*)kvmem; q.kvcache = (bf16*)kvcache; q.tokens = tk; q.ar_logits = d_ar;
    q.gelu_lut = gelu_lut_for_current_device();
    q.dbg = g_dec_dbg;
    q.N = N; q.nsteps = nsteps;
    return q;
  };
  if (fused_ar) {
    DecArParams q = dec_params();
    int G = tn.decoder_mode;
    if (G != 4 && G != 8 && G != 16) G = N <= 1024 ? 4 : 8;
    launch_dec_ar(q, G, stream);
  } else {
  prof_stage = 2;
  const bool tok_fuse = tn.tok_fuse && tn.ln_fuse && prec == kBF16 && N <= skinny_max_rows();
  // upstream PARSeq leaves its AR loop once every crop of the batch has emitted EOS (system.py): the bf16 engine counts them in the skinny
  // GEMM's token prologue, the fp32 / f16x4 engines in the argmax kernel; every kernel of a step returns at once when the counter has
  // reached N, and (ar_crop_exit) the attention kernels skip crops that are done - keys behind a crop's EOS are masked in the
  // refinement pass, so the refined logits do not depend on it (tests)
  const bool early = tn.ar_early_exit && (tok_fuse || prec != kBF16);
  if (early) {
    ar_done.ensure(64);
    TTR_HIP_CHECK(hipMemsetAsync(ar_done.p, 0, 4, stream));
    if (d_ar) TTR_HIP_CHECK(hipMemsetAsync(d_ar, 0, (size_t)N * 26 * 95 * 4, stream));   // steps behind the exit stay zero
    cur_skip = ar_done.as<int>(); cur_skip_n = N;
  }
  // with the early exit, the steps from ar_tail_step on are ONE launch of the fused kernel in its tail form: when every crop
  // has emitted EOS by then (the usual case: words are short) it returns at once, instead of ~9 returning launches per step
  const int tail_at = (early && prec == kBF16 && tn.ar_tail_step > 0 && tn.ar_tail_step < nsteps) ? tn.ar_tail_step : 26;
  struct SkipGuard { Engine& E; ~SkipGuard() { E.cur_skip = nullptr; E.cur_skip_n = 0; } } skip_guard{*this};   // also when a launch throws mid-loop
  int pend_argmax = -1;   // AR step whose logits still await their argmax (dec_embed_ln of the next step takes it)
  for (int i = 0; i < 26; ++i) {
    if (i == tail_at) {
      DecArParams q = dec_params();
      q.first_step = i; q.prev_logits = ar + (size_t)(i - 1) * 95; q.prev_ld = 26 * 95; q.skip = cur_skip; q.skip_n = cur_skip_n;
      if (!d_ar) q.ar_logits = nullptr;
      launch_dec_ar(q, N <= 1024 ? 4 : 8, stream);
      break;
    }
    if (tok_fuse) {   // token of step i = argmax of step i-1's logits, embedded and normalised in the GEMM's loader
      const Linear& L = pq.at("self_kv");
      ConvParams p{};
      p.ln_in = emb; p.ln_ld = 384; p.ln_gamma = gc; p.ln_beta = bc; p.ln_eps = 1e-5f;
      p.tok = tk; p.tok_ld = 26; p.tok_col = i; p.tok_emb = emb; p.tok_max = 96;
      if (i > 0) { p.tok_logits = ar + (size_t)(i - 1) * 95; p.tok_logits_ld = 26 * 95; p.tok_C = 95; p.tok_pos = posq + (size_t)(i - 1) * E; }
      if (early) { p.skip = cur_skip; p.skip_n = cur_skip_n; p.done_count = ar_done.as<int>(); p.tok_eos = 0; }
      p.C0 = L.k; p.B = 1; p.H = 1; p.W = N; p.ks = 1; p.dil = 1;
      p.wgt = L.w.p; p.bias = L.b.as<float>();
      p.out = (char*)kvcache + (size_t)i * 768 * es; p.out_ld = 26 * 768;
      p.Cout = L.cout; p.M = N; p.act = kActNone;
      igemm(p, 2.0 * N * L.cout * L.k);
    } else if (dec_split && tn.embed_fold && tn.skinny_split && N <= 256) {
      // a page's worth of crops: token (or the pending argmax) -> embedding -> norm_c -> self_kv as ONE skinny launch (gemm_skx.hip, token prologue)
      const Linear& L = pq.at("self_kv");
      ConvParams p{};
      p.tok = tk; p.tok_ld = 26; p.tok_col = i; p.tok_emb = emb; p.tok_max = 96; p.tok_pos = i > 0 ? posq + (size_t)(i - 1) * E : nullptr;
      if (pend_argmax >= 0) {
        p.tok_logits = ar + (size_t)pend_argmax * 95; p.tok_logits_ld = 26 * 95; p.tok_C = 95; p.done_count = early ? ar_done.as<int>() : nullptr; p.tok_eos = 0;
        pend_argmax = -1;
      }
      p.ln_gamma = gc; p.ln_beta = bc; p.ln_eps = 1e-5f;
      p.C0 = L.k; p.B = 1; p.H = 1; p.W = N; p.ks = 1; p.dil = 1;
      p.wgt = L.ws.p; p.bias = L.b.as<float>(); p.split = 4; p.out_scale = L.inv_scale;
      p.out = (char*)kvcache + (size_t)i * 768 * 4; p.out_ld = 26 * 768;
      p.Cout = L.cout; p.M = N; p.act = kActNone;
      p.skip = cur_skip; p.skip_n = cur_skip_n;
      if (!gemm_skx_ln_eligible(p)) throw std::runtime_error("AR step: the token prologue does not take this shape");
      timed("dec.embed + norm_c + self_kv (skinny)", 2.0 * N * L.cout * L.k, 2.0 * N * L.cout * L.k * 4, [&] { launch_gemm_skx(p, stream); });
    } else if (dec_split) {
      if (pend_argmax >= 0) {   // column i's token = argmax of step i - 1's logits, found by this kernel's waves (one launch less per step)
        launch_dec_embed_ln(prec, tk, emb, posq, gc, bc, 1e-5f, dpa, N, i, i + 1, stream, cur_skip, cur_skip_n, 3,
                            ar + (size_t)pend_argmax * 95, 26 * 95, 95, early ? ar_done.as<int>() : nullptr, 0);
        pend_argmax = -1;
      } else launch_dec_embed_ln(prec, tk, emb, posq, gc, bc, 1e-5f, dpa, N, i, i + 1, stream, cur_skip, cur_skip_n, 3);
      sgemm(pq.at("self_kv"), dpa, N, (char*)kvcache + (size_t)i * 768 * 4, 26 * 768, kActNone, 0);
    } else {
      launch_dec_embed_ln(prec, tk, emb, posq, gc, bc, 1e-5f, t384, N, i, i + 1, stream, cur_skip, cur_skip_n);
      gemm(pq.at("self_kv"), t384, N, (char*)kvcache + (size_t)i * 768 * es, 26 * 768, kActNone);
    }
    if (i >= nsteps) break;
    const int* crop_done = early && tn.ar_early_exit >= 1 && tn.ar_crop_exit ? tk : nullptr;
    if (dec_split) {
      launch_dec_self_attn(prec, qself.as<float>(), kvcache, tk, dsa, N, 1, i, 0, stream, cur_skip, cur_skip_n, 3);
      decoder_tail_split(dsa, N, 1, posq + (size_t)i * E, 1, tgt, dpa, dpb, dp1536, (float*)d384b, t384, kvmem, ar + (size_t)i * 95, 26 * 95, crop_done, i);
    } else {
    launch_dec_self_attn(prec, qself.as<float>(), kvcache, tk, att, N, 1, i, 0, stream, cur_skip, cur_skip_n);
    decoder_tail(att, N, 1, posq + (size_t)i * E, 1, tgt, t384, d384b, d1536, kvmem, ar + (size_t)i * 95, 26 * 95, crop_done, i);
    }
    // the latency regime (a page or two of crops): every remaining step is ~11 launches that return at once when the batch is done - half a
    // millisecond of them for ten-character words.  The host looks at the counter (one small synchronous read) and stops enqueuing instead
    const bool host_check = early && !tok_fuse && !streaming_recog && tn.ar_host_check > 0 && N <= 256 && i >= tn.ar_host_check && (i - tn.ar_host_check) % 4 == 0 && i + 1 < nsteps;
    if (i + 1 < 26 && !tok_fuse) {
      // the step's argmax: its own launch where the host is about to look at the counter (or nothing follows), else left to the next step's first kernel
      if (dec_split && tn.argmax_fold && !host_check) pend_argmax = i;
      else launch_argmax(ar + (size_t)i * 95, 26 * 95, 95, tk, 26, i + 1, N, stream, cur_skip, cur_skip_n, early ? ar_done.as<int>() : nullptr, 0);
    }
    if (host_check) {
      h_ar_done.ensure(64);
      TTR_HIP_CHECK(hipMemcpyAsync(h_ar_done.p, ar_done.p, 4, hipMemcpyDeviceToHost, stream));
      TTR_HIP_CHECK(hipStreamSynchronize(stream));
      if (*h_ar_done.as<int>() >= N) break;
    }
  }
  cur_skip = nullptr; cur_skip_n = 0;
  prof_stage = 1;
  }
  // ---- refinement pass (cloze mask + EOS key padding), R = 26 query rows per crop
  if (dec_split) {
    launch_dec_self_attn(prec, qself.as<float>(), kvcache, tk, dsa, N, 26, 0, 1, stream, nullptr, 0, 3);
    decoder_tail_split(dsa, N, 26, posq, 26, tgt, dpa, dpb, dp1536, (float*)d384b, t384, kvmem, d_logits, 95);
  } else {
  launch_dec_self_attn(prec, qself.as<float>(), kvcache, tk, att, N, 26, 0, 1, stream);
  decoder_tail(att, N, 26, posq, 26, tgt, t384, d384b, d1536, kvmem, d_logits, 95);
  }
  launch_argmax(d_logits, 95, 95, d_ids, 1, 0, N * 26, stream);
}

}  // namespace ttr
