// Developer hooks of include/tuatara_hip_debug.h: single-kernel test entry points, micro-benchmarks, the tuning registry.
#include "engine.h"

using namespace ttr;

extern "C" {

int ttr_dbg_conv(ttr_engine* e, const float* in0, int C0, const float* in1, int C1, int relu0, int relu1, int B, int H, int W, int ks, int dil,
                 const float* wgt, const float* bias, int Cout, int act, float* out) {
  TTR_GUARD_BEGIN
  Engine& E = *e->e;
  EngineScope lk(E);
  const size_t M = (size_t)B * H * W;
  const int K = ks * ks * (C0 + C1);
  DevBuf d0, d1, dout;
  Linear L;
  auto up = [&](DevBuf& d, const float* src, size_t nel) {
    d.ensure(nel * E.es);
    if (E.prec == kBF16) {
      std::vector<uint16_t> hbuf(nel);
      for (size_t i = 0; i < nel; ++i) hbuf[i] = f32_to_bf16_rne(src[i]);
      TTR_HIP_CHECK(hipMemcpy(d.p, hbuf.data(), nel * 2, hipMemcpyHostToDevice));
    } else TTR_HIP_CHECK(hipMemcpy(d.p, src, nel * 4, hipMemcpyHostToDevice));
  };
  up(d0, in0, M * C0);
  if (C1) up(d1, in1, M * C1);
  E.upload_linear(L, wgt, Cout, K, bias, Cout, K, nullptr, false);
  dout.ensure(M * Cout * 4);
  ConvParams p{};
  p.in0 = d0.p; p.C0 = C0; p.in1 = C1 ? d1.p : nullptr; p.C1 = C1; p.relu0 = relu0; p.relu1 = relu1;
  p.B = B; p.H = H; p.W = W; p.ks = ks; p.dil = dil; p.wgt = L.w.p; p.bias = bias ? L.b.as<float>() : nullptr;
  p.out = nullptr; p.out_f32 = dout.as<float>(); p.out_f32_ld = Cout; p.Cout = Cout; p.M = (int)M; p.act = act;
  const bool bf16_out = E.tn.dbg_bf16_out && E.prec == kBF16;
  if (bf16_out) { p.out = dout.p; p.out_ld = Cout; p.out_f32 = nullptr; p.out_f32_ld = 0; }
  launch_igemm(E.prec, p, E.stream);
  if (bf16_out) {
    std::vector<uint16_t> hb(M * Cout);
    TTR_HIP_CHECK(hipMemcpyAsync(hb.data(), dout.p, M * Cout * 2, hipMemcpyDeviceToHost, E.stream));
    TTR_HIP_CHECK(hipStreamSynchronize(E.stream));
    for (size_t i = 0; i < hb.size(); ++i) { const uint32_t u = (uint32_t)hb[i] << 16; memcpy(&out[i], &u, 4); }
    return 0;
  }
  TTR_HIP_CHECK(hipMemcpyAsync(out, dout.p, M * Cout * 4, hipMemcpyDeviceToHost, E.stream));
  TTR_HIP_CHECK(hipStreamSynchronize(E.stream));
  return 0;
  TTR_GUARD_END(-1)
}

// One split-operand linear layer on its own (tests): out[M][N] = act(x w^T + bias (+ resid)) through launch_gemm2's split mode (gemm_sp.hip's kernels
// where they apply), np = 3 (activation pairs) or 4 (triples); out_planes = 0 (the kernel writes fp32) or 2 / 3 (it writes f16 planes, joined here).
int ttr_dbg_split_gemm(ttr_engine* e, const float* x, int M, int K, const float* w, const float* bias, int N, int np, int act, int out_planes,
                       const float* resid, int cfg, float* out) {
  TTR_GUARD_BEGIN
  Engine& E = *e->e;
  EngineScope lk(E);
  if (E.prec != kSplit) throw std::runtime_error("ttr_dbg_split_gemm: f16x4 engines only");
  if ((np != 3 && np != 4) || (out_planes != 0 && out_planes != 2 && out_planes != 3)) throw std::runtime_error("ttr_dbg_split_gemm: np must be 3 or 4, out_planes 0, 2 or 3");
  const int ipl = np == 3 ? 2 : 3;
  DevBuf dx, dxp, dout, dres;
  Linear L;
  dx.ensure((size_t)M * K * 4); TTR_HIP_CHECK(hipMemcpy(dx.p, x, (size_t)M * K * 4, hipMemcpyHostToDevice));
  dxp.ensure((size_t)M * K * 2 * ipl);
  launch_split_planes(dx.as<float>(), K, dxp.p, M, K, 0, E.stream, ipl);
  E.upload_linear(L, w, N, K, bias, cfg == 7 ? (N + 7) / 8 * 8 : N, K, nullptr, false);   // (the skinny kernel takes any channel count: zero rows pad the planes)
  if (!L.ws.p) throw std::runtime_error("ttr_dbg_split_gemm: the layer has no weight planes");
  if (resid) { dres.ensure((size_t)M * N * 4); TTR_HIP_CHECK(hipMemcpy(dres.p, resid, (size_t)M * N * 4, hipMemcpyHostToDevice)); }
  dout.ensure((size_t)M * N * (out_planes ? 2 * out_planes : 4));
  ConvParams p{};
  p.in0 = dxp.p; p.C0 = K; p.B = 1; p.H = 1; p.W = M; p.ks = 1; p.dil = 1;
  if (E.tn.sp_tiled_w) E.tile_planes(L);                       // (what the engine's own layers hand gemm_sp.hip)
  p.wgt = L.ws.p; p.bias = bias ? L.b.as<float>() : nullptr; p.split = np; p.out_scale = L.inv_scale; p.out_planes = out_planes;
  p.wgt_tiled = E.tn.sp_tiled_w ? L.wst.p : nullptr;
  if (out_planes) { p.out = dout.p; p.out_ld = N; } else { p.out_f32 = dout.as<float>(); p.out_f32_ld = N; }
  p.resid = resid ? dres.as<float>() : nullptr; p.resid_ld = N;
  p.Cout = N; p.M = M; p.act = act;
  if (cfg == 7) launch_gemm_skx(p, E.stream);       // the skinny whole-K kernel (gemm_skx.hip: triples; throws for shapes it does not take)
  else {
    if (const char* err = gemm2_check(p)) throw std::runtime_error(err);
    launch_gemm2(p, cfg, E.stream);
  }
  if (!out_planes) {
    TTR_HIP_CHECK(hipMemcpyAsync(out, dout.p, (size_t)M * N * 4, hipMemcpyDeviceToHost, E.stream));
    TTR_HIP_CHECK(hipStreamSynchronize(E.stream));
    return 0;
  }
  std::vector<uint16_t> h((size_t)M * N * out_planes);
  TTR_HIP_CHECK(hipMemcpyAsync(h.data(), dout.p, h.size() * 2, hipMemcpyDeviceToHost, E.stream));
  TTR_HIP_CHECK(hipStreamSynchronize(E.stream));
  auto f16_to_f32 = [](uint16_t v) -> double {
    const int sgn = v >> 15, ex = (v >> 10) & 31, man = v & 1023;
    double r = ex == 0 ? std::ldexp((double)man, -24) : ex == 31 ? (man ? NAN : INFINITY) : std::ldexp((double)(man | 1024), ex - 25);
    return sgn ? -r : r;
  };
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      const uint16_t* row = h.data() + (size_t)m * out_planes * N;
      double v = f16_to_f32(row[n]), lo = f16_to_f32(row[N + n]);
      if (out_planes == 3) lo += f16_to_f32(row[2 * N + n]);
      out[(size_t)m * N + n] = (float)(v + lo / 2048.0);
    }
  return 0;
  TTR_GUARD_END(-1)
}

int ttr_dbg_mlp(ttr_engine* e, const float* x, int M, const float* ln_g, const float* ln_b, float eps, const float* w1, const float* b1, const float* w2,
                const float* b2, const float* nln_g, const float* nln_b, float* x_out, float* nln_out, const float* att, const float* wp, const float* bp) {
  TTR_GUARD_BEGIN
  Engine& E = *e->e;
  EngineScope lk(E);
  if (E.prec != kBF16) throw std::runtime_error("ttr_dbg_mlp: bf16 engines only");
  const int D = 384, H = 1536;
  DevBuf dx, dout, dg, db, dw1, db1, dw2, db2, dng, dnb, dn;
  auto upf = [&](DevBuf& d, const float* src, size_t n) { d.ensure(n * 4); TTR_HIP_CHECK(hipMemcpy(d.p, src, n * 4, hipMemcpyHostToDevice)); };
  upf(dx, x, (size_t)M * D); upf(dg, ln_g, D); upf(db, ln_b, D); upf(db1, b1, H); upf(db2, b2, D);
  if (nln_out) { upf(dng, nln_g, D); upf(dnb, nln_b, D); dn.ensure((size_t)M * D * 2); }
  std::vector<uint16_t> h((size_t)H * D);
  pack_mlp_w1(w1, h.data());
  dw1.ensure(h.size() * 2); TTR_HIP_CHECK(hipMemcpy(dw1.p, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  pack_mlp_w2(w2, H, h.data());
  dw2.ensure(h.size() * 2); TTR_HIP_CHECK(hipMemcpy(dw2.p, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  dout.ensure((size_t)M * D * 4);
  MlpParams q{};
  q.x = dx.as<float>(); q.x_out = dout.as<float>(); q.M = M; q.ln_g = dg.as<float>(); q.ln_b = db.as<float>(); q.ln_eps = eps;
  q.w1p = dw1.as<bf16>(); q.b1 = db1.as<float>(); q.w2p = dw2.as<bf16>(); q.b2 = db2.as<float>();
  if (nln_out) { q.nln_g = dng.as<float>(); q.nln_b = dnb.as<float>(); q.nln_eps = eps; q.nln_out = dn.as<bf16>(); }
  DevBuf datt, dwp, dbp;
  if (att) {
    std::vector<uint16_t> ha((size_t)M * D), hw((size_t)D * D);
    for (size_t i = 0; i < ha.size(); ++i) ha[i] = f32_to_bf16_rne(att[i]);
    pack_mlp_w2(wp, D, hw.data());
    datt.ensure(ha.size() * 2); TTR_HIP_CHECK(hipMemcpy(datt.p, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
    dwp.ensure(hw.size() * 2); TTR_HIP_CHECK(hipMemcpy(dwp.p, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    upf(dbp, bp, D);
    q.att = datt.as<bf16>(); q.wpp = dwp.as<bf16>(); q.bp = dbp.as<float>();
  }
  launch_mlp_fused(q, E.stream);
  TTR_HIP_CHECK(hipMemcpyAsync(x_out, dout.p, (size_t)M * D * 4, hipMemcpyDeviceToHost, E.stream));
  TTR_HIP_CHECK(hipStreamSynchronize(E.stream));
  if (nln_out) {
    std::vector<uint16_t> hb((size_t)M * D);
    TTR_HIP_CHECK(hipMemcpy(hb.data(), dn.p, hb.size() * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < hb.size(); ++i) { const uint32_t u = (uint32_t)hb[i] << 16; memcpy(&nln_out[i], &u, 4); }
  }
  return 0;
  TTR_GUARD_END(-1)
}

int ttr_dbg_qkv_attn(ttr_engine* e, const float* x, int N, const float* w, const float* b, float* out) {
  TTR_GUARD_BEGIN
  Engine& E = *e->e;
  EngineScope lk(E);
  if (E.prec == kSplit) {   // the fused launch of the default precision: x -> pairs, weight rows head-major, output triples joined here
    const size_t nx = (size_t)N * 128 * 384;
    DevBuf dx, dxp, dout;
    Linear L;
    dx.ensure(nx * 4); TTR_HIP_CHECK(hipMemcpy(dx.p, x, nx * 4, hipMemcpyHostToDevice));
    dxp.ensure(nx * 4);
    launch_split_planes(dx.as<float>(), 384, dxp.p, (int64_t)N * 128, 384, 0, E.stream, 2);
    std::vector<float> wp((size_t)1152 * 384), bp(1152);
    for (int n = 0; n < 1152; ++n) { const int src = Engine::qkv_tile_row(n); memcpy(&wp[(size_t)n * 384], &w[(size_t)src * 384], 384 * 4); bp[n] = b[src]; }
    E.upload_linear(L, wp.data(), 1152, 384, bp.data(), 1152, 384, nullptr, false);
    dout.ensure(nx * 6);
    if (E.tn.sp_tiled_w) E.tile_planes(L);
    launch_qkv_attn_split(dxp.p, L.ws.p, L.b.as<float>(), L.inv_scale, dout.p, N, E.stream, E.tn.sp_tiled_w ? L.wst.p : nullptr);
    std::vector<_Float16> h(nx * 3);
    TTR_HIP_CHECK(hipMemcpyAsync(h.data(), dout.p, nx * 6, hipMemcpyDeviceToHost, E.stream));
    TTR_HIP_CHECK(hipStreamSynchronize(E.stream));
    for (size_t m = 0; m < (size_t)N * 128; ++m)
      for (int c = 0; c < 384; ++c) {
        const _Float16* row = h.data() + m * 1152;
        out[m * 384 + c] = (float)((double)(float)row[c] + ((double)(float)row[384 + c] + (double)(float)row[768 + c]) / 2048.0);
      }
    return 0;
  }
  if (E.prec != kBF16) throw std::runtime_error("ttr_dbg_qkv_attn: bf16 and f16x4 engines only");
  const size_t nx = (size_t)N * 128 * 384, nw = (size_t)1152 * 384;
  DevBuf dx, dw, db, dout;
  std::vector<uint16_t> h(std::max(nx, nw));
  for (size_t i = 0; i < nx; ++i) h[i] = f32_to_bf16_rne(x[i]);
  dx.ensure(nx * 2); TTR_HIP_CHECK(hipMemcpy(dx.p, h.data(), nx * 2, hipMemcpyHostToDevice));
  for (size_t i = 0; i < nw; ++i) h[i] = f32_to_bf16_rne(w[i]);
  dw.ensure(nw * 2); TTR_HIP_CHECK(hipMemcpy(dw.p, h.data(), nw * 2, hipMemcpyHostToDevice));
  db.ensure(1152 * 4); TTR_HIP_CHECK(hipMemcpy(db.p, b, 1152 * 4, hipMemcpyHostToDevice));
  dout.ensure(nx * 2);
  launch_qkv_attn(dx.as<bf16>(), dw.as<bf16>(), db.as<float>(), dout.as<bf16>(), N, E.stream);
  TTR_HIP_CHECK(hipStreamSynchronize(E.stream));
  TTR_HIP_CHECK(hipMemcpy(h.data(), dout.p, nx * 2, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < nx; ++i) { const uint32_t u = (uint32_t)h[i] << 16; memcpy(&out[i], &u, 4); }
  return 0;
  TTR_GUARD_END(-1)
}

int ttr_dbg_attn_enc(ttr_engine* e, const float* qkv, int N, float* out) {
  TTR_GUARD_BEGIN
  Engine& E = *e->e;
  EngineScope lk(E);
  const size_t nin = (size_t)N * 128 * 1152, nout = (size_t)N * 128 * 384;
  DevBuf din, dout;
  din.ensure(nin * E.es); dout.ensure(nout * E.es);
  if (E.prec == kBF16) {
    std::vector<uint16_t> h(nin);
    for (size_t i = 0; i < nin; ++i) h[i] = f32_to_bf16_rne(qkv[i]);
    TTR_HIP_CHECK(hipMemcpy(din.p, h.data(), nin * 2, hipMemcpyHostToDevice));
  } else TTR_HIP_CHECK(hipMemcpy(din.p, qkv, nin * 4, hipMemcpyHostToDevice));
  launch_attn_enc(E.prec, din.p, dout.p, N, E.stream);
  TTR_HIP_CHECK(hipStreamSynchronize(E.stream));
  if (E.prec == kBF16) {
    std::vector<uint16_t> h(nout);
    TTR_HIP_CHECK(hipMemcpy(h.data(), dout.p, nout * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < nout; ++i) { const uint32_t u = (uint32_t)h[i] << 16; memcpy(&out[i], &u, 4); }
  } else TTR_HIP_CHECK(hipMemcpy(out, dout.p, nout * 4, hipMemcpyDeviceToHost));
  return 0;
  TTR_GUARD_END(-1)
}

int ttr_dbg_cross_attn(ttr_engine* e, const float* q, const float* kvmem, int N, int R, float* out) {
  TTR_GUARD_BEGIN
  Engine& E = *e->e;
  EngineScope lk(E);
  if (E.prec == kBF16) throw std::runtime_error("ttr_dbg_cross_attn: split-operand / fp32 engines");
  if (N <= 0 || R <= 0) throw std::runtime_error("ttr_dbg_cross_attn: N, R >= 1");
  const size_t nq = (size_t)N * R * 384, nkv = (size_t)N * 128 * 768;
  DevBuf dq, dkv, dout;
  dq.ensure(nq * 4); dkv.ensure(nkv * 4); dout.ensure(nq * 6);
  TTR_HIP_CHECK(hipMemcpy(dq.p, q, nq * 4, hipMemcpyHostToDevice));
  TTR_HIP_CHECK(hipMemcpy(dkv.p, kvmem, nkv * 4, hipMemcpyHostToDevice));
  launch_dec_cross_attn(kF32, dq.p, dkv.p, dout.p, N, R, E.stream, nullptr, 0, nullptr, 0, 3);   // exact triples [N * R][3][384]
  TTR_HIP_CHECK(hipStreamSynchronize(E.stream));
  std::vector<_Float16> h(nq * 3);
  TTR_HIP_CHECK(hipMemcpy(h.data(), dout.p, nq * 6, hipMemcpyDeviceToHost));
  for (size_t r = 0; r < (size_t)N * R; ++r)
    for (int c = 0; c < 384; ++c)
      out[r * 384 + c] = (float)h[r * 1152 + c] + ((float)h[r * 1152 + 384 + c] + (float)h[r * 1152 + 768 + c]) * (1.f / 2048.f);
  return 0;
  TTR_GUARD_END(-1)
}

int ttr_dbg_conv_pool(ttr_engine* e, const float* in0, int C0, int B, int H, int W, int ks, const float* wgt, const float* bias, int Cout, int act,
                      int pool_relu, float* out_full, float* out_pool) {
  TTR_GUARD_BEGIN
  Engine& E = *e->e;
  EngineScope lk(E);
  if (E.prec != kBF16) throw std::runtime_error("ttr_dbg_conv_pool: bf16 engines only (the fused pool lives in gemm2)");
  const size_t M = (size_t)B * H * W, Mp = M / 4;
  const int K = ks * ks * C0;
  DevBuf d0, dfull, dpool;
  Linear L;
  std::vector<uint16_t> hbuf(M * C0);
  for (size_t i = 0; i < hbuf.size(); ++i) hbuf[i] = f32_to_bf16_rne(in0[i]);
  d0.ensure(hbuf.size() * 2);
  TTR_HIP_CHECK(hipMemcpy(d0.p, hbuf.data(), hbuf.size() * 2, hipMemcpyHostToDevice));
  E.upload_linear(L, wgt, Cout, K, bias, Cout, K, nullptr, false);
  dfull.ensure(M * Cout * 2); dpool.ensure(Mp * Cout * 2);
  ConvParams p{};
  p.in0 = d0.p; p.C0 = C0; p.B = B; p.H = H; p.W = W; p.ks = ks; p.dil = 1; p.wgt = L.w.p; p.bias = bias ? L.b.as<float>() : nullptr;
  p.out = out_full ? dfull.p : nullptr; p.out_ld = Cout; p.out_pool = dpool.p; p.pool_relu = pool_relu;
  p.Cout = Cout; p.M = (int)M; p.act = act;
  launch_igemm(E.prec, p, E.stream);
  auto down = [&](const DevBuf& d, size_t n, float* dst) {
    std::vector<uint16_t> h(n);
    TTR_HIP_CHECK(hipMemcpyAsync(h.data(), d.p, n * 2, hipMemcpyDeviceToHost, E.stream));
    TTR_HIP_CHECK(hipStreamSynchronize(E.stream));
    for (size_t i = 0; i < n; ++i) { uint32_t u = (uint32_t)h[i] << 16; memcpy(&dst[i], &u, 4); }
  };
  if (out_full) down(dfull, M * Cout, out_full);
  down(dpool, Mp * Cout, out_pool);
  return 0;
  TTR_GUARD_END(-1)
}

void ttr_set_gemm_config(int cfg) { set_gemm_config(cfg); }

void ttr_set_decoder_mode(int mode) { g_tuning_default.decoder_mode = mode; }

void ttr_last_host_us(ttr_engine* e, float out[8]) { for (int i = 0; i < 8; ++i) out[i] = e ? e->e->host_us[i] : 0.f; }

int ttr_dbg_dec_stamps_ext(unsigned long long* out, int n) { return g_dec_dbg && n >= 0 && n <= 4096 && hipMemcpy(out, g_dec_dbg, (size_t)n * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1; }
int ttr_dbg_dec_stamps(unsigned long long* out) { return g_dec_dbg && hipMemcpy(out, g_dec_dbg, 26 * 16 * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1; }

// process-wide: the kernel files' variant switches and diagnostics; engine-level keys set the default of engines created afterwards
int ttr_set_tuning(const char* key, int value) {
  const std::string k = key ? key : "";
  if (g_tuning_default.set(k, value)) return 0;
  if (k == "gemm_config") set_gemm_config(value);
  else if (k == "self_refine") set_dec_self_refine(value);
  else if (k == "cross_mfma") set_dec_cross_mfma(value);
  else if (k == "cross_crop") set_dec_cross_crop(value);
  else if (k == "cross_split") set_dec_cross_split(value);
  else if (k == "cross_rows_hsplit") set_dec_cross_rows_hsplit(value);
  else if (k == "mlp_store_nt") set_mlp_store_nt(value);
  else if (k == "mlp_stagger") set_mlp_stagger(value);
  else if (k == "c3s_wgs") set_conv3s_wgs_per_cu(value);
  else if (k == "c3_c32") set_conv3p_c32_tile(value);
  else if (k == "c3_narrow64") set_conv3p_narrow_bn64(value);
  else if (k == "upsample_block") set_upsample_block(value);
  else if (k == "mlp_ablate") set_mlp_ablate(value);
  else if (k == "attn_impl") set_attn_impl(value);
  else if (k == "ws_dbg_flags") set_gemm_ws_dbg_flags(value);
  else if (k == "ws_lean") set_gemm_ws_lean(value);
  else if (k == "store_policy") set_store_policy(value);
  else if (k == "head_persistent") set_conv3p_head_persistent(value);
  else if (k == "up_resident") set_gemm2_up_resident(value);
  else if (k == "up_2d") set_gemm2_up_2d(value);
  else if (k == "c3h_wgs_per_cu") set_conv3h_wgs_per_cu(value);
  else if (k == "g2_x_ring3") set_gemm2_x_ring3(value);
  else if (k == "g2_split_reuse") set_gemm2_split_reuse(value);
  else if (k == "g2_split_cfg") set_gemm2_split_cfg(value);
  else if (k == "g2_split_few") set_gemm2_split_few(value);
  else if (k == "g2_split_dbg") set_gemm2_split_dbg(value);
  else if (k == "g2_split_wreg") set_gemm2_split_wreg(value);
  else if (k == "g2_split_stream") set_gemm2_split_stream(value);
  else if (k == "g2_split_stream4") set_gemm2_split_stream4(value);
  else if (k == "gsp_sched") set_gemm_sp_sched(value);
  else if (k == "gsp_few") set_gemm_sp_few(value);
  else if (k == "gsp_epi") set_gemm_sp_epi(value);
  else if (k == "gsp_ks3") set_gemm_sp_ks3(value);
  else if (k == "gsp_stagger") set_gemm_sp_stagger(value);
  else if (k == "gsp_dbg") set_gemm_sp_dbg(value);
  else if (k == "gsp_stag") set_gemm_sp_stag(value);
  else if (k == "gsp_stagger_groups") set_gemm_sp_stagger_groups(value);
  else if (k == "qkv_attn_dbg") set_qkv_attn_dbg(value);
  else if (k == "qkv_attn4") set_qkv_attn4(value);
  else if (k == "skx_ln_max_rows") set_gemm_skx_ln_max_rows(value);
  else if (k == "c3_xs1_max_cin") set_conv3p_single_stage_max_cin(value);
  else if (k == "c3_force_bn128") set_conv3p_force_bn128(value);
  else if (k == "c3_c64_waves") set_conv3p_c64_waves(value);
  else if (k == "c3_c128_waves") set_conv3p_c128_waves(value);
  else if (k == "c3_narrow_wide") set_conv3p_narrow_wide(value);
  else if (k == "c3_narrow_frac") set_conv3p_narrow_frac(value);
  else if (k == "c3_narrowest_frac") set_conv3p_narrowest_frac(value);
  else if (k == "c3_deep_w") set_conv3p_deep_w(value);
  else if (k == "c3_deep_w64") set_conv3p_deep_w64(value);
  else if (k == "c3_first_persistent") set_conv3p_first_persistent(value);
  else if (k == "sk_max_rows") set_skinny_max_rows(value);
  else if (k == "ws_min_rows") set_gemm_ws_min_rows(value);
  else if (k == "dec_stamps") {   // value != 0: allocate the stamp buffer; read it back with ttr_dev_download via ttr_dbg_dec_stamps
    if (value && !g_dec_dbg) { void* d = nullptr; if (hipMalloc(&d, 4096 * 8) != hipSuccess) return -1; (void)hipMemset(d, 0, 4096 * 8); g_dec_dbg = (unsigned long long*)d; }
    if (!value) g_dec_dbg = nullptr;
    set_gemm_ws_stamps(value == 2 ? g_dec_dbg : nullptr);
    set_conv3p_stamps(value == 4 ? g_dec_dbg : nullptr);    // 4: ... or conv3p_first2 stamps
    set_mlp_stamps(value == 3 ? g_dec_dbg : nullptr);
    set_qkv_attn_stamps(value == 5 ? g_dec_dbg : nullptr);
    set_qkv_attn4_stamps(value == 5 ? g_dec_dbg : nullptr);   // 5: ... or the fused qkv + attention launch's
    set_conv3h_stamps(value == 6 ? g_dec_dbg : nullptr);     // 6: ... or the persistent head kernel's (conv3h.hip)
    // 3: ... or mlp_fused stamps   // 2: the same buffer takes gemm_ws stamps instead
  }
  else return -1;
  return 0;
}

// per engine (under the engine's lock: a batch in flight on another thread keeps the selection it started with)
int ttr_engine_set_tuning(ttr_engine* e, const char* key, int value) {
  TTR_GUARD_BEGIN
  if (!e) return -1;
  const std::string k = key ? key : "";
  {
    std::lock_guard<std::mutex> lk(e->e->mu);
    if (e->e->tn.set(k, value)) return 0;
  }
  return ttr_set_tuning(key, value);   // not an engine key: the process-wide diagnostics setter (documented in tuatara_hip.h)
  TTR_GUARD_END(-1)
}

int ttr_bench_conv(ttr_engine* e, int B, int H, int W, int C0, int C1, int ks, int dil, int Cout, int act, int f32_resid, int iters, float* avg_us) {
  TTR_GUARD_BEGIN
  Engine& E = *e->e;
  EngineScope lk(E);
  const size_t M = (size_t)B * H * W;
  const int K = ks * ks * (C0 + C1);
  DevBuf d0, d1, dw, db, dout, dres;
  d0.ensure(M * C0 * E.es); launch_fill_random(E.prec, d0.p, M * C0, 1u, 1.0f, E.stream);
  if (C1) { d1.ensure(M * C1 * E.es); launch_fill_random(E.prec, d1.p, M * C1, 2u, 1.0f, E.stream); }
  dw.ensure((size_t)Cout * K * E.es); launch_fill_random(E.prec, dw.p, (size_t)Cout * K, 3u, 1.0f / std::sqrt((float)K), E.stream);
  db.ensure((size_t)Cout * 4); launch_fill_random(kF32, db.p, Cout, 4u, 1.0f, E.stream);
  ConvParams p{};
  p.in0 = d0.p; p.C0 = C0; p.in1 = C1 ? d1.p : nullptr; p.C1 = C1;
  p.B = B; p.H = H; p.W = W; p.ks = ks; p.dil = dil; p.wgt = dw.p; p.bias = db.as<float>();
  p.Cout = Cout; p.M = (int)M; p.act = act;
  if (f32_resid) {   // the PARSeq residual-stream form: f32 in, f32 out
    dres.ensure(M * Cout * 4); launch_fill_random(kF32, dres.p, M * Cout, 5u, 1.0f, E.stream);
    p.out_f32 = dres.as<float>(); p.out_f32_ld = Cout; p.resid = dres.as<float>(); p.resid_ld = Cout;
  } else {
    dout.ensure(M * Cout * E.es); p.out = dout.p; p.out_ld = Cout;
  }
  for (int i = 0; i < 2; ++i) launch_igemm(E.prec, p, E.stream);
  hipEvent_t a, b;
  TTR_HIP_CHECK(hipEventCreate(&a)); TTR_HIP_CHECK(hipEventCreate(&b));
  TTR_HIP_CHECK(hipEventRecord(a, E.stream));
  for (int i = 0; i < iters; ++i) launch_igemm(E.prec, p, E.stream);
  TTR_HIP_CHECK(hipEventRecord(b, E.stream));
  TTR_HIP_CHECK(hipStreamSynchronize(E.stream));
  float ms = 0.f;
  TTR_HIP_CHECK(hipEventElapsedTime(&ms, a, b));
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  *avg_us = ms * 1e3f / iters;
  return 0;
  TTR_GUARD_END(-1)
}

int ttr_dbg_min_area_rect(const float* xy, int n, float* r5) {
  TTR_GUARD_BEGIN
  std::vector<Pt2f> p(n);
  for (int i = 0; i < n; ++i) p[i] = Pt2f{xy[2 * i], xy[2 * i + 1]};
  RRect r = min_area_rect(p.data(), n);
  r5[0] = r.cx; r5[1] = r.cy; r5[2] = r.w; r5[3] = r.h; r5[4] = r.angle;
  return 0;
  TTR_GUARD_END(-1)
}

int ttr_dbg_component_rect(int area, int x0, int y0, int x1, int y1, const int32_t* rows, int H, int W, float* r5) {
  TTR_GUARD_BEGIN
  Component c{0, area, x0, y0, x1, y1, rows};
  RRect r;
  if (!component_to_rect(c, H, W, &r)) return 0;
  r5[0] = r.cx; r5[1] = r.cy; r5[2] = r.w; r5[3] = r.h; r5[4] = r.angle;
  return 1;
  TTR_GUARD_END(-1)
}

int ttr_dbg_box_geometry(const float* r5, float ratio, float* adj5, int32_t* xywh, float* bbox4) {
  TTR_GUARD_BEGIN
  RRect r{r5[0], r5[1], r5[2], r5[3], r5[4]};
  RRect b = adjust_coordinates(r, 1.f / ratio, 1.f / ratio);
  adj5[0] = b.cx; adj5[1] = b.cy; adj5[2] = b.w; adj5[3] = b.h; adj5[4] = b.angle;
  int q[4];
  bounding_rect(b, q);
  for (int i = 0; i < 4; ++i) xywh[i] = q[i];
  tesseract_bbox(b, bbox4);
  return 0;
  TTR_GUARD_END(-1)
}

}  // extern "C"
