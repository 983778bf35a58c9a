// CRAFT forward (the TorchScript detector run at tuatara.cpp:376): 27 convolutions, BN folded, pools / upsamples / concats fused or virtual.
#include "engine.h"

namespace ttr {

void Engine::conv(const char* name, const void* in0, int C0, const void* in1, int C1, int relu0, int B, int H, int W, void* out, int act,
          float* out_f32, void* out_relu, void* out_pool, int pool_relu) {
  const Linear& L = craft.at(name);
  ConvParams p{};
  p.in0 = in0; p.C0 = C0; p.in1 = in1; p.C1 = C1; p.relu0 = relu0; p.relu1 = 0;
  p.B = B; p.H = H; p.W = W;
  const int Ct = C0 + C1;
  p.ks = (L.k == Ct) ? 1 : 3;
  if (L.k != p.ks * p.ks * Ct) throw std::runtime_error(std::string("conv shape mismatch at ") + name);
  p.dil = std::string(name) == "slice5.1" ? 6 : 1;
  p.wgt = L.w.p; p.bias = L.b.as<float>();
  p.out = out; p.out_ld = L.cout; p.out_f32 = out_f32; p.out_f32_ld = L.cout; p.out_relu = out_relu; p.out_pool = out_pool; p.pool_relu = pool_relu;
  p.Cout = L.cout; p.M = B * H * W; p.act = act;
  double flops = 0;   // algorithmic: 2 * M * Cout * K of the *unpadded* layer
  for (const auto& c : craft_convs()) if (std::string(c.name) == name) flops = 2.0 * p.M * c.cout * c.ks * c.ks * c.cin;
  igemm(p, flops, prec == kSplit ? "igemm_kernel<f32> (CRAFT head 1x1)" : "CRAFT convolutions (igemm / gemm2 / conv3p)");
}

void Engine::craft_forward(const uint8_t* d_canvas, int B, int H, int W, float* d_heat) {
  if (H % 32 || W % 32) throw std::runtime_error("CRAFT canvas must be a multiple of 32");
  if (prec == kSplit) return craft_forward_split(d_canvas, B, H, W, d_heat);   // (the split_gemm / split_planes knobs act on PARSeq only)
  prof_stage = 0;
  const size_t M0 = (size_t)B * H * W, M1 = M0 / 4, M2 = M1 / 4, M3 = M2 / 4, M4 = M3 / 4;
  const int H1 = H / 2, W1 = W / 2, H2 = H / 4, W2 = W / 4, H3 = H / 8, W3 = W / 8, H4 = H / 16, W4 = W / 16;
  size_t k = 0;
  auto buf = [&](size_t rows, int C) -> void* { return ws(k++, rows * C * es).p; };
  // 2x2 max-pools: fused into the producing conv's epilogue in bf16 mode (gemm2 / conv3p), a separate kernel in f32 mode
  const bool fp = prec == kBF16 && gemm_config() >= 0;
  const bool first_fused = fp && tn.fuse_first && H % 8 == 0 && W % 32 == 0;   // conv1_1 computed inside conv1_2's loader (conv3p FIRST)
  void* a0 = buf(prec == kBF16 ? 0 : M0, 32);
  void* c11 = buf(first_fused ? 0 : M0, 64);
  void* c12 = buf(fp ? 0 : M0, 64); void* p1 = buf(M1, 64);
  if (first_fused) {
    const Linear& L0 = craft.at("slice1.0"); const Linear& L = craft.at("slice1.3");
    ConvParams p{};
    p.in0 = d_canvas; p.C0 = 64; p.B = B; p.H = H; p.W = W; p.ks = 3; p.dil = 1;
    p.pre_wgt = L0.w.p; p.pre_bias = L0.b.as<float>();
    p.wgt = L.w.p; p.bias = L.b.as<float>(); p.out_ld = 64; p.out_pool = p1; p.Cout = 64; p.M = (int)M0; p.act = kActRelu;
    timed("conv3p_first2s (conv1_1 + conv1_2 + pool)", 2.0 * M0 * 64 * (27 + 576), 2.0 * M0 * 64 * (27 + 576), [&] { launch_conv3p(p, stream); });
  } else {
    if (prec == kBF16) {   // conv1_1 straight from the u8 canvas
      const Linear& L = craft.at("slice1.0");
      timed("conv1_direct", 2.0 * M0 * 64 * 27, 2.0 * M0 * 64 * 27, [&] { launch_conv1_direct(d_canvas, L.w.p, L.b.as<float>(), c11, B, H, W, stream); });
    } else {
      prof_break(), launch_im2col_l1(prec, d_canvas, a0, B, H, W, stream);
      conv("slice1.0", a0, 32, nullptr, 0, 0, 1, 1, (int)M0, c11, kActRelu);
    }
    if (fp) conv("slice1.3", c11, 64, nullptr, 0, 0, B, H, W, nullptr, kActRelu, nullptr, nullptr, p1);
    else { conv("slice1.3", c11, 64, nullptr, 0, 0, B, H, W, c12, kActRelu); prof_break(), launch_maxpool2x2(prec, c12, p1, B, H, W, 64, 0, stream); }
  }
  void* c21 = buf(M1, 128); conv("slice1.7", p1, 64, nullptr, 0, 0, B, H1, W1, c21, kActRelu);
  void* c22 = buf(M1, 128); void* p2 = buf(M2, 128);                                                   // relu2_2 skip (pre-ReLU)
  if (fp) conv("slice1.10", c21, 128, nullptr, 0, 0, B, H1, W1, c22, kActNone, nullptr, nullptr, p2, 1);
  else { conv("slice1.10", c21, 128, nullptr, 0, 0, B, H1, W1, c22, kActNone); prof_break(), launch_maxpool2x2(prec, c22, p2, B, H1, W1, 128, 1, stream); }
  void* c31 = buf(M2, 256); conv("slice2.14", p2, 128, nullptr, 0, 0, B, H2, W2, c31, kActRelu);
  void* c32 = buf(M2, 256); void* c32r = buf(M2, 256);
  conv("slice2.17", c31, 256, nullptr, 0, 0, B, H2, W2, c32, kActNone, nullptr, c32r);                // relu3_2 skip (pre-ReLU) + its ReLU
  void* c33 = buf(fp ? 0 : M2, 256); void* p3 = buf(M3, 256);
  if (fp) conv("slice3.20", c32r, 256, nullptr, 0, 0, B, H2, W2, nullptr, kActRelu, nullptr, nullptr, p3);
  else { conv("slice3.20", c32r, 256, nullptr, 0, 0, B, H2, W2, c33, kActRelu); prof_break(), launch_maxpool2x2(prec, c33, p3, B, H2, W2, 256, 0, stream); }
  void* c41 = buf(M3, 512); conv("slice3.24", p3, 256, nullptr, 0, 0, B, H3, W3, c41, kActRelu);
  void* c42 = buf(M3, 512); void* c42r = buf(M3, 512);
  conv("slice3.27", c41, 512, nullptr, 0, 0, B, H3, W3, c42, kActNone, nullptr, c42r);                // relu4_3 skip + its ReLU
  void* c43 = buf(fp ? 0 : M3, 512); void* p4 = buf(M4, 512);
  if (fp) conv("slice4.30", c42r, 512, nullptr, 0, 0, B, H3, W3, nullptr, kActRelu, nullptr, nullptr, p4);
  else { conv("slice4.30", c42r, 512, nullptr, 0, 0, B, H3, W3, c43, kActRelu); prof_break(), launch_maxpool2x2(prec, c43, p4, B, H3, W3, 512, 0, stream); }
  void* c51 = buf(M4, 512); conv("slice4.34", p4, 512, nullptr, 0, 0, B, H4, W4, c51, kActRelu);
  void* c52 = buf(M4, 512); conv("slice4.37", c51, 512, nullptr, 0, 0, B, H4, W4, c52, kActNone);   // relu5_3 skip
  void* mp = buf(M4, 512);  prof_break(), launch_maxpool3x3s1(prec, c52, mp, B, H4, W4, 512, stream);
  void* c6 = buf(M4, 1024); conv("slice5.1", mp, 512, nullptr, 0, 0, B, H4, W4, c6, kActNone);
  void* fc7 = buf(M4, 1024); conv("slice5.2", c6, 1024, nullptr, 0, 0, B, H4, W4, fc7, kActNone);
  void* u1a = buf(M4, 512); conv("upconv1.0", fc7, 1024, c52, 512, 0, B, H4, W4, u1a, kActRelu);
  void* u1b = buf(M4, 256); conv("upconv1.3", u1a, 512, nullptr, 0, 0, B, H4, W4, u1b, kActRelu);
  void* up1 = buf(M3, 256); prof_break(), launch_upsample2x(prec, u1b, up1, B, H4, W4, 256, stream);
  void* u2a = buf(M3, 256); conv("upconv2.0", up1, 256, c42, 512, 0, B, H3, W3, u2a, kActRelu);
  void* u2b = buf(M3, 128); conv("upconv2.3", u2a, 256, nullptr, 0, 0, B, H3, W3, u2b, kActRelu);
  void* up2 = buf(M2, 128); prof_break(), launch_upsample2x(prec, u2b, up2, B, H3, W3, 128, stream);
  void* u3a = buf(M2, 128); conv("upconv3.0", up2, 128, c32, 256, 0, B, H2, W2, u3a, kActRelu);
  void* u3b = buf(M2, 64);  conv("upconv3.3", u3a, 128, nullptr, 0, 0, B, H2, W2, u3b, kActRelu);
  void* up3 = buf(M1, 64);  prof_break(), launch_upsample2x(prec, u3b, up3, B, H2, W2, 64, stream);
  void* u4a = buf(M1, 64);  conv("upconv4.0", up3, 64, c22, 128, 0, B, H1, W1, u4a, kActRelu);
  void* u4b = buf(M1, 32);  conv("upconv4.3", u4a, 64, nullptr, 0, 0, B, H1, W1, u4b, kActRelu);
  void* h0 = buf(M1, 32); void* h2 = buf(M1, 32);
  if (fp && H1 % 8 == 0 && W1 % 32 == 0 && M1 * 64 < ((size_t)1 << 31)) {
    // 32-channel head: conv3s.hip (patch-resident 3x3; conv_cls.4 + .6 + .8 as one kernel writing the f32 heat map)
    auto head = [&](const char* name, const void* in, void* out, bool tail) {
      const Linear& L = craft.at(name);
      Conv3sParams q{};
      q.in = (const bf16*)in; q.wgt = L.w.as<bf16>(); q.bias = L.b.as<float>(); q.out = (bf16*)out; q.B = B; q.H = H1; q.W = W1;
      double flops = 2.0 * M1 * 32 * 288;
      if (tail) {
        const Linear& L6 = craft.at("conv_cls.6"); const Linear& L8 = craft.at("conv_cls.8");
        q.w6 = L6.w.as<bf16>(); q.b6 = L6.b.as<float>(); q.w8 = L8.w.as<bf16>(); q.b8 = L8.b.as<float>(); q.heat = d_heat; q.out = nullptr;
        flops = 2.0 * M1 * (16 * 288 + 16 * 16 + 2 * 16);
      }
      timed("conv3s (32-channel head)", flops, flops, [&] { launch_conv3s(q, stream); });
    };
    head("conv_cls.0", u4b, h0, false);
    head("conv_cls.2", h0, h2, false);
    head("conv_cls.4", h2, nullptr, true);
  } else {
    conv("conv_cls.0", u4b, 32, nullptr, 0, 0, B, H1, W1, h0, kActRelu);
    conv("conv_cls.2", h0, 32, nullptr, 0, 0, B, H1, W1, h2, kActRelu);
    void* h4 = buf(M1, 32);   conv("conv_cls.4", h2, 32, nullptr, 0, 0, B, H1, W1, h4, kActRelu);   // 16 real + 16 zero channels
    void* h6 = buf(M1, 32);   conv("conv_cls.6", h4, 32, nullptr, 0, 0, B, H1, W1, h6, kActRelu);
    conv("conv_cls.8", h6, 32, nullptr, 0, 0, B, H1, W1, nullptr, kActNone, d_heat);
  }
  prof_break();
}

void Engine::sconv(const char* name, const void* in0, int C0, const void* in1, int C1, int B, int H, int W, void* out, int act,
           void* out_relu, void* out_pool, int pool_relu, int out_planes, int out_ld, bool packed, float* tail_heat) {
  const int np = tn.craft_products == 4 ? 4 : 3;             // products per value: 3 = activation pairs (default), 4 = exact triples
  if (out_planes < 0) out_planes = np - 1;
  range_tag(std::string("craft.") + name);
  const Linear& L = craft.at(name);
  if (packed) {   // a 32-channel layer on packed pairs: in0 = pixel rows [x0 (32) | x1 (32)] (a pairs tensor of 32 channels), weights Linear::wsp
    if (np != 3 || C0 != 64 || C1 || !L.wsp.p || L.k != 9 * 64) throw std::runtime_error(std::string("packed split conv: wrong layer ") + name);
    ConvParams p{};
    p.in0 = in0; p.C0 = 64; p.B = B; p.H = H; p.W = W; p.ks = 3; p.dil = 1;
    p.wgt = L.wsp.p; p.bias = L.b.as<float>(); p.split = 2; p.out_scale = L.inv_scale; p.out_planes = out_planes;
    p.out = out; p.out_ld = out_ld ? out_ld : L.cout; p.Cout = L.cout; p.M = B * H * W; p.act = act;
    double flops = 0, tail_flops = 0;
    for (const auto& c : craft_convs()) if (std::string(c.name) == name) flops = 2.0 * p.M * c.cout * c.ks * c.ks * c.cin;
    if (tail_heat) {   // conv_cls.4 with conv_cls.6 + conv_cls.8 as its epilogue: nothing but the heat map leaves the kernel
      p.out = nullptr; p.tail_heat = tail_heat;
      p.tail_w6 = head_tail.w6.p; p.tail_w8 = head_tail.w8.p; p.tail_b6 = head_tail.b6.as<float>(); p.tail_b8 = head_tail.b8.as<float>();
      p.tail_s6 = head_tail.s6; p.tail_s8 = head_tail.s8;
      tail_flops = 2.0 * p.M * (16 * 16 + 2 * 16);
    }
    if (const char* e = conv3p_check(p)) throw std::runtime_error(std::string(name) + ": " + e);
    const char* pk = tail_heat ? "conv3p_kernel<32,NP=2> + conv_cls.6 + conv_cls.8 (head tail)" : "conv3p_kernel<32,NP=2> (packed pairs, 32 input channels)";
    const std::string per_layer = std::string(name) + " | " + pk;
    timed(profiling == 2 ? per_layer.c_str() : pk, flops + tail_flops,
          2.0 * p.M * L.cout * 9 * 64 * 2 + (tail_heat ? 2.0 * p.M * 16 * 32 * 6 : 0.0), [&] { launch_conv3p(p, stream); },
          (double)p.M * 128.0 + (tail_heat ? (double)p.M * 8.0 : (double)p.M * (out_planes ? 128.0 : 32.0 * 4)) + (double)L.cout * 9 * 64 * 2 * 2);   // pixel rows [x0 | x1] of 128 B in and out (or the heat map)
    return;
  }
  ConvParams p{};
  p.in0 = in0; p.C0 = C0; p.in1 = in1; p.C1 = C1; p.B = B; p.H = H; p.W = W;
  const int Ct = C0 + C1;
  p.ks = (L.k == Ct) ? 1 : 3;
  if (L.k != p.ks * p.ks * Ct || !L.ws.p) throw std::runtime_error(std::string("split conv shape mismatch at ") + name);
  p.dil = std::string(name) == "slice5.1" ? 6 : 1;
  p.wgt = L.ws.p; p.bias = L.b.as<float>(); p.split = np; p.out_scale = L.inv_scale; p.out_planes = out_planes;
  p.out = out; p.out_ld = out_ld ? out_ld : L.cout; p.out_relu = out_relu; p.out_pool = out_pool; p.pool_relu = pool_relu;
  p.Cout = L.cout; p.M = B * H * W; p.act = act;
  double flops = 0;
  for (const auto& c : craft_convs()) if (std::string(c.name) == name) flops = 2.0 * p.M * c.cout * c.ks * c.ks * c.cin;
  const bool fused_first = sconv_canvas != nullptr;
  if (fused_first) {   // conv1_1 inside this layer's kernel (conv3p.hip: FIRST on pairs): in0 is the u8 canvas
    const Linear& L0 = craft.at("slice1.0");
    p.in0 = sconv_canvas; p.pre_wgt = L0.ws.p; p.pre_bias = L0.b.as<float>(); p.pre_scale = L0.inv_scale;
    range_tag("craft.slice1.0"); p.pre_range_tag = range_ctx().tag;   // a trip inside the fused layer keeps its own name
    range_tag(std::string("craft.") + name);
    if (const char* e = conv3p_check(p)) throw std::runtime_error(std::string(name) + " (fused first layer): " + e);
    flops += 2.0 * p.M * 64 * 27;
    const char* fk = "conv3p_kernel<64,NP=3> + conv1_1 (fused first layer)";
    const std::string fl = std::string("slice1.0 + ") + name + " | " + fk;
    timed(profiling == 2 ? fl.c_str() : fk, flops, flops * np, [&] { launch_conv3p(p, stream); },
          (double)p.M * 3.0 + (out ? (double)p.M * p.Cout * 2.0 * out_planes : 0.0) + (out_pool ? (double)p.M / 4 * p.Cout * 2.0 * out_planes : 0.0) + (double)p.Cout * L.k * 6.0);
    return;
  }
  const bool c3 = tn.split_conv3p && p.Cout >= 32 && conv3p_check(p) == nullptr;
  if (!c3) { if (const char* e = gemm2_check(p)) throw std::runtime_error(std::string(name) + ": " + e); }
  // kinds by kernel: the patch-stationary 3x3 kernel by its tile width (conv3p.hip picks it), everything else on gemm2's split loop
  const int bn = c3 ? conv3p_split_bn(p) : 0;
  const char* kind = !c3 ? (np == 3 ? "gemm2_kernel<SP,NP=3> (CRAFT 1x1 / dilated)" : "gemm2_kernel<SP,NP=4> (CRAFT 1x1 / dilated)")
                   : bn == 128 ? (np == 3 ? "conv3p_kernel<128,NP=3>" : "conv3p_kernel<128,NP=4>")
                   : bn == 64 ? (np == 3 ? "conv3p_kernel<64,NP=3>" : "conv3p_kernel<64,NP=4>") : (np == 3 ? "conv3p_kernel<32,NP=3>" : "conv3p_kernel<32,NP=4>");
  const double ob = out_planes ? 2.0 * out_planes : 4.0;   // bytes per output value (planes of f16, or fp32)
  const double bytes = (double)p.M * Ct * 2.0 * (np - 1) + (out ? (double)p.M * p.Cout * ob : 0.0) + (out_relu ? (double)p.M * p.Cout * ob : 0.0) +
                       (out_pool ? (double)p.M / 4 * p.Cout * ob : 0.0) + (double)p.Cout * L.k * 2.0 * 3;
  // profiling == 2 (every launch bracketed: bench.py's per-layer pass behind the timed region): the kind is "layer | kernel", so that every layer states its own roof
  const std::string per_layer = std::string(name) + " | " + kind;
  timed(profiling == 2 ? per_layer.c_str() : kind, flops, flops * np, [&] { if (c3) launch_conv3p(p, stream); else launch_gemm2(p, 0, stream); }, bytes);
}

// upconvN.0 = ReLU(W . cat(upsample2x(y), skip) + b), a 1x1 convolution (inside CRAFT's TorchScript module run at tuatara.cpp:376).  The bilinear upsample is a fixed
// linear map over pixels, the 1x1 convolution one over channels: they commute.  z = W_up . y at the LOW resolution (a quarter of the rows, fp32 out, no bias),
// then ReLU(W_skip . skip + b + upsample2x(z)) with the four-tap interpolation in the second launch's epilogue (ConvParams::up_z).  The upsampled tensor - as wide
// as y and four times as long - is neither written nor read; the arithmetic differs from the two-source form at fp32 rounding level only (sums in another order).
void Engine::upconv_commuted(const char* name, const void* y_lo, int C0, const void* skip, int C1, int B, int H, int W, float* z, void* out) {
  range_tag(std::string("craft.") + name);
  const Linear& La = craft.at(std::string(name) + ".up");
  const Linear& Lb = craft.at(std::string(name) + ".skip");
  if (La.k != C0 || Lb.k != C1 || !La.ws.p || !Lb.ws.p || La.cout != Lb.cout || (H & 1) || (W & 1)) throw std::runtime_error(std::string("commuted up-convolution: shape mismatch at ") + name);
  const int np = tn.craft_products == 4 ? 4 : 3, Cout = La.cout;
  const int64_t Mhi = (int64_t)B * H * W, Mlo = Mhi / 4;
  ConvParams a{};
  a.in0 = y_lo; a.C0 = C0; a.B = B; a.H = H / 2; a.W = W / 2; a.ks = 1; a.dil = 1;
  a.wgt = La.ws.p; a.bias = La.b.as<float>(); a.split = np; a.out_scale = La.inv_scale; a.out_planes = 0;
  a.out = z; a.out_ld = Cout; a.Cout = Cout; a.M = (int)Mlo; a.act = kActNone;
  if (const char* e = gemm2_check(a)) throw std::runtime_error(std::string(name) + " (low-resolution half): " + e);
  ConvParams b{};
  b.in0 = skip; b.C0 = C1; b.B = B; b.H = H; b.W = W; b.ks = 1; b.dil = 1;
  b.wgt = Lb.ws.p; b.bias = Lb.b.as<float>(); b.split = np; b.out_scale = Lb.inv_scale; b.out_planes = np - 1;
  b.out = out; b.out_ld = Cout; b.Cout = Cout; b.M = (int)Mhi; b.act = kActRelu;
  b.up_z = z; b.up_ld = Cout;
  if (const char* e = gemm2_check(b)) throw std::runtime_error(std::string(name) + " (skip half): " + e);
  const char* kind = np == 3 ? "gemm2_kernel<SP,NP=3> (CRAFT 1x1 / dilated)" : "gemm2_kernel<SP,NP=4> (CRAFT 1x1 / dilated)";
  // algorithmic flops: the layer's own (SURVEY.md section 8(d) counts the convolution as the reference runs it); executed: what the two launches multiply
  const std::string la = std::string(name) + " (W_up . y at the low resolution) | " + kind, lb = std::string(name) + " (skip half + upsample(z) in the epilogue) | " + kind;
  timed(profiling == 2 ? la.c_str() : kind, 2.0 * Mhi * Cout * C0, 2.0 * Mlo * Cout * C0 * np, [&] { launch_gemm2(a, 0, stream); }, (double)Mlo * C0 * 2.0 * (np - 1) + (double)Mlo * Cout * 4.0 + (double)Cout * C0 * 6.0);
  timed(profiling == 2 ? lb.c_str() : kind, 2.0 * Mhi * Cout * C1, 2.0 * Mhi * Cout * C1 * np, [&] { launch_gemm2(b, 0, stream); },
        (double)Mhi * C1 * 2.0 * (np - 1) + (double)Mlo * Cout * 4.0 + (double)Mhi * Cout * 2.0 * (np - 1) + (double)Cout * C1 * 6.0);
}

void Engine::craft_forward_split(const uint8_t* d_canvas, int B, int H, int W, float* d_heat) {
  prof_stage = 0;
  const size_t M0 = (size_t)B * H * W, M1 = M0 / 4, M2 = M1 / 4, M3 = M2 / 4, M4 = M3 / 4;
  const int H1 = H / 2, W1 = W / 2, H2 = H / 4, W2 = W / 4, H3 = H / 8, W3 = W / 8, H4 = H / 16, W4 = W / 16;
  size_t k = 0;
  const int npl = tn.craft_products == 4 ? 3 : 2;                                       // planes per value
  const int layout = npl * 2 + (tn.head_packed && npl == 2 ? 1 : 0);   // (plane count and the head tensors' row form: both move the zero padding channels)
  if (layout != craft_ws_npl) {   // another plane count: the zero padding channels of the head tensors sit elsewhere - start from fresh buffers
    TTR_HIP_CHECK(hipStreamSynchronize(stream)); TTR_HIP_CHECK(hipStreamSynchronize(lane_stream));
    craft_ws_set[0].clear(); craft_ws_set[1].clear();
    craft_ws_npl = layout;
  }
  auto pbuf = [&](size_t rows, int C) -> void* { return ws(k++, rows * C * 2 * npl).p; };   // planes
  auto fbuf = [&](size_t rows, int C) -> void* { return ws(k++, rows * C * 4).p; };   // fp32
  // conv1_1: a launch of its own (conv1_split_kernel: the 64-channel tensor at full resolution goes out and comes back), or - pairs, 8 x 32 patches, the split
  // 3x3 tiles on - evaluated inside conv1_2's kernel on each halo patch: same arithmetic, same bits, the tensor never exists (tuning key first_fused)
  const bool fuse_first = tn.first_fused && npl == 2 && tn.split_conv3p && H % 8 == 0 && W % 32 == 0 && M0 * 3 < ((size_t)1 << 31);
  void* c11 = fuse_first ? nullptr : pbuf(M0, 64);
  if (!fuse_first) {
    const Linear& L0 = craft.at("slice1.0");
    range_tag("craft.slice1.0");
    timed(profiling == 2 ? "slice1.0 | conv1_split_kernel" : "conv1_split_kernel", 2.0 * M0 * 64 * 27, 2.0 * M0 * 64 * 27 * (npl + 1), [&] { launch_conv1_split(d_canvas, L0.ws.p, L0.b.as<float>(), L0.inv_scale, c11, B, H, W, stream, npl); },
          (double)M0 * 3.0 + (double)M0 * 64 * 2.0 * npl);
  } else k++;   // (the workspace slot stays reserved: the slots behind it keep their sizes whichever way this page goes)
  void* p1 = pbuf(M1, 64);
  {
    struct CanvasScope { Engine& E; ~CanvasScope() { E.sconv_canvas = nullptr; } } scope{*this};
    sconv_canvas = fuse_first ? d_canvas : nullptr;
    sconv("slice1.3", c11, 64, nullptr, 0, B, H, W, nullptr, kActRelu, nullptr, p1, 0);
  }
  void* c21 = pbuf(M1, 128); sconv("slice1.7", p1, 64, nullptr, 0, B, H1, W1, c21, kActRelu);
  void* c22 = pbuf(M1, 128); void* p2 = pbuf(M2, 128);
  sconv("slice1.10", c21, 128, nullptr, 0, B, H1, W1, c22, kActNone, nullptr, p2, 1);                    // relu2_2 skip (pre-ReLU) + pooled ReLU
  void* c31 = pbuf(M2, 256); sconv("slice2.14", p2, 128, nullptr, 0, B, H2, W2, c31, kActRelu);
  void* c32 = pbuf(M2, 256); void* c32r = pbuf(M2, 256);
  sconv("slice2.17", c31, 256, nullptr, 0, B, H2, W2, c32, kActNone, c32r);                               // relu3_2 skip + its ReLU
  void* p3 = pbuf(M3, 256);  sconv("slice3.20", c32r, 256, nullptr, 0, B, H2, W2, nullptr, kActRelu, nullptr, p3, 0);
  if (lane_go_pending) { prof_break(); TTR_HIP_CHECK(hipEventRecord(lane_go, stream)); lane_go_pending = false; }   // (half a group in: the second lane may start)
  void* c41 = pbuf(M3, 512); sconv("slice3.24", p3, 256, nullptr, 0, B, H3, W3, c41, kActRelu);
  void* c42 = pbuf(M3, 512); void* c42r = pbuf(M3, 512);
  sconv("slice3.27", c41, 512, nullptr, 0, B, H3, W3, c42, kActNone, c42r);                               // relu4_3 skip + its ReLU
  void* p4 = pbuf(M4, 512);  sconv("slice4.30", c42r, 512, nullptr, 0, B, H3, W3, nullptr, kActRelu, nullptr, p4, 0);
  void* c51 = pbuf(M4, 512); sconv("slice4.34", p4, 512, nullptr, 0, B, H4, W4, c51, kActRelu);
  void* c52 = pbuf(M4, 512); sconv("slice4.37", c51, 512, nullptr, 0, B, H4, W4, c52, kActNone);          // relu5_3 skip
  void* mp = pbuf(M4, 512);  prof_break(), launch_maxpool3x3s1_planes(c52, mp, B, H4, W4, 512, stream, npl);
  void* c6 = pbuf(M4, 1024); sconv("slice5.1", mp, 512, nullptr, 0, B, H4, W4, c6, kActNone);
  void* fc7 = pbuf(M4, 1024); sconv("slice5.2", c6, 1024, nullptr, 0, B, H4, W4, fc7, kActNone);
  void* u1a = pbuf(M4, 512); sconv("upconv1.0", fc7, 1024, c52, 512, B, H4, W4, u1a, kActRelu);
  void* u1b = pbuf(M4, 256); sconv("upconv1.3", u1a, 512, nullptr, 0, B, H4, W4, u1b, kActRelu);
  // upconvN.0 over cat(upsample(y), skip): the upsample kernel + a two-source 1x1, or (tn.up_commute) the commuted form that never writes the upsampled tensor
  auto upconv = [&](const char* name, const void* y_lo, int C0, const void* skip, int C1, int Cout, size_t Mhi, int Hh, int Wh) -> void* {
    if (tn.up_commute && craft.count(std::string(name) + ".up") && Hh % 2 == 0 && Wh % 2 == 0) {
      float* z = (float*)fbuf(Mhi / 4, Cout);
      void* o = pbuf(Mhi, Cout);
      upconv_commuted(name, y_lo, C0, skip, C1, B, Hh, Wh, z, o);
      return o;
    }
    void* up = pbuf(Mhi, C0); prof_break(), launch_upsample2x_planes(y_lo, up, B, Hh / 2, Wh / 2, C0, stream, npl);
    void* o = pbuf(Mhi, Cout); sconv(name, up, C0, skip, C1, B, Hh, Wh, o, kActRelu);
    return o;
  };
  void* u2a = upconv("upconv2.0", u1b, 256, c42, 512, 256, M3, H3, W3);
  void* u2b = pbuf(M3, 128); sconv("upconv2.3", u2a, 256, nullptr, 0, B, H3, W3, u2b, kActRelu);
  void* u3a = upconv("upconv3.0", u2b, 128, c32, 256, 128, M2, H2, W2);
  void* u3b = pbuf(M2, 64);  sconv("upconv3.3", u3a, 128, nullptr, 0, B, H2, W2, u3b, kActRelu);
  void* u4a = upconv("upconv4.0", u3b, 64, c22, 128, 64, M1, H1, W1);
  // 32-channel head: the 3x3 layers on the f16 kernels over planes with 32 zero channels behind the 32 real ones (row = 64 channels);
  // the two 1x1 layers (16 -> 16 -> 2) on the fp32 MFMA kernel
  // (head_packed, pairs only: the 32-channel tensors as 128-byte pixel rows [x0 | x1], their consumers on packed pairs - conv3p.hip, NP = 2)
  void *u4b, *h0, *h2, *h4 = nullptr;
  if (tn.head_packed && npl == 2 && H1 % 8 == 0 && W1 % 32 == 0) {
    u4b = pbuf(M1, 32); sconv("upconv4.3", u4a, 64, nullptr, 0, B, H1, W1, u4b, kActRelu);
    h0 = pbuf(M1, 32);  sconv("conv_cls.0", u4b, 64, nullptr, 0, B, H1, W1, h0, kActRelu, nullptr, nullptr, 0, -1, 0, true);
    h2 = pbuf(M1, 32);  sconv("conv_cls.2", h0, 64, nullptr, 0, B, H1, W1, h2, kActRelu, nullptr, nullptr, 0, -1, 0, true);
    if (tn.head_tail && head_tail.w6.p) {
      sconv("conv_cls.4", h2, 64, nullptr, 0, B, H1, W1, nullptr, kActRelu, nullptr, nullptr, 0, /*out_planes=*/0, 0, true, d_heat);   // + conv_cls.6 + conv_cls.8
      prof_break();
      return;
    }
    h4 = fbuf(M1, 32);  sconv("conv_cls.4", h2, 64, nullptr, 0, B, H1, W1, h4, kActRelu, nullptr, nullptr, 0, /*out_planes=*/0, 0, true);   // fp32, 16 real + 16 zero channels
  } else {
  auto zbuf = [&](size_t rows) -> void* { return ws(k++, rows * 64 * 2 * npl, true).p; };
  u4b = zbuf(M1); sconv("upconv4.3", u4a, 64, nullptr, 0, B, H1, W1, u4b, kActRelu, nullptr, nullptr, 0, -1, 64);
  h0 = zbuf(M1);  sconv("conv_cls.0", u4b, 64, nullptr, 0, B, H1, W1, h0, kActRelu, nullptr, nullptr, 0, -1, 64);
  h2 = zbuf(M1);  sconv("conv_cls.2", h0, 64, nullptr, 0, B, H1, W1, h2, kActRelu, nullptr, nullptr, 0, -1, 64);
  h4 = fbuf(M1, 32); sconv("conv_cls.4", h2, 64, nullptr, 0, B, H1, W1, h4, kActRelu, nullptr, nullptr, 0, /*out_planes=*/0);   // fp32, 16 real + 16 zero channels
  }
  {   // the two 1x1 head layers stay on the fp32 MFMA kernel (restored also when a launch throws)
    struct Restore { int& v; int keep; ~Restore() { v = keep; } } restore{tn.split_gemm, tn.split_gemm};
    tn.split_gemm = 0;
    void* h6 = fbuf(M1, 32);
    conv("conv_cls.6", h4, 32, nullptr, 0, 0, B, H1, W1, h6, kActRelu);
    conv("conv_cls.8", h6, 32, nullptr, 0, 0, B, H1, W1, nullptr, kActNone, d_heat);
  }
  prof_break();
}

}  // namespace ttr
