// Engine construction: weights to the device (BN-folded fp32, bf16 images, f16 weight planes), workspaces, the per-launch profile.
#include "engine.h"
#include <map>
#include <mutex>

namespace ttr {

void hip_fail(const char* what, hipError_t e, const char* file, int line) {
  char buf[512];
  snprintf(buf, sizeof buf, "HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
  throw std::runtime_error(buf);
}

Tuning g_tuning_default;

RangeCtx& range_ctx() { static thread_local RangeCtx c; return c; }

// ConvParams::tile_ctr: zeroed words per (device, stream), allocated on first use and never freed (a handful of streams per process).  Launches on one
// stream run in order and each leaves the words zero, so a stream's launches share them; two streams never do.
unsigned* tile_counters(hipStream_t s) {
  static std::mutex m;
  static std::map<std::pair<int, hipStream_t>, unsigned*> table;
  int dev = 0;
  TTR_HIP_CHECK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(m);
  unsigned*& d = table[{dev, s}];
  if (!d) {
    void* q = nullptr;
    TTR_HIP_CHECK(hipMalloc(&q, (16 + 2048) * 4));   // (+ one word per CU: qkv_attn4.hip's matrix-phase token)
    TTR_HIP_CHECK(hipMemset(q, 0, (16 + 2048) * 4));   // (synchronous: done before any launch that follows)
    d = (unsigned*)q;
  }
  return d;
}

void Engine::range_tag(const std::string& layer) {
  if (prec != kSplit) return;
  auto it = range_ids.find(layer);
  if (it == range_ids.end()) { range_names.push_back(layer); it = range_ids.emplace(layer, (unsigned)range_names.size()).first; }
  range_ctx().tag = it->second;
}

void Engine::range_fetch(int w) {
  if (!range_flag_ptr()) return;
  TTR_HIP_CHECK(hipMemcpyAsync(h_range.as<unsigned>() + w, range_word.as<unsigned>() + w, 4, hipMemcpyDeviceToHost, stream));
  TTR_HIP_CHECK(hipMemsetAsync(range_word.as<unsigned>() + w, 0, 4, stream));   // cleared by the stream that wrote it: the word's next user starts clean
}

void Engine::range_verify(int w, const char* where) {
  if (!range_flag_ptr()) return;
  unsigned& v = h_range.as<unsigned>()[w];
  if (!v) return;
  const std::string layer = v <= range_names.size() ? range_names[v - 1] : std::string("(untagged kernel)");
  v = 0;
  const std::string msg = std::string("f16x4 range guard (") + where + "): an activation of layer '" + layer + "' reached |x| >= 65504 (or an infinity): the split-operand "
                          "precision cannot represent it and has saturated - these weights need TTR_PREC_F32 (tuatara_amd/csrc/split.h)";
  if (tn.range_guard == 2) { std::cerr << "warning: " << msg << std::endl; return; }
  throw std::runtime_error(msg);
}

// mlp_fused.hip's weight operands are stored as the LDS images the kernel multiplies from (see the layout notes there)
void pack_mlp_w1(const float* w1, uint16_t* out) {               // w1 [1536][384] -> [48 chunks][3 segments][32 rows][16 positions][8]
  for (int c = 0; c < 48; ++c)
    for (int s = 0; s < 3; ++s)
      for (int R = 0; R < 32; ++R) {
        const int n = ((R & 15) >> 2) * 8 + (R >> 4) * 4 + (R & 3);          // hidden unit of LDS row R
        for (int cp = 0; cp < 16; ++cp) {
          const int ch = s * 16 + (cp ^ (R & 15));                            // source 16-byte chunk at position cp
          uint16_t* d = out + ((((size_t)c * 3 + s) * 32 + R) * 16 + cp) * 8;
          for (int e = 0; e < 8; ++e) d[e] = f32_to_bf16_rne(w1[(size_t)(32 * c + n) * 384 + ch * 8 + e]);
        }
      }
}
void pack_mlp_w2(const float* w, int K, uint16_t* out) {          // w [384][K] -> [K/32 chunks][384 rows][4 positions][8]
  for (int c = 0; c < K / 32; ++c)
    for (int R = 0; R < 384; ++R) {
      const int ot = R >> 4, q = R & 15;
      const int oc = (ot >> 1) * 32 + (q >> 2) * 8 + (ot & 1) * 4 + (q & 3);   // output channel of LDS row R
      for (int cp = 0; cp < 4; ++cp) {
        const int g = cp ^ ((R >> 1) & 3);
        uint16_t* d = out + (((size_t)c * 384 + R) * 4 + cp) * 8;
        for (int e = 0; e < 8; ++e) d[e] = f32_to_bf16_rne(w[(size_t)oc * K + 32 * c + g * 8 + e]);
      }
    }
}

void Engine::igemm(const ConvParams& p, double true_flops, const char* kind) {
  if (prec == kSplit && split_gemm(p, true_flops, kind)) return;
  timed(kind, true_flops, true_flops, [&] { launch_igemm(prec, p, stream); });
}

bool Engine::split_gemm(const ConvParams& p, double true_flops, const char* kind) {
  auto it = split_by_w.find(p.wgt);
  if (it == split_by_w.end() || !tn.split_gemm || p.relu0 || p.relu1 || p.ln_in || p.pre_wgt) return false;
  const Linear& L = *it->second;
  ConvParams q = p;
  q.split = 4; q.wgt = L.ws.p; q.out_scale = L.inv_scale; q.out_planes = 0; q.store_policy = 0;
  split_in[0].ensure((size_t)p.M * p.C0 * 6);
  q.in0 = split_in[0].p;
  if (p.C1) { split_in[1].ensure((size_t)p.M * p.C1 * 6); q.in1 = split_in[1].p; }
  const bool c3 = tn.split_conv3p && p.Cout >= 32 && conv3p_check(q) == nullptr;
  if (!c3 && gemm2_check(q) != nullptr) return false;
  prof_break_if_craft();
  launch_split_planes((const float*)p.in0, p.C0, split_in[0].p, p.M, p.C0, 0, stream, 3, p.skip, p.skip_n);
  if (p.C1) launch_split_planes((const float*)p.in1, p.C1, split_in[1].p, p.M, p.C1, 0, stream, 3, p.skip, p.skip_n);
  timed(kind, true_flops, true_flops * 4, [&] { if (c3) launch_conv3p(q, stream); else launch_gemm2(q, 0, stream); });
  return true;
}

void Engine::prof_collect() {
  if (seg_open) return;                        // (never between the two events of a run)
  size_t done = 0;
  for (; done < prof_recs.size(); ++done) {
    if (hipEventQuery(prof_pool[2 * done + 1]) != hipSuccess) { (void)hipGetLastError(); break; }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, prof_pool[2 * done], prof_pool[2 * done + 1]) == hipSuccess) {
      const ProfRec& r = prof_recs[done];
      prof_ms[r.stage] += ms; prof_flops[r.stage] += r.exec; prof_launches[r.stage] += r.launches;
      ProfKind& k = prof_kinds[r.kind];
      k.ms += ms; k.alg += r.alg; k.exec += r.exec; k.launches += r.launches; k.bytes += r.bytes;
    }
  }
  if (done == 0) return;
  std::rotate(prof_pool.begin(), prof_pool.begin() + 2 * done, prof_pool.begin() + 2 * prof_recs.size());
  prof_recs.erase(prof_recs.begin(), prof_recs.begin() + done);
}

void Engine::upload_linear(Linear& L, const float* w, int cout, int k, const float* bias, int cout_pad, int k_pad,
                   const std::vector<int>* kmap, bool own) {
  // kmap: for each padded k index the source k index or -1
  std::vector<float> wp((size_t)cout_pad * k_pad, 0.f);
  for (int o = 0; o < cout; ++o)
    for (int kk = 0; kk < k_pad; ++kk) {
      int src = kmap ? (*kmap)[kk] : (kk < k ? kk : -1);
      if (src >= 0) wp[(size_t)o * k_pad + kk] = w[(size_t)o * k + src];
    }
  L.cout = cout_pad; L.k = k_pad;
  L.w.ensure(wp.size() * es);
  if (prec == kBF16) {
    std::vector<uint16_t> h(wp.size());
    for (size_t i = 0; i < wp.size(); ++i) h[i] = f32_to_bf16_rne(wp[i]);
    TTR_HIP_CHECK(hipMemcpy(L.w.p, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  } else {
    TTR_HIP_CHECK(hipMemcpy(L.w.p, wp.data(), wp.size() * 4, hipMemcpyHostToDevice));
  }
  if (prec == kSplit && k_pad % 32 == 0 && cout_pad % 8 == 0) {   // the f16x4 GEMM's weight planes (the fp32 copy stays for the layers it cannot run)
    float mx = 0.f;
    for (float v : wp) mx = std::max(mx, std::fabs(v));
    int e = 0;
    if (mx > 0.f) { (void)std::frexp(mx, &e); e = 14 - e; }            // max |w| 2^e in [2^13, 2^14)
    e = std::max(-24, std::min(40, e));
    const float S = std::ldexp(1.f, e);
    std::vector<_Float16> h((size_t)cout_pad * 3 * k_pad);
    for (int o = 0; o < cout_pad; ++o)
      for (int kk = 0; kk < k_pad; ++kk) {
        const float v = wp[(size_t)o * k_pad + kk] * S;               // exact
        const _Float16 w0 = (_Float16)v;
        const _Float16 w1 = (_Float16)(v - (float)w0);                 // exact difference, then rounded: 22+ bits in the pair
        _Float16* row = h.data() + (size_t)o * 3 * k_pad;
        row[kk] = w0;
        row[k_pad + kk] = (_Float16)((float)w0 * (1.f / 2048.f));
        row[2 * k_pad + kk] = drop_w1 ? (_Float16)0.f : w1;
      }
    L.ws.ensure(h.size() * 2);
    TTR_HIP_CHECK(hipMemcpy(L.ws.p, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    L.inv_scale = std::ldexp(1.f, -e);
    if (own) split_by_w[L.w.p] = &L;       // (only the engine's own layers: a debug entry's local Linear dies with its call)
  }
  std::vector<float> bp(cout_pad, 0.f);
  if (bias) memcpy(bp.data(), bias, sizeof(float) * cout);
  L.b.ensure(bp.size() * 4);
  TTR_HIP_CHECK(hipMemcpy(L.b.p, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
}

// The weight planes of a linear layer once more, as the 1-KiB pieces gemm_sp.hip's loader fetches (ConvParams::wgt_tiled): piece P = 8 rows x 64 halves
// of one plane and one 64-deep k step, contiguous; [P][plane][k / 64][8][64] with P counting eight-row groups IN THE ORDER OF THE LDS IMAGE - within
// every 32 output channels, image row lr (0 .. 31) holds channel ((lr & 15) >> 2) * 8 + ((lr >> 4) & 1) * 4 + (lr & 3), the permutation that leaves a
// lane of the MFMA result 8 consecutive channels (gemm_sp.hip: tile_offsets).  Channels past cout are zero rows.
void Engine::tile_planes(Linear& L) {
  if (!L.ws.p || L.k % 64 != 0 || L.cout % 8 != 0) return;
  const int K = L.k, k64 = K / 64, groups = (L.cout + 31) / 32;
  std::vector<uint16_t> h((size_t)L.cout * 3 * K), t((size_t)groups * 32 * 3 * K, 0);
  TTR_HIP_CHECK(hipMemcpy(h.data(), L.ws.p, h.size() * 2, hipMemcpyDeviceToHost));
  for (int P = 0; P < groups * 4; ++P)
    for (int j = 0; j < 8; ++j) {
      const int lr = 8 * (P & 3) + j;
      const int n = 32 * (P >> 2) + ((lr & 15) >> 2) * 8 + ((lr >> 4) & 1) * 4 + (lr & 3);
      if (n >= L.cout) continue;
      for (int pl = 0; pl < 3; ++pl)
        for (int kb = 0; kb < k64; ++kb)
          memcpy(&t[((((size_t)P * 3 + pl) * k64 + kb) * 8 + j) * 64], &h[((size_t)n * 3 + pl) * K + (size_t)kb * 64], 128);
    }
  L.wst.ensure(t.size() * 2);
  TTR_HIP_CHECK(hipMemcpy(L.wst.p, t.data(), t.size() * 2, hipMemcpyHostToDevice));
}

void Engine::upload_f32(DevBuf& d, const float* p, size_t n) {
  d.ensure(n * 4);
  TTR_HIP_CHECK(hipMemcpy(d.p, p, n * 4, hipMemcpyHostToDevice));
}

void Engine::load_craft(const std::string& dir) {
  WeightFile wf(dir + "/craft.ttrw");
  for (const auto& c : craft_convs()) {
    const int taps = c.ks * c.ks;
    const auto& w = wf.get(std::string(c.name) + ".w", (size_t)c.cout * taps * c.cin);
    const auto& b = wf.get(std::string(c.name) + ".b", (size_t)c.cout);
    Linear& L = craft[c.name];
    if (std::string(c.name) == "slice1.0") {
      upload_linear(L, w.data.data(), c.cout, 27, b.data.data(), c.cout, 32);  // im2col K 27 -> 32
      continue;
    }
    // channel padding to multiples of 32 (only the 16-channel head tensors need it)
    int cin_pad = (c.cin + 31) / 32 * 32;
    // split-operand engines: the three 3x3 layers of the 32-channel head run on the f16 kernels too, which want Cin % 64: their inputs
    // carry 32 zero channels (the layers are thin: 3 % of CRAFT's flops; on the fp32 MFMA kernel they took 5 % of the time)
    const std::string nm(c.name);
    const bool head3 = prec == kSplit && (nm == "conv_cls.0" || nm == "conv_cls.2" || nm == "conv_cls.4");
    if (head3) cin_pad = 64;
    int cout_pad = c.cout;
    if (std::string(c.name) == "conv_cls.4" || std::string(c.name) == "conv_cls.6") cout_pad = 32;  // feeds a padded-Cin layer
    std::vector<int> kmap((size_t)taps * cin_pad, -1);
    for (int t = 0; t < taps; ++t)
      for (int ci = 0; ci < c.cin; ++ci) kmap[(size_t)t * cin_pad + ci] = t * c.cin + ci;
    // EXPERIMENT (TUATARA_CRAFT_PRODUCTS=2, DESIGN_APPENDIX.md "CRAFT on two products"): the layers of >= 64 input channels with their weights' low parts
    // dropped (w1 = 0: x0 w0 + x1 w0 / 2^11 only) - the arithmetic a two-MFMA-per-product CRAFT would have, on the three-product kernels
    struct DropScope { bool& f; ~DropScope() { f = false; } } drop_scope{drop_w1};
    { const char* ev = getenv("TUATARA_CRAFT_PRODUCTS"); drop_w1 = ev && std::string(ev) == "2" && c.cin >= 64 && nm.rfind("conv_cls", 0) != 0; }
    upload_linear(L, w.data.data(), c.cout, taps * c.cin, b.data.data(), cout_pad, taps * cin_pad, &kmap);
    if (prec == kSplit && nm == "conv_cls.8") load_head_tail(wf);
    if (prec == kSplit && (nm == "upconv2.0" || nm == "upconv3.0" || nm == "upconv4.0")) {
      // the layer's two column blocks as linears of their own (tn.up_commute): [0, C0) multiplies the upsampled tensor, [C0, cin) the skip tensor (+ the bias)
      const int c0 = nm == "upconv2.0" ? 256 : nm == "upconv3.0" ? 128 : 64, c1 = c.cin - c0;
      std::vector<int> ka(c0), kb(c1);
      for (int i = 0; i < c0; ++i) ka[i] = i;
      for (int i = 0; i < c1; ++i) kb[i] = c0 + i;
      upload_linear(craft[nm + ".up"], w.data.data(), c.cout, c.cin, nullptr, c.cout, c0, &ka);
      upload_linear(craft[nm + ".skip"], w.data.data(), c.cout, c.cin, b.data.data(), c.cout, c1, &kb);
    }
    if (head3 && c.cin == 32) {   // the packed pairs form of the same layer (Linear::wsp), same scale S
      const float S = 1.f / L.inv_scale;
      const int kp = taps * 64;
      std::vector<_Float16> h((size_t)cout_pad * 3 * kp, (_Float16)0.f);
      for (int o = 0; o < c.cout; ++o)
        for (int t = 0; t < taps; ++t)
          for (int ci = 0; ci < 32; ++ci) {
            const float v = w.data[(size_t)o * taps * 32 + t * 32 + ci] * S;   // exact (S a power of two)
            const _Float16 w0 = (_Float16)v;
            const _Float16 w1 = (_Float16)(v - (float)w0);
            _Float16* row = h.data() + (size_t)o * 3 * kp;
            row[t * 64 + ci] = w0;
            row[t * 64 + 32 + ci] = (_Float16)((float)w0 * (1.f / 2048.f));
            row[2 * kp + t * 64 + ci] = w1;
          }
      L.wsp.ensure(h.size() * 2);
      TTR_HIP_CHECK(hipMemcpy(L.wsp.p, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    }
  }
}

// conv_cls.6 (16 -> 16) and conv_cls.8 (16 -> 2) as weight pairs in the fragment rows conv3p.hip's fused head tail multiplies from
void Engine::load_head_tail(WeightFile& wf) {
  auto planes = [](const std::vector<float>& w, int rows, int cols, bool slot_order, std::vector<_Float16>& out, float& inv_scale) {
    float mx = 0.f;
    for (float v : w) mx = std::max(mx, std::fabs(v));
    int e = 0;
    if (mx > 0.f) { (void)std::frexp(mx, &e); e = 14 - e; }
    e = std::max(-24, std::min(40, e));
    const float S = std::ldexp(1.f, e);
    inv_scale = std::ldexp(1.f, -e);
    out.assign((size_t)16 * 64, (_Float16)0.f);                       // [16 rows][w0 (32 k) | w1 (32 k)]
    for (int r = 0; r < rows; ++r)
      for (int c = 0; c < cols; ++c) {
        const float v = w[(size_t)r * cols + c] * S;
        const _Float16 w0 = (_Float16)v, w1 = (_Float16)(v - (float)w0);
        const int k = slot_order ? 8 * (c >> 2) + (c & 3) : c;         // conv_cls.8: channel 4 g + e sits in k slot 8 g + e
        out[(size_t)r * 64 + k] = w0; out[(size_t)r * 64 + 32 + k] = w1;
      }
  };
  const auto& w6 = wf.get("conv_cls.6.w", 16 * 16).data; const auto& b6 = wf.get("conv_cls.6.b", 16).data;
  const auto& w8 = wf.get("conv_cls.8.w", 2 * 16).data;  const auto& b8 = wf.get("conv_cls.8.b", 2).data;
  std::vector<_Float16> h;
  planes(w6, 16, 16, false, h, head_tail.s6);
  head_tail.w6.ensure(h.size() * 2); TTR_HIP_CHECK(hipMemcpy(head_tail.w6.p, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  planes(w8, 2, 16, true, h, head_tail.s8);
  head_tail.w8.ensure(h.size() * 2); TTR_HIP_CHECK(hipMemcpy(head_tail.w8.p, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  upload_f32(head_tail.b6, b6.data(), 16);
  upload_f32(head_tail.b8, b8.data(), 2);
}

void Engine::load_parseq(const std::string& dir) {
  WeightFile wf(dir + "/parseq.ttrw");
  auto lin = [&](const std::string& key, const std::string& wname, const std::string& bname, int cout, int k, int row0 = 0, int rows_total = -1) {
    if (rows_total < 0) rows_total = cout;
    const auto& w = wf.get(wname, (size_t)rows_total * k);
    const auto& b = wf.get(bname, (size_t)rows_total);
    upload_linear(pq[key], w.data.data() + (size_t)row0 * k, cout, k, b.data.data() + row0, cout, k);
  };
  auto vec = [&](const std::string& name, size_t n) { upload_f32(pqf[name], wf.get(name, n).data.data(), n); };
  const int E = 384;
  {   // patch embedding: K = 96 (4 x 8 x 3); the bf16 engine pads it to 128 so that the linear runs on gemm2 (K % 64) instead of the
    // first-generation igemm (104 -> ~60 us at 1280 crops); the pad columns are zero in the patches and in the weights
    const auto& w = wf.get("encoder.patch_embed.proj.weight", (size_t)E * 96);
    const auto& b = wf.get("encoder.patch_embed.proj.bias", (size_t)E);
    upload_linear(pq["patch"], w.data.data(), E, 96, b.data.data(), E, prec != kF32 ? 128 : 96);
  }
  vec("encoder.pos_embed", 128 * E);
  for (int i = 0; i < 12; ++i) {
    std::string p = "encoder.blocks." + std::to_string(i) + ".";
    vec(p + "norm1.weight", E); vec(p + "norm1.bias", E); vec(p + "norm2.weight", E); vec(p + "norm2.bias", E);
    lin(p + "qkv", p + "attn.qkv.weight", p + "attn.qkv.bias", 3 * E, E);
    if (prec == kSplit) {   // the fused qkv + attention launch (gemm_sp.hip, attention epilogue) wants the rows tile by tile: head-major, see qkv_tile_row
      const auto& w = wf.get(p + "attn.qkv.weight", (size_t)3 * E * E).data;
      const auto& b = wf.get(p + "attn.qkv.bias", (size_t)3 * E).data;
      std::vector<float> wp((size_t)3 * E * E), bp((size_t)3 * E);
      for (int n = 0; n < 3 * E; ++n) {
        const int src = qkv_tile_row(n);
        memcpy(&wp[(size_t)n * E], &w[(size_t)src * E], sizeof(float) * E);
        bp[n] = b[src];
      }
      upload_linear(pq[p + "qkv_hm"], wp.data(), 3 * E, E, bp.data(), 3 * E, E);
    }
    lin(p + "proj", p + "attn.proj.weight", p + "attn.proj.bias", E, E);
    lin(p + "fc1", p + "mlp.fc1.weight", p + "mlp.fc1.bias", 4 * E, E);
    lin(p + "fc2", p + "mlp.fc2.weight", p + "mlp.fc2.bias", E, 4 * E);
    if (prec == kBF16) {   // mlp_fused.hip's operands as LDS images
      std::vector<uint16_t> h((size_t)E * 4 * E);
      pack_mlp_w1(wf.get(p + "mlp.fc1.weight", (size_t)4 * E * E).data.data(), h.data());
      fc1_packed[i].ensure(h.size() * 2);
      TTR_HIP_CHECK(hipMemcpy(fc1_packed[i].p, h.data(), h.size() * 2, hipMemcpyHostToDevice));
      pack_mlp_w2(wf.get(p + "mlp.fc2.weight", (size_t)E * 4 * E).data.data(), 4 * E, h.data());
      fc2_packed[i].ensure(h.size() * 2);
      TTR_HIP_CHECK(hipMemcpy(fc2_packed[i].p, h.data(), h.size() * 2, hipMemcpyHostToDevice));
      std::vector<uint16_t> hp((size_t)E * E);
      pack_mlp_w2(wf.get(p + "attn.proj.weight", (size_t)E * E).data.data(), E, hp.data());
      proj_packed[i].ensure(hp.size() * 2);
      TTR_HIP_CHECK(hipMemcpy(proj_packed[i].p, hp.data(), hp.size() * 2, hipMemcpyHostToDevice));
    }
  }
  vec("encoder.norm.weight", E); vec("encoder.norm.bias", E);
  const std::string d = "decoder.layers.0.";
  lin("self_kv", d + "self_attn.in_proj_weight", d + "self_attn.in_proj_bias", 2 * E, E, E, 3 * E);
  lin("self_out", d + "self_attn.out_proj.weight", d + "self_attn.out_proj.bias", E, E);
  lin("cross_q", d + "cross_attn.in_proj_weight", d + "cross_attn.in_proj_bias", E, E, 0, 3 * E);
  lin("cross_kv", d + "cross_attn.in_proj_weight", d + "cross_attn.in_proj_bias", 2 * E, E, E, 3 * E);
  lin("cross_out", d + "cross_attn.out_proj.weight", d + "cross_attn.out_proj.bias", E, E);
  lin("ffn1", d + "linear1.weight", d + "linear1.bias", 4 * E, E);
  lin("ffn2", d + "linear2.weight", d + "linear2.bias", E, 4 * E);
  if (prec == kBF16) {   // the refinement pass runs cross_out + norm2 + FFN + final norm through mlp_fused.hip
    std::vector<uint16_t> h((size_t)E * 4 * E), hp((size_t)E * E);
    pack_mlp_w1(wf.get(d + "linear1.weight", (size_t)4 * E * E).data.data(), h.data());
    dec_ffn1_packed.ensure(h.size() * 2); TTR_HIP_CHECK(hipMemcpy(dec_ffn1_packed.p, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    pack_mlp_w2(wf.get(d + "linear2.weight", (size_t)E * 4 * E).data.data(), 4 * E, h.data());
    dec_ffn2_packed.ensure(h.size() * 2); TTR_HIP_CHECK(hipMemcpy(dec_ffn2_packed.p, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    pack_mlp_w2(wf.get(d + "cross_attn.out_proj.weight", (size_t)E * E).data.data(), E, hp.data());
    dec_co_packed.ensure(hp.size() * 2); TTR_HIP_CHECK(hipMemcpy(dec_co_packed.p, hp.data(), hp.size() * 2, hipMemcpyHostToDevice));
  }
  for (const char* n : {"norm1", "norm2", "norm_q", "norm_c"}) { vec(d + n + ".weight", E); vec(d + n + ".bias", E); }
  vec("decoder.norm.weight", E); vec("decoder.norm.bias", E);
  if (prec == kSplit) {   // 95 classes in 96 weight rows (a zero row): the head gets f16 planes and runs on the skinny split kernel in the AR steps
    const auto& w = wf.get("head.weight", (size_t)95 * E);
    const auto& b = wf.get("head.bias", (size_t)95);
    upload_linear(pq["head"], w.data.data(), 95, E, b.data.data(), 96, E);
    pq["head"].cout_valid = 95;
  } else lin("head", "head.weight", "head.bias", 95, E);
  if (prec == kSplit)   // gemm_sp.hip's loader pieces of every recogniser linear it may run
    for (auto& kv : pq) tile_planes(kv.second);
  vec("text_embed.embedding.weight", 97 * E);
  vec("pos_queries", 26 * E);
  // Qself[i] = Wq . norm_q(pos_queries[i]) + bq : crop independent, computed once on the host in fp32
  {
    const auto& pos = wf.get("pos_queries", 26 * E).data;
    const auto& g = wf.get(d + "norm_q.weight", E).data;
    const auto& bt = wf.get(d + "norm_q.bias", E).data;
    const auto& w = wf.get(d + "self_attn.in_proj_weight", (size_t)3 * E * E).data;
    const auto& b = wf.get(d + "self_attn.in_proj_bias", 3 * E).data;
    std::vector<float> q((size_t)26 * E), ln(E);
    for (int i = 0; i < 26; ++i) {
      float mean = 0.f;
      for (int c = 0; c < E; ++c) mean += pos[i * E + c];
      mean /= E;
      float var = 0.f;
      for (int c = 0; c < E; ++c) { float dd = pos[i * E + c] - mean; var += dd * dd; }
      var /= E;
      float rstd = 1.0f / std::sqrt(var + 1e-5f);
      for (int c = 0; c < E; ++c) ln[c] = (pos[i * E + c] - mean) * rstd * g[c] + bt[c];
      for (int o = 0; o < E; ++o) {
        float acc = 0.f;
        for (int c = 0; c < E; ++c) acc += w[(size_t)o * E + c] * ln[c];
        q[(size_t)i * E + o] = acc + b[o];
      }
    }
    upload_f32(qself, q.data(), q.size());
  }
}

Engine::Engine(const std::string& dir, const ttr_config& c) : cfg(c) {
  { const char* v = getenv("TUATARA_VERBOSE"); verbose = cfg.verbose != 0 || (v && *v && std::string(v) != "0"); }
  prec = cfg.precision == TTR_PREC_F32 ? kF32 : cfg.precision == TTR_PREC_F16X4 ? kSplit : kBF16;
  es = prec == kBF16 ? 2 : 4;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) throw std::runtime_error("no HIP device available: the tuatara engine has no CPU fallback");
  TTR_HIP_CHECK(hipSetDevice(cfg.device));
  TTR_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  TTR_HIP_CHECK(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
  TTR_HIP_CHECK(hipStreamCreateWithFlags(&recog_stream, hipStreamNonBlocking));
  TTR_HIP_CHECK(hipStreamCreateWithFlags(&lane_stream, hipStreamNonBlocking));
  for (hipEvent_t* e : {&lane_go, &lane_done, &resize_done}) TTR_HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
  TTR_HIP_CHECK(hipEventCreateWithFlags(&copy_ev, hipEventDisableTiming));
  for (auto& x : done_ev) TTR_HIP_CHECK(hipEventCreateWithFlags(&x, hipEventDisableTiming));
  for (auto& sl : evr) for (auto& x : sl) TTR_HIP_CHECK(hipEventCreate(&x));
  {   // host threads for the per-page calipers / decode: at most 15, and a fair share of the box when several ranks run on it
    // (torch.distributed.run exports LOCAL_WORLD_SIZE); TUATARA_HOST_THREADS overrides
    int hw = std::max(1, (int)std::thread::hardware_concurrency());
    if (const char* lw = getenv("LOCAL_WORLD_SIZE")) { const int n = atoi(lw); if (n > 1) hw = std::max(1, hw / n); }
    int workers = std::min(15, std::max(1, hw - 1));
    if (const char* ht = getenv("TUATARA_HOST_THREADS")) { const int n = atoi(ht); if (n >= 1) workers = std::min(64, n); }
    host_pool.reset(new HostPool(workers));
  }
  for (auto& x : ev) TTR_HIP_CHECK(hipEventCreate(&x));
  range_word.ensure(64); h_range.ensure(64);
  TTR_HIP_CHECK(hipMemset(range_word.p, 0, 64)); memset(h_range.p, 0, 64);
  load_craft(dir);
  load_parseq(dir);
}

Engine::~Engine() {
  (void)hipSetDevice(cfg.device);
  for (hipStream_t st : {stream, recog_stream, lane_stream, copy_stream, up_stream}) if (st) (void)hipStreamSynchronize(st);   // nothing in flight when the buffers go
  for (auto& x : ev) if (x) (void)hipEventDestroy(x);
  for (auto& x : prof_pool) (void)hipEventDestroy(x);
  for (auto& x : group_ev) (void)hipEventDestroy(x);
  if (copy_ev) (void)hipEventDestroy(copy_ev);
  for (auto& x : done_ev) if (x) (void)hipEventDestroy(x);
  for (auto& sl : evr) for (auto& x : sl) if (x) (void)hipEventDestroy(x);
  for (auto& x : up_ev) if (x) (void)hipEventDestroy(x);
  if (up_stream) (void)hipStreamDestroy(up_stream);
  if (copy_stream) (void)hipStreamDestroy(copy_stream);
  if (recog_stream) (void)hipStreamDestroy(recog_stream);
  for (hipEvent_t e : {lane_go, lane_done, resize_done}) if (e) (void)hipEventDestroy(e);
  if (lane_stream) (void)hipStreamDestroy(lane_stream);
  if (stream) (void)hipStreamDestroy(stream);
}

DevBuf& Engine::ws(size_t idx, size_t bytes, bool zero_new) {
  auto& craft_ws = craft_ws_cur();
  while (craft_ws.size() <= idx) craft_ws.emplace_back(new DevBuf());
  DevBuf& d = *craft_ws[idx];
  const size_t cap_before = d.cap;          // (not the pointer: the allocator may hand the grown block the old address)
  d.ensure(bytes);
  if (zero_new && d.cap != cap_before) TTR_HIP_CHECK(hipMemsetAsync(d.p, 0, d.cap, stream));   // padding channels are written once, here
  return d;
}

}  // namespace ttr
