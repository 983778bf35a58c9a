// image_to_data (tuatara.cpp:314-512) over batches of device-resident pages: detector + CCL -> boxes -> crop batch -> recogniser -> strings,
// in four phases so that several batches can be in flight; the multi-GPU exchange points; the sharded latency mode.
#include <deque>

#include "engine.h"

namespace ttr {

void Engine::allgather_host(const void* mine, size_t bytes, void* all) {
  Comm& c = *comm;
  const size_t b = std::max<size_t>(bytes, 1);
  c.h_in.ensure(b); c.h_out.ensure(b * c.world); c.d_in.ensure(b); c.d_out.ensure(b * c.world);
  if (bytes) memcpy(c.h_in.p, mine, bytes);
  TTR_HIP_CHECK(hipMemcpyAsync(c.d_in.p, c.h_in.p, b, hipMemcpyHostToDevice, copy_stream));
  c.tr->all_gather(c.d_in.p, c.d_out.p, b, true, copy_stream);
  TTR_HIP_CHECK(hipMemcpyAsync(c.h_out.p, c.d_out.p, b * c.world, hipMemcpyDeviceToHost, copy_stream));
  TTR_HIP_CHECK(hipStreamSynchronize(copy_stream));
  if (bytes && all) memcpy(all, c.h_out.p, bytes * c.world);
}

void Engine::ccl_launch(const float* d_heat, int p0, int pages, int total, int g, int H2, int W2, int lane) {
  if (p0 == 0) { ccl.ensure(total, H2 * W2, cfg.max_components); h_counters.ensure((size_t)total * 8); }
  ccl.cal_cap_now = tn.gpu_calipers == 2 ? 512 : CclBatch::kCalCap;
  launch_ccl(d_heat, pages, H2, W2, cfg.text_threshold, cfg.link_threshold, cfg.low_text, cfg.min_area, ccl.view(p0, lane), stream);
  if (tn.gpu_calipers) launch_ccl_rects(ccl.view(p0, lane), pages, H2, W2, stream);   // minAreaRect of every candidate, on the stream right behind its row extremes
  TTR_HIP_CHECK(hipMemcpyAsync(h_counters.as<int>() + 2 * p0, ccl.counters.as<int>() + 2 * p0, (size_t)pages * 8, hipMemcpyDeviceToHost, stream));
  while ((int)group_ev.size() <= g) { hipEvent_t e; TTR_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); group_ev.push_back(e); }
  TTR_HIP_CHECK(hipEventRecord(group_ev[g], stream));
}

void Engine::ccl_collect(int p0, int pages, int g, int H2, int W2, std::vector<std::vector<RRect>>& det) {
  const int* counters = h_counters.as<int>() + 2 * p0;
  const double tc0 = now_us();
  spin_event(group_ev[g]);
  const double tc1 = now_us();
  // two strided copies bring every page's candidates and row extremes over (width = the busiest page's share)
  int max_c = 0, max_r = 0;
  for (int pg = 0; pg < pages; ++pg) {
    if (counters[2 * pg] > cfg.max_components) throw std::runtime_error("too many text components on a page; raise ttr_config.max_components");
    max_c = std::max(max_c, counters[2 * pg]); max_r = std::max(max_r, counters[2 * pg + 1]);
  }
  const size_t pitch_c = (size_t)max_c * 32, pitch_r = (size_t)max_r * 8;
  std::vector<size_t> off_c(pages), off_r(pages);
  for (int pg = 0; pg < pages; ++pg) { off_c[pg] = pg * (pitch_c / 4); off_r[pg] = pg * (pitch_r / 4); }
  h_cand.ensure(pitch_c * pages + 4); h_rows.ensure(pitch_r * pages + 4);
  int* cand = h_cand.as<int>();
  int* rw = h_rows.as<int>();
  const CclBuffers v = ccl.view(p0);
  if (tn.gpu_calipers && max_c > 0) {
    // the calipers ran on the GPU (ccl_rects_kernel): candidates (for the label order) + 32 bytes of raw result each; sides and angle here (the host's libm)
    const size_t pitch_q = (size_t)max_c * 32;
    h_rects_f.ensure(pitch_q * pages + 4);
    float* rq = h_rects_f.as<float>();
    TTR_HIP_CHECK(hipMemcpy2DAsync(cand, pitch_c, v.cand, (size_t)ccl.max_cand * 32, pitch_c, pages, hipMemcpyDeviceToHost, copy_stream));
    TTR_HIP_CHECK(hipMemcpy2DAsync(rq, pitch_q, v.rects, (size_t)ccl.max_cand * 32, pitch_q, pages, hipMemcpyDeviceToHost, copy_stream));
    TTR_HIP_CHECK(hipEventRecord(copy_ev, copy_stream));
    spin_event(copy_ev);
    const double tc2g = now_us();
    bool pool_full = false;
    for (int pg = 0; pg < pages && !pool_full; ++pg)
      for (int i = 0; i < counters[2 * pg]; ++i)
        if (reinterpret_cast<const int*>(rq + (size_t)pg * (pitch_q / 4) + 8 * (size_t)i)[0] == 2) { pool_full = true; break; }
    if (!pool_full) {
      for (int pg = 0; pg < pages; ++pg) {
        const int n = counters[2 * pg];
        const int* cd = cand + off_c[pg];
        const float* q = rq + (size_t)pg * (pitch_q / 4);
        std::vector<int> order(n);
        for (int i = 0; i < n; ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](int a, int b) { return cd[8 * a] < cd[8 * b]; });  // label order = ascending root
        for (int i : order) {
          const int kind = reinterpret_cast<const int*>(q + 8 * (size_t)i)[0];
          if (kind != 1 && kind != 3 && kind != 4) continue;
          det[p0 + pg].push_back(finish_min_area_rect(kind, q + 8 * (size_t)i + 1));
        }
      }
      host_us[1] += (float)(tc1 - tc0); host_us[2] += (float)(tc2g - tc1); host_us[3] += (float)(now_us() - tc2g);
      return;
    }
    // (the scratch pool was too small for this group's hulls: the host's calipers below, as without gpu_calipers)
  }
  if (max_c > 0) {
    TTR_HIP_CHECK(hipMemcpy2DAsync(cand, pitch_c, v.cand, (size_t)ccl.max_cand * 32, pitch_c, pages, hipMemcpyDeviceToHost, copy_stream));
    TTR_HIP_CHECK(hipMemcpy2DAsync(rw, pitch_r, v.rows_packed, (size_t)ccl.npx * 8, pitch_r, pages, hipMemcpyDeviceToHost, copy_stream));
    TTR_HIP_CHECK(hipEventRecord(copy_ev, copy_stream));
    spin_event(copy_ev);
  }
  const double tc2 = now_us();
  // the calipers of a page depend on nothing but that page: a few host threads share the group
  parallel_pages(pages, [&](int pg) {
    const int n = counters[2 * pg];
    const int* cd = cand + off_c[pg];
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return cd[8 * a] < cd[8 * b]; });  // label order = ascending root
    for (int i : order) {
      const int* c = &cd[8 * i];
      Component comp{c[0], c[1], c[2], c[3], c[4], c[5], rw + off_r[pg] + 2 * (size_t)c[6]};
      RRect r;
      if (component_to_rect(comp, H2, W2, &r)) det[p0 + pg].push_back(r);
    }
  });
  host_us[1] += (float)(tc1 - tc0); host_us[2] += (float)(tc2 - tc1); host_us[3] += (float)(now_us() - tc2);
}

void Engine::detect_enqueue(PageBatch& B) {
  range_use(kRangeDet0 + (B.slot & 1));   // the detector's kernels of this batch watch its own word (engine.h)
  if (B.h <= 0 || B.w <= 0) throw std::runtime_error("Error reading image from file");  // image.empty(), tuatara.cpp:344
  B.g = canvas_geometry(B.h, B.w, cfg.canvas_size, cfg.mag_ratio);
  if (B.g.target_h <= 0 || B.g.target_w <= 0) throw std::runtime_error("image too thin to resize");
  B.H = B.g.h32; B.W = B.g.w32; B.H2 = B.H / 2; B.W2 = B.W / 2;
  B.page_bytes = (size_t)B.h * B.w * 3;
  const int n = B.n, H = B.H, W = B.W, H2 = B.H2, W2 = B.W2;
  canvas.ensure((size_t)n * H * W * 3);
  heat.ensure((size_t)n * H2 * W2 * 2 * 4);
  TTR_HIP_CHECK(hipEventRecord(ev[0], stream));
  launch_resize_pad_u8(B.d_pages, B.h, B.w, B.w * 3, canvas.as<uint8_t>(), B.g.target_h, B.g.target_w, H, W, 1, stream, n, B.page_bytes);
  // CRAFT in groups of <= 16 pages: bounds the activation workspace (~0.5 GB/page) and keeps every tensor
  // inside the 2 GiB window gemm2's 32-bit buffer offsets address.  Each group's CCL follows its CRAFT, so the host reads
  // group g's components back (and runs its calipers) while the GPU is busy with group g + 1.
  int GP = tn.craft_group;
  if (prec == kSplit) {   // three f16 planes per value: the widest tensor (64 channels at full resolution) must stay inside the 2 GiB window
    const size_t per_page = (size_t)H * W * 64 * (tn.craft_products == 4 ? 6 : 4);
    GP = (int)std::max<size_t>(1, std::min<size_t>(GP, (((size_t)1 << 31) - 1) / per_page));
    if (GP >= 8 && n % 8 == 0 && tn.craft_group >= 8) GP = 8;   // (even groups: 32 pages = 4 x 8 rather than 10 + 10 + 10 + 2)
  }
  B.group = GP;
  const int groups = (n + GP - 1) / GP;
  // Two detector lanes (tn.craft_lanes, engine.h): odd groups on lane_stream with their own workspaces, half a group behind the even ones, so that a lane's
  // matrix-bound full-resolution layers run beside the other lane's HBM-bound U-Net tail and head (a group alone: 9.7 ms of the one, 2.8 of the other)
  const bool two = prec == kSplit && tn.craft_lanes == 2 && groups >= 2;
  ccl.split_pool = two;
  if (two) {
    TTR_HIP_CHECK(hipEventRecord(resize_done, stream));
    TTR_HIP_CHECK(hipStreamWaitEvent(lane_stream, resize_done, 0));     // (the canvas; and everything the main stream held before it: the previous batch's detector)
    lane_go_pending = true;
  }
  struct LaneGuard {   // the odd groups borrow the engine's `stream` and workspace selector; restored also when a launch throws
    Engine& E; bool on = false;
    void enter() { E.prof_break(); std::swap(E.stream, E.lane_stream); E.ws_sel = 1; on = true; }
    void leave() { if (on) { E.prof_break(); std::swap(E.stream, E.lane_stream); E.ws_sel = 0; on = false; } }
    ~LaneGuard() { if (on) { std::swap(E.stream, E.lane_stream); E.ws_sel = 0; } E.lane_go_pending = false; }
  } lane_guard{*this};
  for (int gi = 0; gi < groups; ++gi) {
    const int p0 = gi * GP, cnt = std::min(GP, n - p0);
    const int lane = two ? (gi & 1) : 0;
    if (lane) {
      lane_guard.enter();
      if (gi == 1) TTR_HIP_CHECK(hipStreamWaitEvent(stream, lane_go, 0));   // (recorded inside group 0's forward pass, behind slice3.20)
    }
    craft_forward(canvas.as<uint8_t>() + (size_t)p0 * H * W * 3, cnt, H, W, heat.as<float>() + (size_t)p0 * H2 * W2 * 2);
    if (gi == groups - 1) TTR_HIP_CHECK(hipEventRecord(ev[1], stream));
    ccl_launch(heat.as<float>() + (size_t)p0 * H2 * W2 * 2, p0, cnt, n, gi, H2, W2, lane);
    if (lane) {
      if (gi + 2 >= groups) TTR_HIP_CHECK(hipEventRecord(lane_done, stream));   // this lane's last group
      lane_guard.leave();
    }
  }
  if (two) TTR_HIP_CHECK(hipStreamWaitEvent(stream, lane_done, 0));       // the batch's detector is complete when the main stream gets here
  range_fetch(kRangeDet0 + (B.slot & 1));
  while ((int)group_ev.size() <= groups) { hipEvent_t e; TTR_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); group_ev.push_back(e); }
  TTR_HIP_CHECK(hipEventRecord(group_ev[groups], stream));                // (behind the word's copy: detect_collect_local waits for it)
  B.det_groups = groups;
  TTR_HIP_CHECK(hipEventRecord(ev[2], stream));
}

void Engine::detect_collect(PageBatch& B, std::exception_ptr pre) {
  if (!comm) { if (pre) std::rethrow_exception(pre); detect_collect_local(B); return; }
  std::exception_ptr err = pre;
  if (!err) { try { detect_collect_local(B); } catch (...) { err = std::current_exception(); } }
  const int world = comm->world, n = B.n;
  int32_t hdr[2] = {err ? -1 : 0, n};
  std::vector<int32_t> all(2 * (size_t)world, 0);
  allgather_host(hdr, 8, all.data());
  if (err) std::rethrow_exception(err);
  for (int r = 0; r < world; ++r) {
    if (all[2 * r] < 0) throw std::runtime_error("multi-GPU batch: rank " + std::to_string(r) + " failed before the exchange; the batch is dropped on every rank");
    if (all[2 * r + 1] != n) throw std::runtime_error("multi-GPU batch: rank " + std::to_string(r) + " passed " + std::to_string(all[2 * r + 1]) + " pages, this rank " + std::to_string(n) +
                                                      ": every rank must push the same number of pages per batch");
  }
  // counts (host-side exchange on the control communicator), so that every rank knows the payload's size
  std::vector<int32_t> mine(n, 0);
  for (int pg : B.page_of) mine[pg]++;
  B.all_counts.assign((size_t)world * n, 0);
  allgather_host(mine.data(), (size_t)n * 4, B.all_counts.data());
  B.cap = GatherLayout::from_counts(B.all_counts.data(), world, n).cap;
}

void Engine::detect_collect_local(PageBatch& B) {
  const int n = B.n, GP = B.group, groups = (n + GP - 1) / GP;
  const float ratio_w = 1.f / B.g.ratio, ratio_h = 1.f / B.g.ratio;   // tuatara.cpp:360-361
  std::vector<std::vector<RRect>> dets(n);
  B.boxes.assign(n, std::vector<RRect>());
  B.rects.clear(); B.page_of.clear();        // x0,y0,x1,y1,page per crop; page index per crop
  host_us[1] = host_us[2] = host_us[3] = 0.f;
  for (int gi = 0; gi < groups; ++gi) ccl_collect(gi * GP, std::min(GP, n - gi * GP), gi, B.H2, B.W2, dets);
  // the detector's range word of THIS batch, before any of its boxes is used: a saturated heat map fails this batch and no other
  if (range_flag_ptr() && B.det_groups == groups) { spin_event(group_ev[groups]); range_verify(kRangeDet0 + (B.slot & 1), "the detector of a batch of pages"); }
  if (tn.detector_only) for (auto& d : dets) d.clear();   // profiling (tools/prof_pages.py): the detector and CCL run, nothing goes to the recogniser
  if (tn.bench_grid_boxes) {   // benchmark workload control (tuning key "bench_grid_boxes", tuatara_hip_debug.h): the detector's work is done (and timed); 40 fixed boxes per page go on
    for (int i = 0; i < n; ++i) {
      dets[i].clear();
      for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 5; ++c) {
          RRect g;
          g.cx = (c + 0.5f) * (float)B.W2 / 5.f; g.cy = (r + 0.5f) * (float)B.H2 / 8.f; g.w = 75.f * B.g.ratio; g.h = 20.f * B.g.ratio; g.angle = 0.f;
          dets[i].push_back(g);
        }
    }
  }
  for (int i = 0; i < n; ++i) {
    for (const RRect& r : dets[i]) {
      RRect b = adjust_coordinates(r, ratio_w, ratio_h);            // :406
      int xywh[4];
      bounding_rect(b, xywh);                                       // :416
      int x0 = xywh[0], y0 = xywh[1], x1 = xywh[0] + xywh[2], y1 = xywh[1] + xywh[3];
      if (cfg.strict_crops) {
        if (x0 < 0 || y0 < 0 || x1 > B.w || y1 > B.h) throw std::runtime_error("text box leaves the image (cv::Exception in the reference, tuatara.cpp:416)");
      } else {
        x0 = std::max(x0, 0); y0 = std::max(y0, 0); x1 = std::min(x1, B.w); y1 = std::min(y1, B.h);
      }
      if (x1 <= x0 || y1 <= y0) continue;
      B.boxes[i].push_back(b);
      B.rects.insert(B.rects.end(), {x0, y0, x1, y1, i});
      B.page_of.push_back(i);
    }
  }
  B.N = (int)B.page_of.size();
}

void Engine::recog_enqueue(PageBatch& B) {
  const int N = B.N, sl = B.slot;
  range_use(kRangeRec0 + (sl & 1));          // the recogniser's kernels of this batch watch the slot's own word
  h_ids[sl].ensure((size_t)N * 26 * 4 + 4);
  TTR_HIP_CHECK(hipEventRecord(evr[sl][0], stream));
  if (N > 0) {
    rects_dev.ensure(B.rects.size() * 4);
    h_rects[sl].ensure(B.rects.size() * 4);
    memcpy(h_rects[sl].p, B.rects.data(), B.rects.size() * 4);
    crops.ensure((size_t)N * 32 * 128 * 3);
    logits.ensure((size_t)N * 26 * 95 * 4);
    ids_dev.ensure((size_t)std::max(N, B.cap) * 26 * 4);
    TTR_HIP_CHECK(hipMemcpyAsync(rects_dev.p, h_rects[sl].p, B.rects.size() * 4, hipMemcpyHostToDevice, stream));
    launch_pack_crops(B.d_pages, B.page_bytes, B.w * 3, rects_dev.as<int>(), crops.as<uint8_t>(), N, stream);
    TTR_HIP_CHECK(hipEventRecord(evr[sl][1], stream));
    parseq_forward(crops.as<uint8_t>(), N, logits.as<float>(), nullptr, ids_dev.as<int>());
    TTR_HIP_CHECK(hipEventRecord(evr[sl][2], stream));
    TTR_HIP_CHECK(hipMemcpyAsync(h_ids[sl].as<int32_t>(), ids_dev.p, (size_t)N * 26 * 4, hipMemcpyDeviceToHost, stream));
  } else {
    TTR_HIP_CHECK(hipEventRecord(evr[sl][1], stream));
    TTR_HIP_CHECK(hipEventRecord(evr[sl][2], stream));
  }
  if (comm && B.cap > 0) {   // the payload: cap rows of 26 ids per rank, straight from the recogniser's device buffer
    const size_t per = (size_t)B.cap * 26;
    ids_dev.ensure(per * 4);
    gath_dev[sl].ensure(per * 4 * comm->world);
    h_gath[sl].ensure(per * 4 * comm->world);
    comm->tr->all_gather(ids_dev.p, gath_dev[sl].p, per * 4, false, stream);
    TTR_HIP_CHECK(hipMemcpyAsync(h_gath[sl].p, gath_dev[sl].p, per * 4 * comm->world, hipMemcpyDeviceToHost, stream));
  }
  range_fetch(kRangeRec0 + (sl & 1));
  TTR_HIP_CHECK(hipEventRecord(done_ev[sl], stream));
  B.enqueued = true;
}

void Engine::finish(PageBatch& B, std::vector<Result>& results) {
  const int n = B.n, N = B.N;
  results.assign(n, Result());
  const double th2 = now_us();
  spin_event(done_ev[B.slot]);
  range_verify(kRangeRec0 + (B.slot & 1), "the recogniser of a batch of pages");
  const double th3 = now_us();
  // stage times: detector events belong to the latest batch enqueued (complete by now: its components were collected), recogniser events to this one
  (void)hipEventElapsedTime(&stage_ms[0], ev[0], ev[1]); (void)hipEventElapsedTime(&stage_ms[1], ev[1], ev[2]);
  (void)hipEventElapsedTime(&stage_ms[2], evr[B.slot][0], evr[B.slot][1]); (void)hipEventElapsedTime(&stage_ms[3], evr[B.slot][1], evr[B.slot][2]);
  if (profiling) prof_collect();
  const double th4 = now_us();
  if (comm) {   // compact the gathered payload: (rank, page, crop) order, no padding
    const GatherLayout L = GatherLayout::from_counts(B.all_counts.data(), comm->world, n);
    last_gathered.world = L.world; last_gathered.pages = n; last_gathered.counts = B.all_counts;
    last_gathered.ids.resize((size_t)L.first.back() * 26);
    const int32_t* g = h_gath[B.slot].as<int32_t>();
    for (int r = 0; r < L.world; ++r)
      if (L.total[r]) memcpy(&last_gathered.ids[(size_t)L.first[(size_t)r * n] * 26], g + (size_t)r * B.cap * 26, (size_t)L.total[r] * 26 * 4);
  }
  const int32_t* ids = h_ids[B.slot].as<int32_t>();
  // crops are ordered by page: page pg owns crops [first[pg], first[pg + 1]); pages decode independently
  std::vector<int> first(n + 1, 0);
  for (int c = 0; c < N; ++c) first[B.page_of[c] + 1]++;
  for (int pg = 0; pg < n; ++pg) first[pg + 1] += first[pg];
  auto decode_page = [&](int pg) {
    Result& r = results[pg];
    const int c0 = first[pg], cnt = first[pg + 1] - c0;
    r.text.reserve(cnt); r.bbox.reserve((size_t)cnt * 4);
    r.ids.assign(&ids[(size_t)c0 * 26], &ids[(size_t)(c0 + cnt) * 26]);
    for (int k = 0; k < cnt; ++k) {
      r.text.push_back(tok.decode(&ids[(size_t)(c0 + k) * 26], 26));   // :486-505
      float bb[4];
      tesseract_bbox(B.boxes[pg][k], bb);                               // :511
      r.bbox.insert(r.bbox.end(), bb, bb + 4);
    }
  };
  if (N >= 256) parallel_pages(n, decode_page);
  else for (int pg = 0; pg < n; ++pg) decode_page(pg);
  host_us[5] = (float)(th3 - th2); host_us[6] = (float)(th4 - th3); host_us[7] = (float)(now_us() - th4);
  B.live = false; B.enqueued = false;
}

void Engine::run_pages(const uint8_t* d_pages, int n, int h, int w, std::vector<Result>& results) {
  results.assign(n, Result());
  if (n <= 0) return;
  if (q1.live || q2.live) throw std::runtime_error("streamed batches are in flight: call ttr_stream_flush until it returns none");
  const double th0 = now_us();
  // the reference's progress lines (tuatara.cpp:328-329, :342, :421, :434: the models are loaded once per engine here, so those
  // lines report a fact; :386, :488, :509), on request only: callers do not parse stdout
  if (verbose) std::cout << ttr_version() << " (HIP " << HIP_VERSION_MAJOR << "." << HIP_VERSION_MINOR << ")\ncraft model loaded" << std::endl;
  PageBatch B;
  B.d_pages = d_pages; B.n = n; B.h = h; B.w = w; B.slot = 0;
  std::exception_ptr pre;
  { RangeScope r("ttr:detect_enqueue"); try { detect_enqueue(B); } catch (...) { if (!comm) throw; pre = std::current_exception(); } }
  host_us[0] = (float)(now_us() - th0);
  if (verbose) std::cout << "post processing craft predictions..." << std::endl;
  { RangeScope r("ttr:detect_collect"); detect_collect(B, pre); }
  const double th1 = now_us();
  if (verbose) std::cout << "loading parseq model...\nparseq model loaded" << std::endl;
  { RangeScope r("ttr:recog_enqueue"); recog_enqueue(B); }
  host_us[4] = (float)(now_us() - th1);
  if (verbose) std::cout << "Running tokenizer..." << std::endl;
  { RangeScope r("ttr:finish"); finish(B, results); }
  if (verbose) std::cout << "Elapsed time: " << (now_us() - th0) * 1e-6 << " seconds " << std::endl;
}

void Engine::run_pages_sharded(const uint8_t* d_pages, int n, int h, int w, std::vector<Result>& results) {
  if (!comm) throw std::runtime_error("latency mode needs a communicator (ttr_engine_attach_comm)");
  if (q1.live || q2.live) throw std::runtime_error("streamed batches are in flight: call ttr_stream_flush until it returns none");
  Comm* const c = comm;
  const int world = c->world, rank = c->rank;
  PageBatch B;
  int32_t hdr[2] = {0, 0};                                     // {pages, crops} of rank 0
  comm = nullptr;                                              // (the detector below is not the throughput mode's: no per-batch gather)
  try {
    if (rank == 0) {
      if (!d_pages || n <= 0) throw std::runtime_error("latency mode: rank 0 passes the pages");
      B.d_pages = d_pages; B.n = n; B.h = h; B.w = w; B.slot = 0;
      detect_enqueue(B);
      detect_collect(B);
      hdr[0] = n; hdr[1] = B.N;
    }
  } catch (...) { comm = c; hdr[0] = -1; std::vector<int32_t> all(2 * world); allgather_host(hdr, 8, all.data()); throw; }
  comm = c;
  std::vector<int32_t> all(2 * (size_t)world);
  allgather_host(hdr, 8, all.data());
  if (all[0] < 0) throw std::runtime_error("latency mode: rank 0 failed in the detector");
  const int pages = all[0], N = all[1];
  results.assign(rank == 0 ? pages : std::max(n, 0), Result());
  if (N == 0) return;
  const int per = (N + world - 1) / world;
  const int lo = std::min(N, rank * per), hi = std::min(N, lo + per);
  crops.ensure((size_t)world * per * 32 * 128 * 3);           // (the last shard may be ragged: the buffer holds world * per crops)
  if (rank == 0) {
    rects_dev.ensure(B.rects.size() * 4);
    h_rects[0].ensure(B.rects.size() * 4);
    memcpy(h_rects[0].p, B.rects.data(), B.rects.size() * 4);
    TTR_HIP_CHECK(hipMemcpyAsync(rects_dev.p, h_rects[0].p, B.rects.size() * 4, hipMemcpyHostToDevice, stream));
    launch_pack_crops(B.d_pages, B.page_bytes, B.w * 3, rects_dev.as<int>(), crops.as<uint8_t>(), N, stream);
  }
  c->tr->broadcast(crops.p, (size_t)N * 32 * 128 * 3, 0, stream);
  logits.ensure((size_t)per * 26 * 95 * 4);
  ids_dev.ensure((size_t)per * 26 * 4);
  range_use(kRangeRec0);
  if (hi > lo) parseq_forward(crops.as<uint8_t>() + (size_t)lo * 32 * 128 * 3, hi - lo, logits.as<float>(), nullptr, ids_dev.as<int>());
  gath_dev[0].ensure((size_t)world * per * 26 * 4);
  h_gath[0].ensure((size_t)world * per * 26 * 4);
  c->tr->all_gather(ids_dev.p, gath_dev[0].p, (size_t)per * 26 * 4, false, stream);
  TTR_HIP_CHECK(hipMemcpyAsync(h_gath[0].p, gath_dev[0].p, (size_t)world * per * 26 * 4, hipMemcpyDeviceToHost, stream));
  range_fetch(kRangeRec0);
  TTR_HIP_CHECK(hipEventRecord(done_ev[0], stream));
  spin_event(done_ev[0]);
  range_verify(kRangeRec0, "the recogniser of a sharded page");
  if (rank != 0) return;
  const int32_t* ids = h_gath[0].as<int32_t>();                // shard r occupies rows [r * per, r * per + its size): crop k = row k
  std::vector<int> first(pages + 1, 0);
  for (int k = 0; k < N; ++k) first[B.page_of[k] + 1]++;
  for (int pg = 0; pg < pages; ++pg) first[pg + 1] += first[pg];
  for (int pg = 0; pg < pages; ++pg) {
    Result& r = results[pg];
    const int c0 = first[pg], cnt = first[pg + 1] - c0;
    r.ids.assign(&ids[(size_t)c0 * 26], &ids[(size_t)(c0 + cnt) * 26]);
    for (int k = 0; k < cnt; ++k) {
      r.text.push_back(tok.decode(&ids[(size_t)(c0 + k) * 26], 26));
      float bb[4];
      tesseract_bbox(B.boxes[pg][k], bb);
      r.bbox.insert(r.bbox.end(), bb, bb + 4);
    }
  }
}

void Engine::stream_push(const uint8_t* d_pages, int n, int h, int w, std::vector<Result>& prev_results, int& prev_n) {
  prev_results.clear(); prev_n = 0;
  if (n <= 0) throw std::runtime_error("stream_push: empty batch");
  const double th0 = now_us();
  PageBatch B;
  B.d_pages = d_pages; B.n = n; B.h = h; B.w = w;
  B.slot = q1.live ? (q1.slot ^ 1) : 0;     // from the pipeline's state, not a counter: a push that throws leaves q1 / q2 and the slot parity as they were
  std::exception_ptr pre;      // (with a communicator: a failing rank still takes part in this batch's header exchange, detect_collect)
  stream_fail_age = 0;
  { RangeScope r("ttr:detect_enqueue"); try { detect_enqueue(B); } catch (...) { if (!comm) throw; pre = std::current_exception(); } }
  host_us[0] = (float)(now_us() - th0);
  const double th1 = now_us();
  if (q1.live && !q1.enqueued) {
    RangeScope r("ttr:recog_enqueue");
    struct Flag { bool& f; ~Flag() { f = false; } } flag{streaming_recog};
    streaming_recog = true;
    // (recog_overlap: everything recog_enqueue puts on "the stream" - packer, recogniser, id copy, completion event - goes to the recogniser's own stream)
    struct StreamSwap { Engine& E; bool on; StreamSwap(Engine& e, bool o) : E(e), on(o) { if (on) std::swap(E.stream, E.recog_stream); } ~StreamSwap() { if (on) std::swap(E.stream, E.recog_stream); } }
        swap_guard{*this, tn.recog_overlap != 0};
    recog_enqueue(q1);
  }
  host_us[4] = (float)(now_us() - th1);
  { RangeScope r("ttr:detect_collect"); detect_collect(B, pre); }
  // (a batch whose recogniser tripped the range guard fails HERE, once: the pipeline still advances - on every rank alike, so no rank skips a collective its
  // peers issue - and the neighbours' results survive)
  std::exception_ptr fin;
  if (q2.live) { RangeScope r("ttr:finish"); prev_n = q2.n; try { finish(q2, prev_results); } catch (...) { fin = std::current_exception(); prev_results.clear(); prev_n = 0; q2.live = false; } }
  if (q1.live) q2 = std::move(q1);
  q1 = std::move(B);
  q1.live = true; q1.enqueued = false;
  if (fin) { stream_fail_age = 2; std::rethrow_exception(fin); }
}

void Engine::stream_flush(std::vector<Result>& prev_results, int& prev_n) {
  prev_results.clear(); prev_n = 0;
  stream_fail_age = 2;
  if (q1.live && !q1.enqueued) {
    const bool sw = tn.recog_overlap != 0;
    if (sw) std::swap(stream, recog_stream);
    try { recog_enqueue(q1); } catch (...) { if (sw) std::swap(stream, recog_stream); throw; }
    if (sw) std::swap(stream, recog_stream);
  }
  // (a batch that fails in finish leaves the pipeline: the next flush returns the next batch)
  if (q2.live) { prev_n = q2.n; try { finish(q2, prev_results); } catch (...) { q2.live = false; prev_results.clear(); prev_n = 0; throw; } return; }
  if (q1.live) { prev_n = q1.n; try { finish(q1, prev_results); } catch (...) { q1.live = false; prev_results.clear(); prev_n = 0; throw; } }
}

// ---- image_to_data over a list of host images (engine.h: run_images)
void Engine::run_images(const std::vector<HostImage>& imgs, std::vector<Result>& results, std::vector<int>& failed, std::string& first_error) {
  const int n = (int)imgs.size();
  results.assign(n, Result());
  failed.clear(); first_error.clear();
  if (n == 0) return;
  if (q1.live || q2.live) throw std::runtime_error("streamed batches are in flight: call ttr_stream_flush until it returns none");
  if (comm) throw std::runtime_error("ttr_images_to_data runs on one engine: detach the communicator (every rank takes its own list)");
  std::vector<char> bad(n, 0);
  for (int i = 0; i < n; ++i) {
    const HostImage& im = imgs[i];
    if (!im.data || im.h <= 0 || im.w <= 0 || (im.row_stride >= 0 && im.row_stride < (std::ptrdiff_t)im.w * 3)) {   // tuatara.cpp:344-347: this image yields nothing, the others go on
      std::cerr << "Error reading image from file";
      bad[i] = 1; failed.push_back(i);
      if (first_error.empty()) first_error = "Error reading image from file (image " + std::to_string(i) + ")";
    }
  }
  // buckets of equal (h, w), the largest canvases first (the engine's grow-only workspaces then grow once), cut into batches
  std::map<std::pair<int, int>, std::vector<int>> by_size;
  for (int i = 0; i < n; ++i) if (!bad[i]) by_size[{imgs[i].h, imgs[i].w}].push_back(i);
  std::vector<std::pair<std::pair<int, int>, std::vector<int>>> buckets(by_size.begin(), by_size.end());
  std::stable_sort(buckets.begin(), buckets.end(), [&](const auto& a, const auto& b) {
    const CanvasGeom ga = canvas_geometry(a.first.first, a.first.second, cfg.canvas_size, cfg.mag_ratio), gb = canvas_geometry(b.first.first, b.first.second, cfg.canvas_size, cfg.mag_ratio);
    return (size_t)ga.h32 * ga.w32 > (size_t)gb.h32 * gb.w32;
  });
  struct Batch { int h, w; std::vector<int> idx; };
  std::vector<Batch> batches;
  size_t max_bytes = 0;
  const int cap = std::max(1, tn.images_batch);
  for (auto& b : buckets)
    for (size_t o = 0; o < b.second.size(); o += cap) {
      Batch t{b.first.first, b.first.second, std::vector<int>(b.second.begin() + o, b.second.begin() + std::min(b.second.size(), o + cap))};
      max_bytes = std::max(max_bytes, t.idx.size() * (size_t)t.h * t.w * 3);
      batches.push_back(std::move(t));
    }
  if (!up_stream) {
    TTR_HIP_CHECK(hipStreamCreateWithFlags(&up_stream, hipStreamNonBlocking));
    for (auto& x : up_ev) TTR_HIP_CHECK(hipEventCreateWithFlags(&x, hipEventDisableTiming));
  }
  for (int s = 0; s < kStageSlots && s < (int)batches.size(); ++s) { stage_host[s].ensure(max_bytes); stage_dev[s].ensure(max_bytes); }
  // stage(j): the rows of batch j's images, tightly packed, into pinned slot j % 4; one copy to the device on the upload stream; an event behind it
  std::exception_ptr stage_err;
  auto stage = [&](int j) {
    try {
      TTR_HIP_CHECK(hipSetDevice(cfg.device));
      const Batch& b = batches[j];
      const int sl = j % kStageSlots;
      const size_t page = (size_t)b.h * b.w * 3, row = (size_t)b.w * 3;
      uint8_t* dst = stage_host[sl].as<uint8_t>();
      for (size_t k = 0; k < b.idx.size(); ++k) {
        const HostImage& im = imgs[b.idx[k]];
        if (im.row_stride == (std::ptrdiff_t)row) memcpy(dst + k * page, im.data, page);
        else for (int y = 0; y < b.h; ++y) memcpy(dst + k * page + (size_t)y * row, im.data + (std::ptrdiff_t)y * im.row_stride, row);
      }
      TTR_HIP_CHECK(hipMemcpyAsync(stage_dev[sl].p, dst, b.idx.size() * page, hipMemcpyHostToDevice, up_stream));
      TTR_HIP_CHECK(hipEventRecord(up_ev[sl], up_stream));
    } catch (...) { stage_err = std::current_exception(); }
  };
  const int nb = (int)batches.size();
  std::thread helper;
  struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{helper};
  auto deliver = [&](int j, std::vector<Result>& res, int cnt) {       // batch j's results go to its images' places in the caller's order
    if (cnt != (int)batches[j].idx.size()) throw std::runtime_error("ttr_images_to_data: a batch came back with another page count");
    for (int k = 0; k < cnt; ++k) results[batches[j].idx[k]] = std::move(res[k]);
  };
  auto fail_batch = [&](int j, const std::string& why) {               // a batch that failed on the GPU: its images keep empty results, the list goes on
    for (int i : batches[j].idx) failed.push_back(i);
    if (first_error.empty()) first_error = why + " (images of batch " + std::to_string(j) + ")";
    std::cerr << "tuatara: " << why << std::endl;
  };
  auto drain = [&]() {                                                 // a call-level error mid-list: nothing stays in flight behind it
    std::vector<Result> r; int c = 0;
    for (int guard = 0; guard < 3 && (q1.live || q2.live); ++guard) { try { stream_flush(r, c); } catch (...) { q1 = PageBatch(); q2 = PageBatch(); } }
  };
  if (nb == 0) { std::sort(failed.begin(), failed.end()); return; }
  std::deque<int> inflight;                                            // batches inside the streamed pipeline, oldest first
  stage(0);
  if (stage_err) std::rethrow_exception(stage_err);
  try {
    for (int j = 0; j < nb; ++j) {
      if (helper.joinable()) helper.join();
      if (stage_err) std::rethrow_exception(stage_err);
      if (j + 1 < nb) helper = std::thread(stage, j + 1);               // (slot (j + 1) % 4 last held batch j - 3: returned one push ago at the latest)
      TTR_HIP_CHECK(hipStreamWaitEvent(stream, up_ev[j % kStageSlots], 0));
      std::vector<Result> prev; int np = 0;
      try {
        stream_push(stage_dev[j % kStageSlots].as<uint8_t>(), (int)batches[j].idx.size(), batches[j].h, batches[j].w, prev, np);
        inflight.push_back(j);
        if (np) { deliver(inflight.front(), prev, np); inflight.pop_front(); }
      } catch (const std::exception& ex) {
        if (stream_fail_age == 2 && !inflight.empty()) { fail_batch(inflight.front(), ex.what()); inflight.pop_front(); inflight.push_back(j); }   // the batch whose results were due; j is in
        else fail_batch(j, ex.what());                                                                                                            // batch j's own detector: it never entered
      }
    }
    if (helper.joinable()) helper.join();
    while (!inflight.empty()) {
      std::vector<Result> prev; int np = 0;
      try {
        stream_flush(prev, np);
        if (np) deliver(inflight.front(), prev, np);
        else throw std::runtime_error("ttr_images_to_data: the pipeline ran dry with batches outstanding");
      } catch (const std::exception& ex) { fail_batch(inflight.front(), ex.what()); }
      inflight.pop_front();
    }
  } catch (...) {
    if (helper.joinable()) helper.join();
    drain();
    throw;
  }
  drain();   // (nothing should be left; a batch that failed inside a push may have left a neighbour parked)
  std::sort(failed.begin(), failed.end());
}

}  // namespace ttr
