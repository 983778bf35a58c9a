// Host-only pieces of the engine (no HIP): the .ttrw weight-file reader and the persistent host-thread pool.  Kept apart from
// engine.cpp so that the CPU test suite can compile them - together with geometry.cpp and examples/png_decode.h - under
// AddressSanitizer / UBSan / ThreadSanitizer (tests/native/host_san.cpp, tests/test_sanitizers_cpu.py): the weight file and the PNG
// reader parse bytes a caller hands over, and the pool replaces the reference's ad-hoc thread fan-out (tuatara.cpp:461-475).
#pragma once
#include <cmath>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <exception>
#include <fstream>
#include <functional>
#include <iterator>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace ttr {

struct HostTensor { std::vector<uint32_t> dims; std::vector<float> data; };

struct WeightFile {
  std::map<std::string, HostTensor> t;
  explicit WeightFile(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open weight file " + path);
    std::vector<char> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (buf.size() < 12 || memcmp(buf.data(), "TTRW0001", 8) != 0) throw std::runtime_error("not a .ttrw file: " + path);
    size_t p = 8;
    auto rd = [&](void* dst, size_t n) { if (p + n > buf.size()) throw std::runtime_error("truncated .ttrw: " + path); memcpy(dst, buf.data() + p, n); p += n; };
    uint32_t n; rd(&n, 4);
    if ((uint64_t)n * 20 > buf.size()) throw std::runtime_error("corrupt .ttrw (tensor count): " + path);   // an entry takes >= 20 bytes of table
    struct Ent { std::string name; std::vector<uint32_t> dims; uint64_t off, nb; };
    std::vector<Ent> ents(n);
    for (auto& e : ents) {
      uint16_t ln; rd(&ln, 2);
      e.name.resize(ln); rd(&e.name[0], ln);
      uint8_t dt, nd; rd(&dt, 1); rd(&nd, 1);
      if (dt != 0) throw std::runtime_error("unsupported dtype in " + path);
      e.dims.resize(nd); rd(e.dims.data(), 4 * nd);
      rd(&e.off, 8); rd(&e.nb, 8);
    }
    uint64_t data0; rd(&data0, 8);
    if (data0 > buf.size()) throw std::runtime_error("corrupt .ttrw (data offset): " + path);
    const uint64_t room = buf.size() - data0;
    for (auto& e : ents) {
      uint64_t numel = 1;
      for (uint32_t d : e.dims) { if (d && numel > (uint64_t)1 << 40) throw std::runtime_error("corrupt .ttrw (dims): " + path); numel *= d; }
      if (e.nb % 4 != 0 || e.nb != 4 * numel) throw std::runtime_error("corrupt .ttrw (byte count of " + e.name + "): " + path);
      if (e.off > room || e.nb > room - e.off) throw std::runtime_error("tensor out of range in " + path);   // overflow-safe
      HostTensor ht; ht.dims = e.dims; ht.data.resize(e.nb / 4);
      memcpy(ht.data.data(), buf.data() + data0 + e.off, e.nb);
      // a NaN or an infinity in a weight poisons every product it meets, and the split-operand mode's range guard (split.h) only sees magnitudes:
      // such a file is refused here, by name
      for (const float v : ht.data) if (!std::isfinite(v)) throw std::runtime_error("non-finite value in weight tensor " + e.name + " of " + path);
      t[e.name] = std::move(ht);
    }
  }
  const HostTensor& get(const std::string& name, size_t numel) const {
    auto it = t.find(name);
    if (it == t.end()) throw std::runtime_error("weight tensor missing: " + name);
    if (it->second.data.size() != numel) throw std::runtime_error("weight tensor has wrong size: " + name);
    return it->second;
  }
};

// A few persistent host threads for the per-page host work (calipers, token decode): spawning threads per batch cost more
// than the work itself.  run(n, f) calls f(0..n-1) across the workers and the caller; the first exception is rethrown.
class HostPool {
 public:
  explicit HostPool(int workers) {
    for (int t = 0; t < workers; ++t) th_.emplace_back([this] { loop(); });
  }
  ~HostPool() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; ++gen_; }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  void run(int n, const std::function<void(int)>& f) {
    if (n <= 0) return;
    if (n == 1 || th_.empty()) { for (int i = 0; i < n; ++i) f(i); return; }
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &f; n_ = n; next_.store(0); pending_ = n; err_ = nullptr; ++gen_;
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [this] { return pending_ == 0 && busy_ == 0; });
    fn_ = nullptr;
    if (err_) std::rethrow_exception(err_);
  }

 private:
  void work() {
    int finished = 0;
    std::exception_ptr err;
    for (;;) {
      const int i = next_.fetch_add(1);
      if (i >= n_) break;
      try { (*fn_)(i); } catch (...) { if (!err) err = std::current_exception(); }
      ++finished;
    }
    if (finished || err) {
      std::lock_guard<std::mutex> lk(mu_);
      pending_ -= finished;
      if (err && !err_) err_ = err;
    }
  }
  void loop() {
    unsigned long long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
        if (!fn_) continue;
        ++busy_;
      }
      work();
      {
        std::lock_guard<std::mutex> lk(mu_);
        --busy_;
        if (pending_ == 0 && busy_ == 0) done_.notify_all();
      }
    }
  }
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable cv_, done_;
  const std::function<void(int)>* fn_ = nullptr;
  std::atomic<int> next_{0};
  int n_ = 0, pending_ = 0, busy_ = 0;
  unsigned long long gen_ = 0;
  bool stop_ = false;
  std::exception_ptr err_;
};

}  // namespace ttr
