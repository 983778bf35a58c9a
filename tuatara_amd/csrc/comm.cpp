// Multi-GPU exchange: the transports behind ttr::Transport (RCCL; framed TCP through rank 0), the TCP rendezvous, the ttr_comm_* entry points.
#include "engine.h"

namespace ttr {

struct RcclTransport : Transport {
  ncclComm_t data = nullptr, ctl = nullptr;
  RcclTransport(int rank, int world, const ncclUniqueId ids[2]) {
    TTR_NCCL_CHECK(ncclCommInitRank(&data, world, ids[0], rank));
    TTR_NCCL_CHECK(ncclCommInitRank(&ctl, world, ids[1], rank));
  }
  ~RcclTransport() override {
    if (data) (void)ncclCommDestroy(data);
    if (ctl) (void)ncclCommDestroy(ctl);
  }
  const char* name() const override { return "rccl"; }
  void all_gather(const void* d_send, void* d_recv, size_t bytes, bool control, hipStream_t stream) override {
    TTR_NCCL_CHECK(ncclAllGather(d_send, d_recv, bytes, ncclChar, control ? ctl : data, stream));
  }
  void broadcast(void* d_buf, size_t bytes, int root, hipStream_t stream) override {
    TTR_NCCL_CHECK(ncclBroadcast(d_buf, d_buf, bytes, ncclChar, root, data, stream));
  }
};

// ---- TCP rendezvous: rank 0 listens on addr:port until every other rank has said hello exactly once; strays, duplicates and ranks out of
// range are turned away; every socket has send / receive timeouts and the whole meeting a deadline.
namespace rendezvous {
constexpr uint32_t kMagic = 0x54545243u;   // "TTRC"
struct Hello { uint32_t magic; int32_t rank, world; };
enum : uint32_t { kWelcome = 0, kWrongWorld = 1, kBadRank = 2, kDuplicate = 3 };   // rank 0's answer to a hello
inline void fail(const std::string& m) { throw std::runtime_error("comm rendezvous: " + m + (errno ? std::string(": ") + strerror(errno) : std::string())); }
inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline void set_timeouts(int fd, double seconds) {
  timeval tv; tv.tv_sec = (long)seconds; tv.tv_usec = (long)((seconds - (long)seconds) * 1e6);
  setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
  setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);
  int one = 1; setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
}
inline bool send_all(int fd, const void* p, size_t n) {
  size_t off = 0;
  while (off < n) { const ssize_t w = send(fd, (const char*)p + off, n - off, MSG_NOSIGNAL); if (w <= 0) return false; off += (size_t)w; }
  return true;
}
inline bool recv_all(int fd, void* p, size_t n) {
  size_t off = 0;
  while (off < n) { const ssize_t r = recv(fd, (char*)p + off, n - off, 0); if (r <= 0) return false; off += (size_t)r; }
  return true;
}
inline sockaddr_in resolve(const char* addr, int port) {
  sockaddr_in sa{};
  sa.sin_family = AF_INET; sa.sin_port = htons((uint16_t)port);
  const char* a = (addr && *addr) ? addr : "127.0.0.1";
  if (inet_pton(AF_INET, a, &sa.sin_addr) != 1) {   // a name: getaddrinfo (thread-safe: two engines may set up their communicators from two threads)
    addrinfo hints{}, *res = nullptr;
    hints.ai_family = AF_INET; hints.ai_socktype = SOCK_STREAM;
    const int rc = getaddrinfo(a, nullptr, &hints, &res);
    if (rc != 0 || !res) { errno = 0; fail(std::string("cannot resolve ") + a + (rc ? std::string(": ") + gai_strerror(rc) : std::string())); }
    sa.sin_addr = reinterpret_cast<const sockaddr_in*>(res->ai_addr)->sin_addr;
    freeaddrinfo(res);
  }
  return sa;
}
inline double deadline_seconds() { const char* v = getenv("TUATARA_COMM_TIMEOUT"); const double d = v ? atof(v) : 0.0; return d > 0 ? d : 120.0; }
// rank 0: fds[r] = the connection of rank r (fds[0] = -1).  The listener binds the given address (not INADDR_ANY)
inline std::vector<int> serve(int world, const char* addr, int port, double deadline_s) {
  std::vector<int> fds(world, -1);
  const double t_end = now_s() + deadline_s;
  int ls = socket(AF_INET, SOCK_STREAM, 0);
  if (ls < 0) fail("socket");
  auto close_all = [&]() { for (int& f : fds) if (f >= 0) { close(f); f = -1; } close(ls); };
  int one = 1;
  setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
  sockaddr_in sa = resolve(addr, port);
  while (bind(ls, (sockaddr*)&sa, sizeof(sa)) < 0) {   // (a previous run's listener may still be closing)
    if (errno != EADDRINUSE || now_s() > t_end) { const int e = errno; close(ls); errno = e; fail("bind " + std::string(addr ? addr : "") + ":" + std::to_string(port)); }
    usleep(100000);
  }
  if (listen(ls, world + 8) < 0) { const int e = errno; close(ls); errno = e; fail("listen"); }
  timeval tv{1, 0};
  setsockopt(ls, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);   // accept() wakes up once a second to look at the deadline
  int have = 0;
  while (have < world - 1) {
    if (now_s() > t_end) { close_all(); errno = 0; fail("rank 0 waited " + std::to_string((int)deadline_s) + " s and " + std::to_string(world - 1 - have) + " rank(s) never arrived"); }
    int cs = accept(ls, nullptr, nullptr);
    if (cs < 0) { if (errno == EAGAIN || errno == EWOULDBLOCK || errno == EINTR) continue; const int e = errno; close_all(); errno = e; fail("accept"); }
    set_timeouts(cs, 5.0);
    Hello h{};
    if (!recv_all(cs, &h, sizeof h) || h.magic != kMagic) { close(cs); continue; }   // a stray or a stranger: no answer
    // one of ours: it is told why it is turned away (kWelcome or a reason word) before the socket closes
    const uint32_t why = h.world != world ? kWrongWorld : (h.rank <= 0 || h.rank >= world) ? kBadRank : fds[h.rank] >= 0 ? kDuplicate : kWelcome;
    if (!send_all(cs, &why, 4) || why != kWelcome) { close(cs); continue; }
    set_timeouts(cs, deadline_s);
    fds[h.rank] = cs; ++have;
  }
  close(ls);
  return fds;
}
inline int join(int rank, int world, const char* addr, int port, double deadline_s) {
  const sockaddr_in sa = resolve(addr, port);
  const double t_end = now_s() + deadline_s;
  for (;;) {          // rank 0 may not be listening yet
    int cs = socket(AF_INET, SOCK_STREAM, 0);
    if (cs < 0) fail("socket");
    if (connect(cs, (const sockaddr*)&sa, sizeof(sa)) == 0) {
      set_timeouts(cs, deadline_s);
      const Hello h{kMagic, rank, world};
      if (!send_all(cs, &h, sizeof h)) { const int e = errno; close(cs); errno = e; fail("hello"); }
      uint32_t why = kWelcome;
      if (!recv_all(cs, &why, 4)) { const int e = errno; close(cs); errno = e; fail("no answer to rank " + std::to_string(rank) + "'s hello from " + std::string(addr ? addr : "") + ":" + std::to_string(port) + " (not this job's rank 0?)"); }
      if (why != kWelcome) {
        close(cs); errno = 0;
        fail("rank 0 turned rank " + std::to_string(rank) + " away: " + (why == kWrongWorld ? "it was started with another world size than " + std::to_string(world)
             : why == kBadRank ? "rank out of range for world size " + std::to_string(world) : why == kDuplicate ? "a rank with this number has already joined" : "unknown reason"));
      }
      return cs;
    }
    close(cs);
    if (now_s() > t_end) fail("connect to " + std::string(addr ? addr : "") + ":" + std::to_string(port));
    usleep(20000);
  }
}
}  // namespace rendezvous

struct SocketTransport : Transport {
  int rank, world;
  std::vector<int> fds;      // rank 0: one per peer; else fds[0] = the connection to rank 0
  uint64_t seq = 0;
  PinnedBuf h_send, h_all;
  struct Frame { uint32_t magic; uint32_t kind; uint64_t seq, bytes; };   // kind: 1 all_gather data, 2 all_gather control, 3 broadcast
  SocketTransport(int rank_, int world_, const char* addr, int port) : rank(rank_), world(world_) {
    if (world > 1) {
      if (rank == 0) fds = rendezvous::serve(world, addr, port, rendezvous::deadline_seconds());
      else fds.assign(1, rendezvous::join(rank, world, addr, port, rendezvous::deadline_seconds()));
    }
  }
  ~SocketTransport() override { for (int f : fds) if (f >= 0) close(f); }
  const char* name() const override { return "socket"; }
  void need(bool ok, const char* what) { if (!ok) { throw std::runtime_error(std::string("socket transport: ") + what + " (peer gone, timeout, or a mismatched collective)"); } }
  // every rank announces what it is about to do; rank 0 checks that all announcements agree before any payload moves
  void announce(uint32_t kind, size_t bytes) {
    ++seq;
    const Frame mine{rendezvous::kMagic, kind, seq, (uint64_t)bytes};
    if (rank == 0) {
      bool ok = true; Frame bad{};
      for (int r = 1; r < world; ++r) {
        Frame f{};
        need(rendezvous::recv_all(fds[r], &f, sizeof f), "receiving a frame");
        if (f.magic != mine.magic || f.kind != kind || f.seq != seq || f.bytes != mine.bytes) { ok = false; bad = f; }
      }
      const uint32_t verdict = ok ? 1u : 0u;
      for (int r = 1; r < world; ++r) need(rendezvous::send_all(fds[r], &verdict, 4), "sending the verdict");
      if (!ok) throw std::runtime_error("socket transport: collective mismatch at call " + std::to_string(seq) + ": rank 0 has kind " + std::to_string(kind) + " / " +
                                        std::to_string(bytes) + " bytes, a peer kind " + std::to_string(bad.kind) + " / " + std::to_string(bad.bytes) + " bytes (call " + std::to_string(bad.seq) + ")");
    } else {
      need(rendezvous::send_all(fds[0], &mine, sizeof mine), "sending a frame");
      uint32_t verdict = 0;
      need(rendezvous::recv_all(fds[0], &verdict, 4), "receiving the verdict");
      if (!verdict) throw std::runtime_error("socket transport: collective mismatch at call " + std::to_string(seq) + " (this rank: kind " + std::to_string(kind) + ", " + std::to_string(bytes) + " bytes)");
    }
  }
  void all_gather(const void* d_send, void* d_recv, size_t bytes, bool control, hipStream_t stream) override {
    h_send.ensure(std::max<size_t>(bytes, 1)); h_all.ensure(std::max<size_t>(bytes * world, 1));
    if (bytes) TTR_HIP_CHECK(hipMemcpyAsync(h_send.p, d_send, bytes, hipMemcpyDeviceToHost, stream));
    TTR_HIP_CHECK(hipStreamSynchronize(stream));
    announce(control ? 2u : 1u, bytes);
    char* all = h_all.as<char>();
    if (rank == 0) {
      if (bytes) memcpy(all, h_send.p, bytes);
      for (int r = 1; r < world; ++r) need(bytes == 0 || rendezvous::recv_all(fds[r], all + (size_t)r * bytes, bytes), "gathering");
      for (int r = 1; r < world; ++r) need(bytes == 0 || rendezvous::send_all(fds[r], all, bytes * world), "returning the gather");
    } else {
      need(bytes == 0 || rendezvous::send_all(fds[0], h_send.p, bytes), "contributing");
      need(bytes == 0 || rendezvous::recv_all(fds[0], all, bytes * world), "receiving the gather");
    }
    if (bytes) TTR_HIP_CHECK(hipMemcpyAsync(d_recv, all, bytes * world, hipMemcpyHostToDevice, stream));
    TTR_HIP_CHECK(hipStreamSynchronize(stream));     // (the staging buffer is reused by the next call)
  }
  void broadcast(void* d_buf, size_t bytes, int root, hipStream_t stream) override {
    if (root != 0) throw std::runtime_error("socket transport: broadcast from rank 0 only");
    h_all.ensure(std::max<size_t>(bytes, 1));
    if (rank == 0 && bytes) TTR_HIP_CHECK(hipMemcpyAsync(h_all.p, d_buf, bytes, hipMemcpyDeviceToHost, stream));
    TTR_HIP_CHECK(hipStreamSynchronize(stream));
    announce(3u, bytes);
    if (rank == 0) { for (int r = 1; r < world; ++r) need(bytes == 0 || rendezvous::send_all(fds[r], h_all.p, bytes), "broadcasting"); }
    else {
      need(bytes == 0 || rendezvous::recv_all(fds[0], h_all.p, bytes), "receiving the broadcast");
      if (bytes) TTR_HIP_CHECK(hipMemcpyAsync(d_buf, h_all.p, bytes, hipMemcpyHostToDevice, stream));
      TTR_HIP_CHECK(hipStreamSynchronize(stream));
    }
  }
};

}  // namespace ttr

using namespace ttr;

// rank 0 listens on addr:port and hands its bytes to the world - 1 peers that say hello (rendezvous:: above: each distinct rank once, strays turned away, timeouts)
static void tcp_share(int rank, int world, const char* addr, int port, void* buf, size_t bytes) {
  if (world <= 1) return;
  const double dl = rendezvous::deadline_seconds();
  if (rank == 0) {
    std::vector<int> fds = rendezvous::serve(world, addr, port, dl);
    bool ok = true;
    for (int r = 1; r < world; ++r) { ok = ok && rendezvous::send_all(fds[r], buf, bytes); }
    for (int r = 1; r < world; ++r) close(fds[r]);
    if (!ok) rendezvous::fail("send");
  } else {
    const int fd = rendezvous::join(rank, world, addr, port, dl);
    const bool ok = rendezvous::recv_all(fd, buf, bytes);
    close(fd);
    if (!ok) rendezvous::fail("recv");
  }
}

static ttr_comm* comm_wrap(ttr_engine* e, int rank, int world, std::unique_ptr<Transport> tr) {
  Engine& E = *e->e;
  std::unique_ptr<ttr_comm> h(new ttr_comm());
  h->c.reset(new Comm());
  h->c->rank = rank; h->c->world = world; h->c->E = &E;
  h->c->tr = std::move(tr);
  return h.release();
}

static ttr_comm* comm_create(ttr_engine* e, int rank, int world, const ncclUniqueId ids[2]) {
  if (!e || world < 1 || rank < 0 || rank >= world) throw std::runtime_error("ttr_comm_create: bad arguments");
  EngineScope lk(*e->e);
  return comm_wrap(e, rank, world, std::unique_ptr<Transport>(new RcclTransport(rank, world, ids)));
}

extern "C" {

int ttr_dbg_tcp_share(int rank, int world, const char* addr, int port, void* buf, size_t bytes) {
  TTR_GUARD_BEGIN
  if (!buf || world < 1 || rank < 0 || rank >= world) throw std::runtime_error("bad arguments");
  tcp_share(rank, world, addr, port, buf, bytes);
  return 0;
  TTR_GUARD_END(-1)
}

int ttr_comm_unique_id(void* id256) {
  TTR_GUARD_BEGIN
  if (!id256) throw std::runtime_error("null argument");
  static_assert(sizeof(ncclUniqueId) == 128, "TTR_COMM_ID_BYTES");
  ncclUniqueId ids[2];
  TTR_NCCL_CHECK(ncclGetUniqueId(&ids[0]));
  TTR_NCCL_CHECK(ncclGetUniqueId(&ids[1]));
  memcpy(id256, ids, sizeof(ids));
  return 0;
  TTR_GUARD_END(-1)
}

ttr_comm* ttr_comm_create(ttr_engine* e, int rank, int world, const void* id256) {
  TTR_GUARD_BEGIN
  if (!id256) throw std::runtime_error("null argument");
  ncclUniqueId ids[2];
  memcpy(ids, id256, sizeof(ids));
  return comm_create(e, rank, world, ids);
  TTR_GUARD_END(nullptr)
}

ttr_comm* ttr_comm_create_tcp(ttr_engine* e, int rank, int world, const char* addr, int port) {
  TTR_GUARD_BEGIN
  if (!e || world < 1 || rank < 0 || rank >= world) throw std::runtime_error("ttr_comm_create_tcp: bad arguments");   // before the (blocking) rendezvous
  ncclUniqueId ids[2];
  if (rank == 0) { TTR_NCCL_CHECK(ncclGetUniqueId(&ids[0])); TTR_NCCL_CHECK(ncclGetUniqueId(&ids[1])); }
  tcp_share(rank, world, addr, port, ids, sizeof(ids));
  return comm_create(e, rank, world, ids);
  TTR_GUARD_END(nullptr)
}

// The same communicator over TCP through rank 0 (SocketTransport above): for ranks that share one GPU - RCCL refuses two ranks on a device -
// and as a fallback; every collective is framed and checked, so a mismatched call sequence raises instead of hanging.
ttr_comm* ttr_comm_create_socket(ttr_engine* e, int rank, int world, const char* addr, int port) {
  TTR_GUARD_BEGIN
  if (!e || world < 1 || rank < 0 || rank >= world) throw std::runtime_error("ttr_comm_create_socket: bad arguments");
  EngineScope lk(*e->e);
  return comm_wrap(e, rank, world, std::unique_ptr<Transport>(new SocketTransport(rank, world, addr, port)));
  TTR_GUARD_END(nullptr)
}

const char* ttr_comm_transport(const ttr_comm* c) { return c && c->c && c->c->tr ? c->c->tr->name() : ""; }

int ttr_comm_describe(const ttr_comm* c, char* buf, size_t cap) {
  TTR_GUARD_BEGIN
  if (!c || !c->c || !c->c->E || !buf || cap < 2) throw std::runtime_error("null argument");
  const Engine& E = *c->c->E;
  int ver = 0;
  (void)ncclGetVersion(&ver);
  char bus[64] = "";
  (void)hipDeviceGetPCIBusId(bus, (int)sizeof bus, E.cfg.device);
  hipDeviceProp_t prop{};
  (void)hipGetDeviceProperties(&prop, E.cfg.device);
  char line[512];
  const int n = snprintf(line, sizeof line, "{\"rank\": %d, \"world\": %d, \"transport\": \"%s\", \"rccl_version\": %d, \"device\": %d, \"pci_bus_id\": \"%s\", \"gpu\": \"%s\", \"pid\": %d}",
                         c->c->rank, c->c->world, c->c->tr ? c->c->tr->name() : "", ver, E.cfg.device, bus, prop.gcnArchName, (int)getpid());
  const size_t m = std::min((size_t)std::max(n, 0), cap - 1);
  memcpy(buf, line, m); buf[m] = 0;
  return n;
  TTR_GUARD_END(-1)
}

void ttr_comm_destroy(ttr_comm* c) {
  if (!c) return;
  try {
    if (c->c && c->c->E) {
      Engine& E = *c->c->E;
      EngineScope lk(E);
      if (E.comm == c->c.get()) E.comm = nullptr;
      (void)hipStreamSynchronize(E.stream); (void)hipStreamSynchronize(E.recog_stream); (void)hipStreamSynchronize(E.copy_stream);
      c->c.reset();
    }
  } catch (...) {}
  delete c;
}

int ttr_comm_rank(const ttr_comm* c) { return c && c->c ? c->c->rank : -1; }

int ttr_comm_world(const ttr_comm* c) { return c && c->c ? c->c->world : -1; }

int ttr_engine_attach_comm(ttr_engine* e, ttr_comm* c) {
  TTR_GUARD_BEGIN
  if (!e) throw std::runtime_error("null argument");
  EngineScope lk(*e->e);
  if (e->e->q1.live || e->e->q2.live) throw std::runtime_error("streamed batches are in flight");
  if (c && c->c->E != e->e.get()) throw std::runtime_error("the communicator belongs to another engine");
  e->e->comm = c ? c->c.get() : nullptr;
  return 0;
  TTR_GUARD_END(-1)
}

int ttr_comm_allgather_host(ttr_comm* c, const void* mine, size_t bytes, void* all) {
  TTR_GUARD_BEGIN
  if (!c || !c->c) throw std::runtime_error("null argument");
  Engine& E = *c->c->E;
  EngineScope lk(E);
  Comm* keep = E.comm;
  E.comm = c->c.get();
  try { E.allgather_host(mine, bytes, all); } catch (...) { E.comm = keep; throw; }
  E.comm = keep;
  return 0;
  TTR_GUARD_END(-1)
}

int ttr_last_gathered(ttr_engine* e, int* world, int* pages, int32_t* counts, size_t counts_cap, int32_t* ids, size_t ids_cap, size_t* ids_need) {
  TTR_GUARD_BEGIN
  if (!e) throw std::runtime_error("null argument");
  EngineScope lk(*e->e);
  const auto& g = e->e->last_gathered;
  if (world) *world = g.world;
  if (pages) *pages = g.pages;
  if (ids_need) *ids_need = g.ids.size();
  if (counts && counts_cap >= g.counts.size() && !g.counts.empty()) memcpy(counts, g.counts.data(), g.counts.size() * 4);
  if (ids && ids_cap >= g.ids.size() && !g.ids.empty()) memcpy(ids, g.ids.data(), g.ids.size() * 4);
  return (int)(g.ids.size() / 26);
  TTR_GUARD_END(-1)
}

int ttr_gather_layout(const int32_t* counts, int world, int pages, int* cap, int32_t* total, int64_t* first) {
  TTR_GUARD_BEGIN
  if (!counts || world < 1 || pages < 0) throw std::runtime_error("bad arguments");
  const GatherLayout L = GatherLayout::from_counts(counts, world, pages);
  if (cap) *cap = L.cap;
  if (total) memcpy(total, L.total.data(), (size_t)world * 4);
  if (first) memcpy(first, L.first.data(), L.first.size() * 8);
  return 0;
  TTR_GUARD_END(-1)
}

int ttr_pages_to_data_dev_sharded(ttr_comm* c, const uint8_t* d_pages, int n, int h, int w, ttr_result** out) {
  TTR_GUARD_BEGIN
  if (!c || !c->c || !out) throw std::runtime_error("null argument");
  Engine& E = *c->c->E;
  EngineScope lk(E);
  Comm* keep = E.comm;
  E.comm = c->c.get();
  std::vector<Result> res;
  try { E.run_pages_sharded(d_pages, n, h, w, res); } catch (...) { E.comm = keep; throw; }
  E.comm = keep;
  for (size_t i = 0; i < res.size(); ++i) { out[i] = new ttr_result(); out[i]->r = std::move(res[i]); }
  return (int)res.size();
  TTR_GUARD_END(-1)
}

}  // extern "C"
