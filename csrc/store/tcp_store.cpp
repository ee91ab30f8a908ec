#include "tcp_store.h"

#include <arpa/inet.h>
#include <errno.h>
#include <fcntl.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <cstring>
#include <stdexcept>

namespace dpa {

namespace {

using Clock = std::chrono::steady_clock;

void put_u32(std::vector<uint8_t>& b, uint32_t v) {
  for (int i = 0; i < 4; ++i) b.push_back(static_cast<uint8_t>(v >> (8 * i)));
}
void put_i64(std::vector<uint8_t>& b, int64_t v) {
  uint64_t u = static_cast<uint64_t>(v);
  for (int i = 0; i < 8; ++i) b.push_back(static_cast<uint8_t>(u >> (8 * i)));
}
void put_bytes(std::vector<uint8_t>& b, const std::string& s) {
  put_u32(b, static_cast<uint32_t>(s.size()));
  b.insert(b.end(), s.begin(), s.end());
}
uint32_t rd_u32(const uint8_t* p) {
  return uint32_t(p[0]) | (uint32_t(p[1]) << 8) | (uint32_t(p[2]) << 16) | (uint32_t(p[3]) << 24);
}
int64_t rd_i64(const uint8_t* p) {
  uint64_t u = 0;
  for (int i = 0; i < 8; ++i) u |= uint64_t(p[i]) << (8 * i);
  return static_cast<int64_t>(u);
}

bool send_all(int fd, const uint8_t* data, size_t n) {
  while (n > 0) {
    ssize_t w = ::send(fd, data, n, MSG_NOSIGNAL);
    if (w < 0) {
      if (errno == EINTR) continue;
      if (errno == EAGAIN || errno == EWOULDBLOCK) {
        pollfd p{fd, POLLOUT, 0};
        ::poll(&p, 1, 1000);
        continue;
      }
      return false;
    }
    data += w;
    n -= static_cast<size_t>(w);
  }
  return true;
}

void set_nodelay(int fd) {
  int one = 1;
  ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
}

// Parses one request frame starting at buf[0]. Returns bytes consumed, 0 if incomplete.
size_t parse_request(const std::vector<uint8_t>& buf, StoreOp* op, std::vector<std::string>* keys,
                     std::string* value, int64_t* arg) {
  const size_t n = buf.size();
  size_t off = 0;
  if (n < 5) return 0;
  *op = static_cast<StoreOp>(buf[0]);
  uint32_t nkeys = rd_u32(&buf[1]);
  off = 5;
  keys->clear();
  for (uint32_t i = 0; i < nkeys; ++i) {
    if (n < off + 4) return 0;
    uint32_t len = rd_u32(&buf[off]);
    off += 4;
    if (n < off + len) return 0;
    keys->emplace_back(reinterpret_cast<const char*>(&buf[off]), len);
    off += len;
  }
  if (n < off + 4) return 0;
  uint32_t vlen = rd_u32(&buf[off]);
  off += 4;
  if (n < off + vlen + 8) return 0;
  value->assign(reinterpret_cast<const char*>(&buf[off]), vlen);
  off += vlen;
  *arg = rd_i64(&buf[off]);
  off += 8;
  return off;
}

}  // namespace

// ----------------------------------------------------------------------------- server

TcpStoreServer::TcpStoreServer(const std::string& bind_addr, int port) {
  listen_fd_ = ::socket(AF_INET, SOCK_STREAM, 0);
  if (listen_fd_ < 0) throw std::runtime_error("TcpStoreServer: socket() failed");
  int one = 1;
  // Before bind, so a server restarted on a fixed port is not refused while the old
  // connections sit in TIME_WAIT.
  ::setsockopt(listen_fd_, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
  sockaddr_in addr{};
  addr.sin_family = AF_INET;
  addr.sin_port = htons(static_cast<uint16_t>(port));
  if (bind_addr.empty() || bind_addr == "*" || bind_addr == "0.0.0.0") {
    addr.sin_addr.s_addr = htonl(INADDR_ANY);
  } else if (bind_addr == "localhost") {
    addr.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
  } else if (::inet_pton(AF_INET, bind_addr.c_str(), &addr.sin_addr) != 1) {
    addr.sin_addr.s_addr = htonl(INADDR_ANY);
  }
  if (::bind(listen_fd_, reinterpret_cast<sockaddr*>(&addr), sizeof(addr)) != 0) {
    int e = errno;
    ::close(listen_fd_);
    listen_fd_ = -1;
    throw std::runtime_error(std::string("TcpStoreServer: bind failed: ") + std::strerror(e));
  }
  if (::listen(listen_fd_, 512) != 0) {
    ::close(listen_fd_);
    listen_fd_ = -1;
    throw std::runtime_error("TcpStoreServer: listen failed");
  }
  socklen_t len = sizeof(addr);
  ::getsockname(listen_fd_, reinterpret_cast<sockaddr*>(&addr), &len);
  port_ = ntohs(addr.sin_port);
  if (::pipe(wake_pipe_) != 0) throw std::runtime_error("TcpStoreServer: pipe failed");
  ::fcntl(listen_fd_, F_SETFL, ::fcntl(listen_fd_, F_GETFL, 0) | O_NONBLOCK);
  thread_ = std::thread([this] { loop(); });
}

TcpStoreServer::~TcpStoreServer() { shutdown(); }

void TcpStoreServer::shutdown() {
  bool expected = false;
  if (!stop_.compare_exchange_strong(expected, true)) {
    if (thread_.joinable()) thread_.join();
    return;
  }
  if (wake_pipe_[1] >= 0) {
    char c = 'x';
    ssize_t r = ::write(wake_pipe_[1], &c, 1);
    (void)r;
  }
  if (thread_.joinable()) thread_.join();
  for (auto& kv : conns_) ::close(kv.second.fd);
  conns_.clear();
  if (listen_fd_ >= 0) ::close(listen_fd_);
  listen_fd_ = -1;
  for (int& fd : wake_pipe_) {
    if (fd >= 0) ::close(fd);
    fd = -1;
  }
}

size_t TcpStoreServer::num_keys() {
  std::lock_guard<std::mutex> g(mu_);
  return kv_.size();
}

void TcpStoreServer::reply(int fd, uint8_t status, const std::string& value, int64_t arg) {
  std::vector<uint8_t> out;
  out.reserve(13 + value.size());
  out.push_back(status);
  put_bytes(out, value);
  put_i64(out, arg);
  send_all(fd, out.data(), out.size());
}

bool TcpStoreServer::have_all(const std::vector<std::string>& keys) const {
  for (const auto& k : keys)
    if (kv_.find(k) == kv_.end()) return false;
  return true;
}

void TcpStoreServer::wake_waiters() {
  for (size_t i = 0; i < waiters_.size();) {
    Waiter& w = waiters_[i];
    if (have_all(w.keys)) {
      if (w.op == StoreOp::kGet)
        reply(w.fd, 0, kv_[w.keys[0]], 0);
      else
        reply(w.fd, 0, "", 0);
      waiters_.erase(waiters_.begin() + static_cast<long>(i));
    } else {
      ++i;
    }
  }
}

void TcpStoreServer::handle_request(int fd, StoreOp op, const std::vector<std::string>& keys,
                                    const std::string& value, int64_t arg) {
  std::lock_guard<std::mutex> g(mu_);
  switch (op) {
    case StoreOp::kSet:
      if (keys.size() != 1) return reply(fd, 2, "set: one key expected", 0);
      kv_[keys[0]] = value;
      reply(fd, 0, "", 0);
      wake_waiters();
      break;
    case StoreOp::kAppend:
      if (keys.size() != 1) return reply(fd, 2, "append: one key expected", 0);
      kv_[keys[0]] += value;
      reply(fd, 0, "", 0);
      wake_waiters();
      break;
    case StoreOp::kGet:
      if (keys.size() != 1) return reply(fd, 2, "get: one key expected", 0);
      if (kv_.count(keys[0]))
        reply(fd, 0, kv_[keys[0]], 0);
      else
        waiters_.push_back(Waiter{fd, op, keys});
      break;
    case StoreOp::kWait:
      if (have_all(keys))
        reply(fd, 0, "", 0);
      else
        waiters_.push_back(Waiter{fd, op, keys});
      break;
    case StoreOp::kAdd: {
      if (keys.size() != 1) return reply(fd, 2, "add: one key expected", 0);
      int64_t cur = 0;
      auto it = kv_.find(keys[0]);
      if (it != kv_.end()) {
        try {
          cur = std::stoll(it->second);
        } catch (...) {
          return reply(fd, 2, "add: value is not an integer", 0);
        }
      }
      cur += arg;
      kv_[keys[0]] = std::to_string(cur);
      reply(fd, 0, "", cur);
      wake_waiters();
      break;
    }
    case StoreOp::kCheck:
      reply(fd, 0, "", have_all(keys) ? 1 : 0);
      break;
    case StoreOp::kDelete: {
      int64_t n = 0;
      for (const auto& k : keys) n += static_cast<int64_t>(kv_.erase(k));
      reply(fd, 0, "", n);
      break;
    }
    case StoreOp::kCompareSet: {
      // keys[0] = key, keys[1] = expected; value = desired.  An absent key matches an
      // empty `expected`, mirroring c10d Store::compareSet.
      if (keys.size() != 2) return reply(fd, 2, "compare_set: key and expected value", 0);
      auto it = kv_.find(keys[0]);
      if (it == kv_.end()) {
        if (keys[1].empty()) {
          kv_[keys[0]] = value;
          reply(fd, 0, value, 0);
          wake_waiters();
        } else {
          reply(fd, 0, keys[1], 0);
        }
      } else {
        if (it->second == keys[1]) it->second = value;
        reply(fd, 0, it->second, 0);
      }
      break;
    }
    case StoreOp::kNumKeys:
      reply(fd, 0, "", static_cast<int64_t>(kv_.size()));
      break;
    case StoreOp::kPing:
      reply(fd, 0, "", arg);
      break;
    default:
      reply(fd, 2, "unknown opcode", 0);
  }
}

bool TcpStoreServer::handle_frames(Conn& c) {
  for (;;) {
    StoreOp op;
    std::vector<std::string> keys;
    std::string value;
    int64_t arg = 0;
    size_t used = parse_request(c.inbuf, &op, &keys, &value, &arg);
    if (used == 0) return true;
    c.inbuf.erase(c.inbuf.begin(), c.inbuf.begin() + static_cast<long>(used));
    handle_request(c.fd, op, keys, value, arg);
  }
}

void TcpStoreServer::loop() {
  std::vector<pollfd> pfds;
  while (!stop_.load()) {
    pfds.clear();
    pfds.push_back({listen_fd_, POLLIN, 0});
    pfds.push_back({wake_pipe_[0], POLLIN, 0});
    for (auto& kv : conns_) pfds.push_back({kv.first, POLLIN, 0});
    int rc = ::poll(pfds.data(), pfds.size(), 500);
    if (rc < 0) {
      if (errno == EINTR) continue;
      break;
    }
    if (rc == 0) continue;
    if (pfds[1].revents & POLLIN) break;
    if (pfds[0].revents & POLLIN) {
      for (;;) {
        int fd = ::accept(listen_fd_, nullptr, nullptr);
        if (fd < 0) break;
        set_nodelay(fd);
        Conn c;
        c.fd = fd;
        conns_[fd] = std::move(c);
      }
    }
    for (size_t i = 2; i < pfds.size(); ++i) {
      if (!(pfds[i].revents & (POLLIN | POLLHUP | POLLERR))) continue;
      int fd = pfds[i].fd;
      auto it = conns_.find(fd);
      if (it == conns_.end()) continue;
      uint8_t tmp[65536];
      ssize_t r = ::recv(fd, tmp, sizeof(tmp), 0);
      if (r <= 0) {
        if (r < 0 && (errno == EINTR || errno == EAGAIN)) continue;
        // Peer went away: forget anything it was waiting for.
        waiters_.erase(std::remove_if(waiters_.begin(), waiters_.end(),
                                      [fd](const Waiter& w) { return w.fd == fd; }),
                       waiters_.end());
        ::close(fd);
        conns_.erase(it);
        continue;
      }
      it->second.inbuf.insert(it->second.inbuf.end(), tmp, tmp + r);
      handle_frames(it->second);
    }
  }
}

// ----------------------------------------------------------------------------- client

TcpStoreClient::TcpStoreClient(const std::string& host, int port, double timeout_s)
    : host_(host), port_(port), timeout_s_(timeout_s) {
  connect_with_retry();
}

TcpStoreClient::~TcpStoreClient() { close_fd(); }

void TcpStoreClient::close_fd() {
  if (fd_ >= 0) ::close(fd_);
  fd_ = -1;
}

void TcpStoreClient::connect_with_retry() {
  auto deadline = Clock::now() + std::chrono::duration<double>(timeout_s_ > 0 ? timeout_s_ : 300.0);
  std::string last_err = "unknown";
  int backoff_ms = 10;
  for (;;) {
    addrinfo hints{};
    hints.ai_family = AF_INET;
    hints.ai_socktype = SOCK_STREAM;
    addrinfo* res = nullptr;
    std::string h = host_ == "localhost" ? "127.0.0.1" : host_;
    int gai = ::getaddrinfo(h.c_str(), std::to_string(port_).c_str(), &hints, &res);
    if (gai == 0 && res != nullptr) {
      int fd = ::socket(res->ai_family, res->ai_socktype, res->ai_protocol);
      if (fd >= 0) {
        if (::connect(fd, res->ai_addr, res->ai_addrlen) == 0) {
          ::freeaddrinfo(res);
          set_nodelay(fd);
          fd_ = fd;
          return;
        }
        last_err = std::strerror(errno);
        ::close(fd);
      }
      ::freeaddrinfo(res);
    } else {
      last_err = ::gai_strerror(gai);
    }
    if (Clock::now() >= deadline)
      throw std::runtime_error("TcpStoreClient: cannot connect to " + host_ + ":" +
                               std::to_string(port_) + " (" + last_err + ")");
    std::this_thread::sleep_for(std::chrono::milliseconds(backoff_ms));
    backoff_ms = std::min(backoff_ms * 2, 500);
  }
}

TcpStoreClient::Response TcpStoreClient::call(StoreOp op, const std::vector<std::string>& keys,
                                              const std::string& value, int64_t arg,
                                              double timeout_s) {
  std::lock_guard<std::mutex> g(mu_);
  if (fd_ < 0) connect_with_retry();
  std::vector<uint8_t> out;
  out.push_back(static_cast<uint8_t>(op));
  put_u32(out, static_cast<uint32_t>(keys.size()));
  for (const auto& k : keys) put_bytes(out, k);
  put_bytes(out, value);
  put_i64(out, arg);
  if (!send_all(fd_, out.data(), out.size())) {
    close_fd();
    throw std::runtime_error("TcpStoreClient: send failed (server gone?)");
  }
  double t = timeout_s >= 0 ? timeout_s : timeout_s_;
  auto deadline = Clock::now() + std::chrono::duration<double>(t > 0 ? t : 1e9);
  std::vector<uint8_t> in;
  for (;;) {
    if (in.size() >= 5) {
      uint32_t vlen = rd_u32(&in[1]);
      if (in.size() >= 5 + size_t(vlen) + 8) {
        Response r;
        r.status = in[0];
        r.value.assign(reinterpret_cast<const char*>(&in[5]), vlen);
        r.arg = rd_i64(&in[5 + vlen]);
        if (r.status == 2) throw std::runtime_error("TcpStore: server error: " + r.value);
        return r;
      }
    }
    auto now = Clock::now();
    if (now >= deadline) {
      // The request may still be parked on the server; drop the connection so the
      // late answer cannot be mistaken for the reply to a later request.
      close_fd();
      std::string what = keys.empty() ? std::string("<none>") : keys[0];
      throw std::runtime_error("TcpStore: timed out after " + std::to_string(t) +
                               " s waiting for key '" + what + "'");
    }
    int ms = static_cast<int>(std::min<double>(
        500.0, std::chrono::duration<double, std::milli>(deadline - now).count() + 1));
    pollfd p{fd_, POLLIN, 0};
    int rc = ::poll(&p, 1, ms);
    if (rc < 0 && errno != EINTR) {
      close_fd();
      throw std::runtime_error("TcpStoreClient: poll failed");
    }
    if (rc <= 0) continue;
    uint8_t tmp[65536];
    ssize_t r = ::recv(fd_, tmp, sizeof(tmp), 0);
    if (r == 0 || (r < 0 && errno != EINTR && errno != EAGAIN)) {
      close_fd();
      throw std::runtime_error("TcpStoreClient: connection closed by server");
    }
    if (r > 0) in.insert(in.end(), tmp, tmp + r);
  }
}

void TcpStoreClient::set(const std::string& key, const std::string& value) {
  call(StoreOp::kSet, {key}, value, 0, -1);
}
std::string TcpStoreClient::get(const std::string& key, double timeout_s) {
  return call(StoreOp::kGet, {key}, "", 0, timeout_s).value;
}
int64_t TcpStoreClient::add(const std::string& key, int64_t delta) {
  return call(StoreOp::kAdd, {key}, "", delta, -1).arg;
}
void TcpStoreClient::wait(const std::vector<std::string>& keys, double timeout_s) {
  call(StoreOp::kWait, keys, "", 0, timeout_s);
}
bool TcpStoreClient::check(const std::vector<std::string>& keys) {
  return call(StoreOp::kCheck, keys, "", 0, -1).arg != 0;
}
bool TcpStoreClient::delete_key(const std::string& key) {
  return call(StoreOp::kDelete, {key}, "", 0, -1).arg > 0;
}
std::string TcpStoreClient::compare_set(const std::string& key, const std::string& expected,
                                        const std::string& desired) {
  return call(StoreOp::kCompareSet, {key, expected}, desired, 0, -1).value;
}
int64_t TcpStoreClient::num_keys() { return call(StoreOp::kNumKeys, {}, "", 0, -1).arg; }
void TcpStoreClient::append(const std::string& key, const std::string& value) {
  call(StoreOp::kAppend, {key}, value, 0, -1);
}
void TcpStoreClient::ping() { call(StoreOp::kPing, {}, "", 42, -1); }

void TcpStoreClient::barrier(const std::string& name, int world, double timeout_s) {
  int64_t gen;
  {
    std::lock_guard<std::mutex> g(mu_);
    gen = barrier_gen_[name]++;
  }
  const std::string base = "__barrier__/" + name + "/" + std::to_string(gen);
  int64_t arrived = add(base + "/count", 1);
  if (arrived == world) set(base + "/done", "1");
  wait({base + "/done"}, timeout_s);
}

}  // namespace dpa
