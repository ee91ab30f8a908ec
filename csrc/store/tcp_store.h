// Rendezvous key/value store: one poll()-driven server thread plus blocking clients.
//
// Replaces the c10d TCPStore the reference reaches through init_method="env://"
// (reference distributed.py:65-66).  The server binds port 0 when asked to, so the
// launcher never hands out a port that another process can grab between "find a
// free port" and "bind it" (the race in reference distributed.py:32-37).
//
// Wire format (all integers little-endian):
//   request  : u8 opcode | u32 nkeys | nkeys x (u32 len | bytes) | u32 vlen | value | i64 arg
//   response : u8 status (0 ok, 1 timeout/missing, 2 error) | u32 vlen | value | i64 arg
// A GET / WAIT on a missing key is parked server-side and answered when a SET / ADD
// creates the key; the client enforces its own deadline and reconnects on expiry.
#pragma once

#include <atomic>
#include <chrono>
#include <cstdint>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace dpa {

enum class StoreOp : uint8_t {
  kSet = 1,
  kGet = 2,
  kAdd = 3,
  kWait = 4,
  kCheck = 5,
  kDelete = 6,
  kCompareSet = 7,
  kNumKeys = 8,
  kAppend = 9,
  kPing = 10,
};

class TcpStoreServer {
 public:
  // port == 0 lets the kernel choose; the chosen port is available from port().
  TcpStoreServer(const std::string& bind_addr, int port);
  ~TcpStoreServer();
  TcpStoreServer(const TcpStoreServer&) = delete;
  TcpStoreServer& operator=(const TcpStoreServer&) = delete;

  int port() const { return port_; }
  void shutdown();
  size_t num_keys();

 private:
  struct Conn {
    int fd = -1;
    std::vector<uint8_t> inbuf;
  };
  struct Waiter {
    int fd;
    StoreOp op;
    std::vector<std::string> keys;
  };

  void loop();
  bool handle_frames(Conn& c);
  void handle_request(int fd, StoreOp op, const std::vector<std::string>& keys,
                      const std::string& value, int64_t arg);
  void wake_waiters();
  bool have_all(const std::vector<std::string>& keys) const;
  static void reply(int fd, uint8_t status, const std::string& value, int64_t arg);

  int listen_fd_ = -1;
  int wake_pipe_[2] = {-1, -1};
  int port_ = 0;
  std::atomic<bool> stop_{false};
  std::thread thread_;
  std::mutex mu_;  // guards kv_ for num_keys() callers outside the loop thread
  std::unordered_map<std::string, std::string> kv_;
  std::map<int, Conn> conns_;
  std::vector<Waiter> waiters_;
};

class TcpStoreClient {
 public:
  TcpStoreClient(const std::string& host, int port, double timeout_s);
  ~TcpStoreClient();
  TcpStoreClient(const TcpStoreClient&) = delete;
  TcpStoreClient& operator=(const TcpStoreClient&) = delete;

  void set(const std::string& key, const std::string& value);
  // Blocks until the key exists or the timeout expires (throws std::runtime_error).
  std::string get(const std::string& key, double timeout_s = -1.0);
  int64_t add(const std::string& key, int64_t delta);
  void wait(const std::vector<std::string>& keys, double timeout_s = -1.0);
  bool check(const std::vector<std::string>& keys);
  bool delete_key(const std::string& key);
  std::string compare_set(const std::string& key, const std::string& expected,
                          const std::string& desired);
  int64_t num_keys();
  void append(const std::string& key, const std::string& value);
  void ping();
  // Counter-based barrier over `world` participants; reusable (generation counter).
  void barrier(const std::string& name, int world, double timeout_s = -1.0);

  double timeout() const { return timeout_s_; }
  void set_timeout(double t) { timeout_s_ = t; }
  const std::string& host() const { return host_; }
  int port() const { return port_; }

 private:
  struct Response {
    uint8_t status;
    std::string value;
    int64_t arg;
  };
  void connect_with_retry();
  void close_fd();
  Response call(StoreOp op, const std::vector<std::string>& keys, const std::string& value,
                int64_t arg, double timeout_s);

  std::string host_;
  int port_;
  double timeout_s_;
  int fd_ = -1;
  std::mutex mu_;
  std::unordered_map<std::string, int64_t> barrier_gen_;
};

}  // namespace dpa
