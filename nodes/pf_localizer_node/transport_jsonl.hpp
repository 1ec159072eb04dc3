// transport_jsonl.hpp -- the transport that builds everywhere: one JSON object per line, over stdin/stdout or a UNIX-domain
// stream socket (PF_TRANSPORT=stdio | unix:/path/to.sock).  It is what tests/test_gpu_node.py drives, and the same codec
// carries the payloads of the dora adapter (the reference's dora example moves JSON strings too, dora_ekf_node.rs:4-7).
//
//   in : {"topic": "<odom topic>", "stamp": [sec, nanosec], "frame_id": "odom", "child_frame_id": "base_link",
//         "pose": [x, y, z, qx, qy, qz, qw], "twist": [vx, vy, vz, wx, wy, wz]}            (covariances optional: 36 numbers)
//        {"topic": "<ranges topic>", "stamp": [sec, nanosec], "ranges": [d0, lx0, ly0, d1, lx1, ly1, ...]}
//   out: {"topic": "<pose topic>", "stamp": [..], "frame_id": "..", "pose": [x, y, 0, 0, 0, qz, qw], "state": [x, y, yaw, v]}
//        {"topic": "<odom out topic>", ... "pose": [...], "twist": [...], "state": [...]}
//        {"log": "info" | "warn", "text": "..."}
// Numbers are written with 17 significant digits: a reader gets the doubles back bit for bit.
#pragma once

#include <sys/socket.h>
#include <sys/un.h>
#include <unistd.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "messages.hpp"

namespace pfnode {

// ---- a flat JSON object: string keys -> string | number | array of numbers (all the wire format needs)
struct JsonValue {
  enum Kind { kString, kNumber, kArray } kind = kNumber;
  std::string s;
  double x = 0.0;
  std::vector<double> a;
};
using JsonObject = std::map<std::string, JsonValue>;

class JsonReader {
 public:
  explicit JsonReader(const std::string& text) : p_(text.c_str()) {}
  JsonObject object() {
    JsonObject o;
    ws();
    expect('{');
    ws();
    if (*p_ == '}') return ++p_, o;
    for (;;) {
      ws();
      const std::string key = string();
      ws();
      expect(':');
      ws();
      JsonValue v;
      if (*p_ == '"') {
        v.kind = JsonValue::kString;
        v.s = string();
      } else if (*p_ == '[') {
        v.kind = JsonValue::kArray;
        ++p_;
        ws();
        if (*p_ != ']')
          for (;;) {
            ws();
            v.a.push_back(number());
            ws();
            if (*p_ == ',') {
              ++p_;
              continue;
            }
            break;
          }
        expect(']');
      } else {
        v.kind = JsonValue::kNumber;
        v.x = number();
      }
      o[key] = std::move(v);
      ws();
      if (*p_ == ',') {
        ++p_;
        continue;
      }
      break;
    }
    expect('}');
    return o;
  }

 private:
  void ws() {
    while (*p_ == ' ' || *p_ == '\t' || *p_ == '\r' || *p_ == '\n') ++p_;
  }
  void expect(char c) {
    if (*p_ != c) throw std::runtime_error(std::string("JSON: expected '") + c + "' near \"" + std::string(p_).substr(0, 16) + "\"");
    ++p_;
  }
  std::string string() {
    expect('"');
    std::string s;
    while (*p_ && *p_ != '"') {
      if (*p_ == '\\' && p_[1]) ++p_;  // (topic and frame names need no escapes beyond \" and \\)
      s.push_back(*p_++);
    }
    expect('"');
    return s;
  }
  double number() {
    char* end = nullptr;
    const double x = std::strtod(p_, &end);
    if (end == p_) throw std::runtime_error("JSON: expected a number near \"" + std::string(p_).substr(0, 16) + "\"");
    p_ = end;
    return x;
  }
  const char* p_;
};

inline void json_numbers(std::string* out, const double* v, size_t n) {
  char b[40];
  out->push_back('[');
  for (size_t k = 0; k < n; ++k) {
    std::snprintf(b, sizeof b, k ? ", %.17g" : "%.17g", v[k]);
    out->append(b);
  }
  out->push_back(']');
}
inline std::string json_escape(const std::string& s) {
  std::string r;
  for (char c : s) {
    if (c == '"' || c == '\\') r.push_back('\\');
    r.push_back(c == '\n' ? ' ' : c);
  }
  return r;
}

// decode one input line; returns false for a topic the node did not subscribe to
inline bool decode_input(const std::string& line, const Topics& topics, Input* out) {
  const JsonObject o = JsonReader(line).object();
  auto arr = [&](const char* key, size_t n) -> const std::vector<double>* {
    auto it = o.find(key);
    if (it == o.end()) return nullptr;
    if (it->second.kind != JsonValue::kArray || it->second.a.size() != n)
      throw std::runtime_error(std::string("\"") + key + "\" must be an array of " + std::to_string(n) + " numbers");
    return &it->second.a;
  };
  auto str = [&](const char* key) {
    auto it = o.find(key);
    return it != o.end() && it->second.kind == JsonValue::kString ? it->second.s : std::string();
  };
  const std::string topic = str("topic");
  Stamp stamp;
  if (const auto* st = arr("stamp", 2)) {
    // (a cast of a NaN, negative or huge double is undefined: range first)
    const double sec = (*st)[0], nsec = (*st)[1];
    if (!(sec >= -2147483648.0 && sec <= 2147483647.0) || !(nsec >= 0.0 && nsec < 4294967296.0)) throw std::runtime_error("stamp out of range");
    stamp.sec = (int32_t)sec;
    stamp.nanosec = (uint32_t)nsec;
  }
  if (topic == topics.input_ranges) {
    out->kind = Input::kLandmarkRanges;
    out->ranges.stamp = stamp;
    auto it = o.find("ranges");
    if (it == o.end() || it->second.kind != JsonValue::kArray) throw std::runtime_error("a ranges message needs \"ranges\": [...]");
    out->ranges.ranges = it->second.a;
    return true;
  }
  if (topic != topics.input_odom) return false;
  out->kind = Input::kOdometry;
  Odometry& m = out->odom;
  m = Odometry{};
  m.stamp = stamp;
  m.frame_id = str("frame_id");
  m.child_frame_id = str("child_frame_id");
  if (const auto* p = arr("pose", 7)) {
    for (int k = 0; k < 3; ++k) m.position[k] = (*p)[k];
    for (int k = 0; k < 4; ++k) m.orientation[k] = (*p)[3 + k];
  }
  if (const auto* t = arr("twist", 6)) {
    for (int k = 0; k < 3; ++k) m.linear[k] = (*t)[k];
    for (int k = 0; k < 3; ++k) m.angular[k] = (*t)[3 + k];
  }
  if (const auto* c = arr("pose_covariance", 36)) std::copy(c->begin(), c->end(), m.pose_covariance.begin());
  if (const auto* c = arr("twist_covariance", 36)) std::copy(c->begin(), c->end(), m.twist_covariance.begin());
  return true;
}

inline std::string encode_output(const Output& msg) {
  std::string s = "{\"topic\": \"" + json_escape(msg.topic) + "\", \"stamp\": [";
  const bool pose = msg.kind == Output::kPose;
  const Stamp st = pose ? msg.pose.stamp : msg.odom.stamp;
  s += std::to_string(st.sec) + ", " + std::to_string(st.nanosec) + "], \"frame_id\": \"" +
       json_escape(pose ? msg.pose.frame_id : msg.odom.frame_id) + "\"";
  double p[7];
  const double* pos = pose ? msg.pose.position : msg.odom.position;
  const double* q = pose ? msg.pose.orientation : msg.odom.orientation;
  for (int k = 0; k < 3; ++k) p[k] = pos[k];
  for (int k = 0; k < 4; ++k) p[3 + k] = q[k];
  s += ", \"pose\": ";
  json_numbers(&s, p, 7);
  if (!pose) {
    s += ", \"child_frame_id\": \"" + json_escape(msg.odom.child_frame_id) + "\", \"twist\": ";
    double t[6];
    for (int k = 0; k < 3; ++k) t[k] = msg.odom.linear[k], t[3 + k] = msg.odom.angular[k];
    json_numbers(&s, t, 6);
    s += ", \"pose_covariance\": ";
    json_numbers(&s, msg.odom.pose_covariance.data(), 36);
    s += ", \"twist_covariance\": ";
    json_numbers(&s, msg.odom.twist_covariance.data(), 36);
  }
  s += ", \"state\": ";
  json_numbers(&s, msg.state, 4);
  s += "}\n";
  return s;
}

class JsonLinesTransport : public Transport {
 public:
  // spec: "stdio" or "unix:/path" (the node listens, serves ONE peer and shuts down when it hangs up)
  JsonLinesTransport(const std::string& spec, Topics topics) : topics_(std::move(topics)) {
    if (spec.rfind("unix:", 0) == 0) {
      const std::string path = spec.substr(5);
      listen_fd_ = ::socket(AF_UNIX, SOCK_STREAM, 0);
      if (listen_fd_ < 0) throw std::runtime_error("socket(): " + std::string(std::strerror(errno)));
      sockaddr_un addr{};
      addr.sun_family = AF_UNIX;
      if (path.size() >= sizeof addr.sun_path) throw std::runtime_error("socket path too long");
      std::strcpy(addr.sun_path, path.c_str());
      ::unlink(path.c_str());
      if (::bind(listen_fd_, (sockaddr*)&addr, sizeof addr) != 0 || ::listen(listen_fd_, 1) != 0)
        throw std::runtime_error("bind/listen(" + path + "): " + std::strerror(errno));
      path_ = path;
      log(LogLevel::kInfo, "listening on " + path);
      in_fd_ = out_fd_ = ::accept(listen_fd_, nullptr, nullptr);
      if (in_fd_ < 0) throw std::runtime_error("accept(): " + std::string(std::strerror(errno)));
    } else if (spec == "stdio" || spec.empty()) {
      in_fd_ = 0;
      out_fd_ = 1;
    } else {
      throw std::runtime_error("PF_TRANSPORT must be stdio or unix:/path");
    }
  }
  ~JsonLinesTransport() override {
    if (listen_fd_ >= 0) {
      ::close(in_fd_);
      ::close(listen_fd_);
      ::unlink(path_.c_str());
    }
  }

  bool wait(Input* out) override {
    for (;;) {
      std::string line;
      if (!read_line(&line)) return false;
      if (line.find_first_not_of(" \t\r") == std::string::npos || line[0] == '#') continue;
      try {
        if (decode_input(line, topics_, out)) return true;
        log(LogLevel::kWarn, "ignoring a message of a topic this node does not subscribe to");
      } catch (const std::exception& e) {
        log(LogLevel::kWarn, std::string("malformed message dropped: ") + e.what());
      }
    }
  }
  void publish(const Output& msg) override { write_all(out_fd_, encode_output(msg)); }
  void log(LogLevel level, const std::string& text) override {
    const std::string line = std::string("{\"log\": \"") + (level == LogLevel::kWarn ? "warn" : "info") + "\", \"text\": \"" + json_escape(text) + "\"}\n";
    write_all(in_fd_ == 0 ? 2 : out_fd_, line);  // stdio: the log goes to stderr, the data stream stays clean
  }

 private:
  bool read_line(std::string* line) {
    for (;;) {
      const size_t nl = buf_.find('\n');
      if (nl != std::string::npos) {
        *line = buf_.substr(0, nl);
        buf_.erase(0, nl + 1);
        return true;
      }
      if (buf_.size() > kMaxLineBytes) {  // a peer that never sends a newline must not grow this process without bound
        log(LogLevel::kWarn, "input line longer than 1 MiB dropped");
        buf_.clear();
      }
      char chunk[4096];
      const ssize_t got = ::read(in_fd_, chunk, sizeof chunk);
      if (got < 0 && errno == EINTR) continue;
      if (got <= 0) {
        if (buf_.empty()) return false;
        *line = buf_;
        buf_.clear();
        return true;
      }
      buf_.append(chunk, (size_t)got);
    }
  }
  static constexpr size_t kMaxLineBytes = 1u << 20;
  static void write_all(int fd, const std::string& s) {
    size_t off = 0;
    while (off < s.size()) {
      const ssize_t n = ::write(fd, s.data() + off, s.size() - off);
      if (n < 0 && errno == EINTR) continue;
      if (n <= 0) return;  // the peer is gone: wait() will see it
      off += (size_t)n;
    }
  }
  Topics topics_;
  int in_fd_ = -1, out_fd_ = -1, listen_fd_ = -1;
  std::string path_, buf_;
};

}  // namespace pfnode
