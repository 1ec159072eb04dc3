// transport_dora.cpp -- dora-rs adapter (the C++ node API, dora-node-api.h) of the node's transport, the PF twin of the
// reference's crates/rust_robotics/examples/dora_ekf_node.rs:47-125: inputs `odom` and `ranges`, outputs `pose` and `odom_out`,
// payloads = the JSON lines of transport_jsonl.hpp (the reference's dora node moves JSON strings as well, :4-7,62-66).
// SOURCE ONLY in this repository's image (no dora there): it has not met a compiler.
#include <memory>
#include <string>

#include "dora-node-api.h"  // dora's cxx bridge: init_dora_node, next_event, event_type, event_as_input, send_output

#include "transport_jsonl.hpp"

namespace pfnode {

class DoraTransport : public Transport {
 public:
  explicit DoraTransport(Topics topics) : topics_(std::move(topics)), node_(init_dora_node()) {}

  bool wait(Input* out) override {
    for (;;) {
      auto event = node_.events->next();
      const auto ty = event_type(event);
      if (ty == DoraEventType::Stop || ty == DoraEventType::AllInputsClosed) return false;  // dora_ekf_node.rs:102-112
      if (ty != DoraEventType::Input) continue;
      auto input = event_as_input(std::move(event));
      const std::string id(input.id);
      std::string line(reinterpret_cast<const char*>(input.data.data()), input.data.size());
      // the dataflow names the inputs; the JSON payload need not repeat the topic
      Topics by_id = topics_;
      try {
        if (line.find("\"topic\"") == std::string::npos) line.insert(1, "\"topic\": \"" + (id == "ranges" ? topics_.input_ranges : topics_.input_odom) + "\", ");
        if (decode_input(line, by_id, out)) return true;
        log(LogLevel::kWarn, "ignoring unexpected input `" + id + "`");  // dora_ekf_node.rs:100
      } catch (const std::exception& e) {
        log(LogLevel::kWarn, std::string("malformed payload on `") + id + "`: " + e.what());
      }
    }
  }
  void publish(const Output& msg) override {
    const std::string line = encode_output(msg);
    const std::string id = msg.kind == Output::kPose ? "pose" : "odom_out";
    rust::Slice<const uint8_t> data{reinterpret_cast<const uint8_t*>(line.data()), line.size()};
    auto r = send_output(node_.send_output, id, data);
    if (!std::string(r.error).empty()) log(LogLevel::kWarn, "send_output(" + id + "): " + std::string(r.error));
  }
  void log(LogLevel level, const std::string& text) override {
    std::fprintf(stderr, "[pf_localizer_node %s] %s\n", level == LogLevel::kWarn ? "warn" : "info", text.c_str());
  }

 private:
  Topics topics_;
  DoraNode node_;
};

std::unique_ptr<Transport> make_dora_transport(const Topics& topics) { return std::make_unique<DoraTransport>(topics); }

}  // namespace pfnode
