"""Drives nodes/pf_localizer_node over its JSON-lines transport (stdio or a UNIX socket): the feeder a test or a latency
measurement needs.  The scenario is the reference's particle-filter demo, crates/rust_robotics/examples/
render_gif_particle_filter.rs:33-98: five range-only landmarks, 150 particles, 300 steps of a rounded rectangle."""
import json
import math
import os
import socket
import subprocess
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = os.path.join(ROOT, "nodes", "pf_localizer_node", "pf_localizer_node")

LANDMARKS = [(2.0, 2.0), (10.0, 2.0), (2.0, 8.0), (10.0, 8.0), (6.0, 5.0)]  # render_gif_particle_filter.rs:25-31
DT, STEPS = 0.1, 300                                                         # :16-17
FILTER_ENV = {"PF_PARTICLES": "150", "PF_DT": "0.1", "PF_RANGE_NOISE": "0.25"}  # :35-40 (the rest: ParticleFilterConfig::default())
INITIAL = (5.0, 5.0, 0.0, 0.0)                                               # :34


def scenario(steps=STEPS, seed=42):
    """(controls [steps x 2], observations [steps x 5 x 3], truth [steps x 3]) -- :56-78 with numpy's generator standing in
    for StdRng::seed_from_u64(42) + Normal(0, 0.15) (the reference's RNG stream is not reproducible outside Rust)."""
    rng = np.random.default_rng(seed)
    truth = np.array([5.0, 5.0, 0.0])
    u, obs, tr = [], [], []
    for k in range(steps):
        c = (1.1, 0.0) if (k // 25) % 2 == 0 else (0.5, 0.63)
        truth = truth + [c[0] * math.cos(truth[2]) * DT, c[0] * math.sin(truth[2]) * DT, c[1] * DT]
        obs.append([(max(math.hypot(truth[0] - lx, truth[1] - ly) + rng.normal(0.0, 0.15), 0.0), lx, ly) for lx, ly in LANDMARKS])
        u.append(c)
        tr.append(truth.copy())
    return np.array(u), np.array(obs), np.array(tr)


def odom_line(topic, k, pose_xyyaw, v, omega, frame="odom", child="base_link"):
    x, y, yaw = pose_xyyaw
    ns = int(round(k * DT * 1e9))
    return json.dumps({"topic": topic, "stamp": [ns // 10**9, ns % 10**9], "frame_id": frame, "child_frame_id": child,
                       "pose": [x, y, 0.0, 0.0, 0.0, math.sin(0.5 * yaw), math.cos(0.5 * yaw)], "twist": [v, 0.0, 0.0, 0.0, 0.0, omega]})


def ranges_line(topic, k, rows):
    ns = int(round(k * DT * 1e9))
    return json.dumps({"topic": topic, "stamp": [ns // 10**9, ns % 10**9], "ranges": [float(v) for r in rows for v in r]})


class NodeProcess:
    """The node as a child process.  transport: 'stdio' or 'unix' (a socket in tmpdir)."""

    def __init__(self, env=None, transport="stdio", tmpdir=None):
        e = dict(os.environ)
        e.update(FILTER_ENV)
        e.update(env or {})
        self.sock = None
        if transport == "unix":
            self.path = os.path.join(tmpdir, "pfnode.sock")
            e["PF_TRANSPORT"] = "unix:" + self.path
            self.proc = subprocess.Popen([NODE], env=e, stdin=subprocess.DEVNULL, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            deadline = time.time() + 120
            while True:
                try:
                    self.sock = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                    self.sock.connect(self.path)
                    break
                except OSError:
                    self.sock.close()
                    if time.time() > deadline or self.proc.poll() is not None:
                        raise RuntimeError("the node never listened: " + (self.proc.stderr.read().decode() if self.proc.poll() is not None else "timeout"))
                    time.sleep(0.05)
            self.rfile = self.sock.makefile("rb")
            self.wfile = self.sock.makefile("wb")
        else:
            self.proc = subprocess.Popen([NODE], env=e, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            self.rfile, self.wfile = self.proc.stdout, self.proc.stdin
        self.logs = []

    def send(self, *lines):
        self.wfile.write(("\n".join(lines) + "\n").encode())
        self.wfile.flush()

    def recv(self):
        """next data message (log lines of the socket transport are collected in self.logs)"""
        while True:
            line = self.rfile.readline()
            if not line:
                raise EOFError("the node hung up: " + self.stderr_text())
            m = json.loads(line)
            if "log" in m:
                self.logs.append(m)
                continue
            return m

    def stderr_text(self):
        try:
            return self.proc.stderr.read().decode() if self.proc.poll() is not None else ""
        except Exception:
            return ""

    def close(self, timeout=60):
        try:
            self.wfile.close()
            if self.sock is not None:
                self.sock.shutdown(socket.SHUT_WR)
        except OSError:
            pass
        if self.sock is not None:  # what the node still says on its way out (the closing statistics line) comes through the socket
            try:
                while True:
                    line = self.rfile.readline()
                    if not line:
                        break
                    m = json.loads(line)
                    if "log" in m:
                        self.logs.append(m)
            except (OSError, ValueError) as e:  # kept in the log text, so that a failing assertion on it says why
                self.logs.append({"log": "driver", "text": f"reading the node's last words failed: {e!r}"})
        rc = self.proc.wait(timeout=timeout)
        err = self.proc.stderr.read().decode()
        if self.sock is not None:
            self.rfile.close()
            self.sock.close()
        return rc, err


def run_scenario(node, steps=STEPS, seed=42, topics=("/odom", "/landmark_ranges")):
    """Feed the scenario; returns (states [steps+1 x 4] from the pose messages, round-trip microseconds per step [steps], all messages)."""
    u, obs, truth = scenario(steps, seed)
    msgs = []
    node.send(odom_line(topics[0], 0, INITIAL[:3], INITIAL[3], 0.0))
    first = [node.recv(), node.recv()]
    msgs += first
    states, lat = [first[0]["state"]], []
    for k in range(steps):
        a = ranges_line(topics[1], k + 1, obs[k])
        b = odom_line(topics[0], k + 1, truth[k], u[k][0], u[k][1])
        t0 = time.perf_counter_ns()
        node.send(a, b)
        pose, odom = node.recv(), node.recv()
        lat.append((time.perf_counter_ns() - t0) / 1e3)
        states.append(pose["state"])
        msgs += [pose, odom]
    return np.array(states), np.array(lat), msgs
