"""nodes/pf_localizer_node without a GPU: it builds, its helpers pass the reference node's own unit tests
(ros2_nodes/ekf_localizer_node/src/main.rs:303-384, run by --self-test), and it fails the way the reference's node does when the
localizer cannot be made: a warning per message, no output, no crash (main.rs:225-231)."""
import os
import subprocess

import pytest

from tests import node_driver as D


def test_node_binary_is_built_and_passes_its_self_test():
    assert os.path.exists(D.NODE), "run `make -C nodes/pf_localizer_node` (or __graft_entry__.build())"
    r = subprocess.run([D.NODE, "--self-test"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert "self-test ok" in r.stdout


def test_node_without_a_device_logs_and_carries_on():
    from rust_robotics_amd import _ffi

    if _ffi.lib().rr_device_count() > 0:
        pytest.skip("a GPU is present: covered by tests/test_gpu_node.py")
    node = D.NodeProcess(env={"PF_LOG_INTERVAL_S": "0"})
    node.send(D.odom_line("/odom", 0, (5.0, 5.0, 0.0), 0.0, 0.0), D.odom_line("/odom", 1, (5.0, 5.0, 0.0), 1.0, 0.0))
    rc, err = node.close()
    assert rc == 0
    assert err.count("failed to initialize PF state") == 2 and "pf localizer started" in err


def test_invalid_configuration_is_fatal_at_start():
    """main.rs:178-180: an invalid initial configuration ends main() with an error."""
    r = subprocess.run([D.NODE], env=dict(os.environ, PF_PARTICLES="0"), stdin=subprocess.DEVNULL, capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "at least one particle" in r.stderr
