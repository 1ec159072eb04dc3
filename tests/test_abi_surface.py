"""CPU-only checks of the drop-in boundary: the shared library loads without a GPU, exports
every function include/*.h declares, fails loudly (no CPU fallback) when asked to compute, and
its host-side validation carries the reference's error messages."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADERS = ["rr_pf.h", "rr_fastslam1.h", "rr_fastslam2.h"]


def declared_functions(header):
    path = os.path.join(ROOT, "include", header)
    if not os.path.exists(path):
        return []
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rr_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def ffi():
    from rust_robotics_amd import _ffi

    return _ffi


def test_library_loads_and_exports_every_declared_symbol(ffi):
    L = ffi.lib()
    names = [n for h in HEADERS for n in declared_functions(h)]
    assert len(names) > 40
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, f"declared in include/ but not exported: {missing}"
    assert L.rr_version().decode().startswith("rust_robotics_amd")


def test_config_defaults_and_validation_messages(ffi):
    L = ffi.lib()
    cfg = ffi.PfConfig()
    L.rr_pf_config_default(C.byref(cfg))
    # particle_filter.rs:67-78
    assert (cfg.n_particles, cfg.resample_threshold, cfg.range_noise, cfg.velocity_noise, cfg.dt) == (100, 0.5, 0.2, 2.0, 0.1)
    assert abs(cfg.yaw_rate_noise - 0.6981317007977318) < 1e-15
    assert L.rr_pf_config_validate(C.byref(cfg)) == ffi.RR_OK
    cases = [
        ("n_particles", 0, "particle filter requires at least one particle"),
        ("resample_threshold", 1.5, "particle filter resample_threshold must be within [0.0, 1.0]"),
        ("resample_threshold", float("nan"), "particle filter resample_threshold must be within [0.0, 1.0]"),
        ("range_noise", 0.0, "particle filter range_noise must be positive and finite"),
        ("velocity_noise", -1.0, "particle filter velocity_noise must be non-negative and finite"),
        ("yaw_rate_noise", float("inf"), "particle filter yaw_rate_noise must be non-negative and finite"),
        ("dt", 0.0, "particle filter dt must be positive and finite"),
    ]
    for field, value, msg in cases:  # particle_filter.rs:81-117
        c = ffi.PfConfig()
        L.rr_pf_config_default(C.byref(c))
        setattr(c, field, value)
        assert L.rr_pf_config_validate(C.byref(c)) == ffi.RR_INVALID_PARAMETER
        assert ffi.last_error() == msg


def test_no_cpu_fallback(ffi):
    import rust_robotics_amd.localization as loc

    if ffi.lib().rr_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(loc.RoboticsError) as ei:
        loc.ParticleFilterLocalizer.with_defaults()
    assert "no CPU fallback" in str(ei.value)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "rust_robotics_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "ref_literal" not in text and "det_spec" not in text.replace("det_spec.c)", ""), f


def test_library_reports_the_hash_of_its_sources(ffi):
    """rr_version carries the first 16 hex digits of the SHA-256 over the sources the library was built from (csrc/Makefile: cat SRCS
    HDRS | sha256sum) -- what ties a PMC summary under profiles/ to a build (benchlib.common.library_sha16 / measured_traffic): two
    builds of the same sources are different bytes, the same kernels.  A library older than its sources fails here: run make."""
    import hashlib
    import re

    from benchlib.common import library_sha16

    csrc = os.path.join(ROOT, "rust_robotics_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    files = re.search(r"^SRCS\s*:=\s*(.*)$", mk, flags=re.M).group(1).split() + re.search(r"^HDRS\s*:=\s*(.*)$", mk, flags=re.M).group(1).split()
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(csrc, f), "rb").read())
    version = ffi.lib().rr_version().decode()
    m = re.search(r"sources ([0-9a-f]{16})\)", version)
    assert m, version
    assert m.group(1) == h.hexdigest()[:16], "librust_robotics_amd.so is older than its sources: make -C rust_robotics_amd/csrc"
    assert library_sha16() == m.group(1)


def test_no_plain_hipmemset_in_the_engine():
    """hipMemset(device memory) returns before it has run (tools/ubench/memset_sync_probe.hip, profiles/r06o_*) and the engine's
    streams are non-blocking ones: a plain hipMemset is a race with the next kernel.  Fills go through rr::memset_on / hipMemsetAsync
    on the consumer's stream; the transport's set-up (p2p_core.hpp) may use hipMemset only with a hipDeviceSynchronize behind it."""
    import re

    csrc = os.path.join(ROOT, "rust_robotics_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".hip", ".hpp", ".inc", ".h")):
            continue
        text = open(os.path.join(csrc, f)).read()
        code = re.sub(r"//[^\n]*", "", text)
        hits = [m.start() for m in re.finditer(r"\bhipMemset\(", code)]
        if f in ("p2p_core.hpp", "selftest.hip"):  # every one of them is followed, in the same function, by a device-wide synchronisation
            for at in hits:
                end = code.find("\n  }\n", at)
                assert "hipDeviceSynchronize()" in code[at:end if end > 0 else len(code)], (f, code[at:at + 80])
        else:
            assert not hits, (f, [code[at:at + 80] for at in hits])
        # ... and device memory comes from rr::dev_malloc (rr_common.hpp), which is what RR_DEBUG_POISON_ALLOC poisons: an allocation
        # made with a bare hipMalloc would escape tests/test_gpu_poison.py
        bare = [m.start() for m in re.finditer(r"(?<![\w:])hipMalloc\(", code)]
        assert not bare or f == "rr_common.hpp", (f, [code[at:at + 80] for at in bare])


def test_mirror_exposes_reference_names():
    import rust_robotics_amd.localization as loc

    for name in ("try_new", "with_defaults", "try_with_initial_state", "with_initial_state_2d", "try_set_landmarks",
                 "set_landmarks_from_obstacles", "set_range_noise", "get_landmarks", "get_particles",
                 "try_predict_with_control", "try_update_with_observations", "resample", "estimate", "state_2d",
                 "calc_covariance", "try_predict_input", "try_step_state", "try_step", "step", "predict", "update",
                 "get_state", "get_covariance"):  # particle_filter.rs:130-573
        assert hasattr(loc.ParticleFilterLocalizer, name), name
    for name in ("try_new", "try_with_initial_state", "try_predict_with_control", "try_update_with_observations",
                 "try_step", "estimate", "state_2d", "particle_count"):  # monte_carlo_localization.rs:144-320
        assert hasattr(loc.MonteCarloLocalizer, name), name


def test_cpp_wrapper_compiles(tmp_path):
    """include/rust_robotics.hpp (header-only C++ mirror of the reference's struct surface) must
    parse against the C ABI headers and link against the library's exported symbols"""
    import shutil
    import subprocess

    if not shutil.which("g++"):
        pytest.skip("no g++")
    src = tmp_path / "t.cpp"
    src.write_text('#include "rust_robotics.hpp"\n'
                   "int main() {\n"
                   "  rr::MonteCarloLocalizationConfig c; c.min_particles = 10; c.max_particles = 20;\n"
                   "  try { rr::MonteCarloLocalizer m(c); rr::ParticleFilterLocalizer p; (void)m.particle_count(); }\n"
                   "  catch (const std::exception&) { return 0; }\n"  # no GPU here: creation fails loudly
                   "  return 0;\n}\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(root, "include"), str(src)], capture_output=True,
                       text=True)
    assert r.returncode == 0, r.stderr[-2000:]


def test_rust_sys_bindings_are_current():
    """bindings/rust/rust_robotics_amd-sys/src/lib.rs is generated from include/*.h: it must be up to
    date and declare every entry point the headers declare (the crate itself cannot be compiled here)"""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_rust_sys.py"), "--check"])
    assert r.returncode == 0, "run `python tools/gen_rust_sys.py`"
    text = open(os.path.join(root, "bindings", "rust", "rust_robotics_amd-sys", "src", "lib.rs")).read()
    for sym in (n for h in HEADERS for n in declared_functions(h)):
        assert f"pub fn {sym}(" in text, sym


def test_rust_sys_struct_layouts_match_the_c_headers(tmp_path):
    """The generated `#[repr(C)]` structs cannot be compiled here (no rustc): instead the generator computes, from the RUST field
    types alone, the size / alignment / offset of every field as repr(C) lays it out, writes them as `const` assertions into
    the -sys crate and as _Static_asserts into tests/c/abi_layout.c -- and this makes the C compiler confirm every one of them
    against include/*.h."""
    import shutil
    import subprocess

    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "tests", "c", "abi_layout.c")
    text = open(src).read()
    assert text.count("_Static_assert") >= 60
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(root, "include"), "-c", src, "-o", str(tmp_path / "abi_layout.o")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lib = open(os.path.join(root, "bindings", "rust", "rust_robotics_amd-sys", "src", "lib.rs")).read()
    assert lib.count("::core::mem::offset_of!(") == text.count("offsetof(")
    # a deliberately wrong number must fail: the assertions are live
    bad = tmp_path / "bad.c"
    bad.write_text(text.replace("sizeof(rr_pf_config) == 48", "sizeof(rr_pf_config) == 56"))
    r = subprocess.run(["gcc", "-std=c11", "-I", os.path.join(root, "include"), "-c", str(bad), "-o", str(tmp_path / "bad.o")], capture_output=True, text=True)
    assert r.returncode != 0


def test_rust_examples_call_only_what_the_safe_layer_declares():
    """bindings/rust/rust_robotics_amd/examples/*.rs cannot be compiled here (no rustc): at least every method they call on a
    localizer / engine and every item they import from the crate must be declared `pub` in the safe layer, every sys:: symbol the
    safe layer itself calls must be in the generated -sys crate, and braces / parentheses must balance."""
    import glob
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    crate = os.path.join(root, "bindings", "rust", "rust_robotics_amd")
    lib = open(os.path.join(crate, "src", "lib.rs")).read()
    sys_lib = open(os.path.join(root, "bindings", "rust", "rust_robotics_amd-sys", "src", "lib.rs")).read()
    declared = set(re.findall(r"pub (?:fn|struct|type|mod|enum|trait) (\w+)", lib))
    for name in set(re.findall(r"sys::(rr_\w+)", lib)):
        assert re.search(r"\b%s\b" % name, sys_lib), f"the safe layer calls sys::{name}, which the -sys crate does not declare"
    examples = sorted(glob.glob(os.path.join(crate, "examples", "*.rs")))
    assert len(examples) >= 3
    for path in examples:
        text = open(path).read()
        code = "\n".join(l for l in text.splitlines() if not l.lstrip().startswith("//"))
        code = re.sub(r'"(?:[^"\\]|\\.)*"', '""', code)  # string literals out of the way
        code = re.sub(r"//.*", "", code)
        for o, c in ("{}", "()", "[]"):
            assert code.count(o) == code.count(c), f"{os.path.basename(path)}: unbalanced {o}{c}"
        for imp in re.findall(r"use rust_robotics_amd::(?:fastslam::)?\{([^}]*)\}", code):
            for item in (x.strip() for x in imp.split(",")):
                assert item in declared, f"{os.path.basename(path)} imports {item}, which the safe layer does not declare"
        for var in ("pf", "mcl", "slam", "engine", "generic"):
            for meth in set(re.findall(r"\b%s\.(\w+)\(" % var, code)):
                assert meth in declared, f"{os.path.basename(path)} calls {var}.{meth}(), which the safe layer does not declare"
        for ty, fn in re.findall(r"\b(ParticleFilterLocalizer|MonteCarloLocalizer|FastSlam1|Engine)::(\w+)\(", code):
            assert fn in declared, f"{os.path.basename(path)} calls {ty}::{fn}(), which the safe layer does not declare"



def test_instruction_budget_is_reproducible_from_the_isa():
    """rust_robotics_amd/csrc/INSTRUCTION_BUDGET.json (bench.py's roofline.fp64_valu) cites tools/count_isa.py: the tool must
    exist and read the same per-pair count off the compiler's ISA for the kernel as it is today (hipcc cross-compiles here)."""
    import json
    import shutil
    import subprocess
    import sys

    if not os.path.exists("/opt/rocm/bin/hipcc") and not shutil.which("hipcc"):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "count_isa.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = float(r.stdout.split("per_pair =")[1].split()[0])
    want = json.load(open(os.path.join(root, "rust_robotics_amd", "csrc", "INSTRUCTION_BUDGET.json")))
    assert abs(got - want["per_pair"]) < 1e-3, (got, want["per_pair"], "run `python tools/count_isa.py --write`")
    assert "tools/count_isa.py" in want["source"]


def test_every_part_of_the_translation_units_is_a_build_dependency():
    """pf_engine.hip / fs1_engine.hip #include their kernels as .inc parts: each part, and every shared header, must be listed in
    the Makefile's HDRS (an edit that does not rebuild the library would ship a stale .so to the GPU box), and must be included."""
    import glob
    import re

    csrc = os.path.join(ROOT, "rust_robotics_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    hdrs = set(re.search(r"^HDRS\s*:=\s*(.*)$", mk, flags=re.M).group(1).split())
    parts = {os.path.basename(f) for f in glob.glob(os.path.join(csrc, "*.inc")) + glob.glob(os.path.join(csrc, "*.hpp"))}
    assert parts <= hdrs, f"not a build dependency: {sorted(parts - hdrs)}"
    sources = "".join(open(os.path.join(csrc, f)).read() for f in ("pf_engine.hip", "fs1_engine.hip"))
    for p in parts:
        assert f'#include "{p}"' in sources or p.endswith(".hpp"), f"{p} is not included by any translation unit"
