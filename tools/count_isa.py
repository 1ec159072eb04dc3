#!/usr/bin/env python3
"""Reads the instruction budget of the MCL propagate + weight kernel off its gfx950 ISA and writes
rust_robotics_amd/csrc/INSTRUCTION_BUDGET.json (bench.py's `roofline.fp64_valu` uses it).

    python tools/count_isa.py                      # print the counts
    python tools/count_isa.py --write [--sq-insts-valu-per-wave 1738]

per_pair      VALU instructions of one (particle, landmark) pair: the pair loop is the innermost loop of
              k_step_lazy<kernarg obs, kSrcMarkers, RR_LIK_FUSED> that holds v_rsq_f64 (the square-root core); one
              iteration handles `unroll` observations for the thread's kResolveRows rows.  Counted from the compiler's
              own output (hipcc -S with the flags of csrc/Makefile), nothing assumed.
per_particle  everything else, per particle.  A static count cannot give this (both branches of every `if` are in the
              text), so it comes from the hardware counter: SQ_INSTS_VALU per wave (rocprofv3 --pmc, a wave carries
              kResolveRows x 64 / 64 = 2 particle rows) / 2 - L * per_pair at the profiled L = 32.  Pass the counter with
              --sq-insts-valu-per-wave (profiles/r*_mcl_pmc_sq_summary.csv); without it the JSON keeps its previous value.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rust_robotics_amd", "csrc")
OUT = os.path.join(CSRC, "INSTRUCTION_BUDGET.json")
KERNEL = "k_step_lazyILb1ELi0ELi0ELb0ELb0EE"  # <OBS_KERNARG = true, SRC = kSrcMarkers, LIK = RR_LIK_FUSED, PACKED = false, EST = false>
KERNEL_EST = "k_step_lazyILb1ELi0ELi0ELb0ELb1EE"  # ... EST = true: the build that also adds up the deferred in-step estimate (--est)
ROWS = 2


def makefile_flags():
    text = open(os.path.join(CSRC, "Makefile")).read()
    m = re.search(r"^FLAGS\s*\?=\s*(.*)$", text, flags=re.M)
    flags = m.group(1).replace("$(ARCH)", "gfx950").split()
    return [f for f in flags if f not in ("-fPIC",)]


def isa():
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "pf.s")
        cmd = ["/opt/rocm/bin/hipcc"] + makefile_flags() + ["-S", "--cuda-device-only", "-o", out, "pf_engine.hip"]
        r = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
        if r.returncode != 0:
            sys.exit(r.stderr[-2000:])
        return open(out).read()


def kernel_body(text, kernel=None):
    global KERNEL
    if kernel:
        KERNEL = kernel
    start = None
    lines = text.splitlines()
    for i, ln in enumerate(lines):
        if re.match(r"^_ZN\S*" + KERNEL + r"\S*:\s*(;.*)?$", ln):
            start = i
            break
    if start is None:
        sys.exit(f"kernel {KERNEL} not found in the ISA")
    body = []
    for ln in lines[start + 1:]:
        body.append(ln)
        if "s_endpgm" in ln:
            break
    return body


def loops(body):
    """(first line, last line) of every backward branch target span"""
    labels = {m.group(1): i for i, ln in enumerate(body) for m in [re.match(r"^(\.LBB\S+):", ln)] if m}
    spans = []
    for i, ln in enumerate(body):
        m = re.match(r"^\s+s_cbranch_\w+\s+(\.LBB\S+)", ln) or re.match(r"^\s+s_branch\s+(\.LBB\S+)", ln)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            spans.append((labels[m.group(1)], i))
    return spans


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true")
    ap.add_argument("--sq-insts-valu-per-wave", type=float, default=None)
    ap.add_argument("--landmarks", type=int, default=32)
    ap.add_argument("--est", action="store_true", help="count the EST build (the headline step since round 5: the deferred in-step estimate); "
                                                       "--write then fills the JSON's \"est\" entry")
    args = ap.parse_args()
    body = kernel_body(isa(), KERNEL_EST if args.est else None)
    cand = [(a, b) for a, b in loops(body) if any("v_rsq_f64" in ln for ln in body[a:b + 1])]
    if not cand:
        sys.exit("no loop with v_rsq_f64 in the kernel")
    # the main pair loop is unrolled (4 observations x 2 rows = 8 square roots per iteration); the remainder loop and the
    # exact-form fallback (an overflowing squared range) hold one or two: take the loop with the most, the shortest among equals
    def n_rsq_of(ab):
        return sum(1 for ln in body[ab[0]:ab[1] + 1] if "v_rsq_f64" in ln)

    best = max(n_rsq_of(ab) for ab in cand)
    a, b = min((ab for ab in cand if n_rsq_of(ab) == best), key=lambda ab: ab[1] - ab[0])
    ops = {}
    for ln in body[a:b + 1]:
        m = re.match(r"^\s+(v_\w+)", ln)
        if m:
            ops[m.group(1)] = ops.get(m.group(1), 0) + 1
    n_valu = sum(ops.values())
    n_rsq = ops.get("v_rsq_f64_e32", 0) + ops.get("v_rsq_f64_e64", 0) + ops.get("v_rsq_f64", 0)
    pairs = n_rsq  # one square root per pair
    per_pair = n_valu / pairs
    total_valu = sum(1 for ln in body if re.match(r"^\s+v_\w+", ln))
    print(f"kernel {KERNEL}: {total_valu} VALU instructions in the text; pair loop lines {a}..{b}: {n_valu} VALU instructions for {pairs} pairs "
          f"({pairs // ROWS} observations x {ROWS} rows) -> per_pair = {per_pair:.4f}")
    print("  " + ", ".join(f"{k} {v}" for k, v in sorted(ops.items(), key=lambda kv: -kv[1])))
    old = json.load(open(OUT)) if os.path.exists(OUT) else {}
    old_entry = old.get("est", {}) if args.est else old
    per_particle = old_entry.get("per_particle")
    src_pp = old_entry.get("per_particle_source", "previous value")
    if args.sq_insts_valu_per_wave is not None:
        per_particle = args.sq_insts_valu_per_wave / ROWS - args.landmarks * per_pair
        src_pp = f"SQ_INSTS_VALU per wave {args.sq_insts_valu_per_wave:g} (rocprofv3 --pmc) / {ROWS} rows - {args.landmarks} x per_pair"
        print(f"per_particle = {per_particle:.1f}  ({src_pp})")
    if args.write and args.est:
        old["est"] = {"kernel": "k_step_lazy<true, kSrcMarkers, RR_LIK_FUSED, false, EST> (the same + the deferred in-step estimate of the step before)",
                      "per_pair": round(per_pair, 4), "per_particle": per_particle,
                      "source": f"tools/count_isa.py --est: {n_valu} VALU instructions for {pairs} pairs in the pair loop", "per_particle_source": src_pp}
        json.dump(old, open(OUT, "w"), indent=2)
        print("wrote", OUT, "(est)")
    elif args.write:
        d = {"kernel": "k_step_lazy<true, kSrcMarkers, RR_LIK_FUSED> (MCL propagate + weight, 10^6 particles x 32 landmarks)",
             "per_pair": round(per_pair, 4), "per_particle": per_particle,
             "source": f"tools/count_isa.py: per_pair = VALU instructions of the pair loop in the gfx950 ISA ({n_valu} for {pairs} pairs: "
                       + ", ".join(f"{v} {k}" for k, v in sorted(ops.items(), key=lambda kv: -kv[1])) + ")",
             "per_particle_source": src_pp}
        if "measured_issue_rate" in old:
            d["measured_issue_rate"] = old["measured_issue_rate"]
        if "est" in old:
            d["est"] = old["est"]
        json.dump(d, open(OUT, "w"), indent=2)
        print("wrote", OUT)


if __name__ == "__main__":
    main()
