"""The output side: every leg in full as its own JSON line, and LAST the compact line (< LINE_LIMIT bytes) the driver parses."""
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np

from .common import BENCH_PY, ROOT  # noqa: F401
from .common import _OUT, log


def start_deadline(rank):
    """A rank that dies or a transport that stalls must not keep the whole job (and whoever launched it) waiting for a
    chain of collective time-outs: after RR_BENCH_DEADLINE_S seconds (default 600, 0 = none) rank 0 prints what it has --
    the headline leg if that finished, flagged `deadline_exceeded` -- and every rank leaves."""
    import threading

    limit = float(os.environ.get("RR_BENCH_DEADLINE_S", "600"))
    if limit <= 0:
        return

    def fire():
        log(f"deadline of {limit:.0f} s exceeded -- leaving")
        rc = 3
        if rank == 0 and _OUT["partial"] is not None and not _OUT["emitted"]:
            line = dict(_OUT["partial"])
            line["deadline_exceeded"] = True
            emit(line)
            rc = 0
        os._exit(rc if rank == 0 else 0)

    t = threading.Timer(limit, fire)
    t.daemon = True
    t.start()


HEADLINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data")


LINE_LIMIT = 3500  # bytes: the driver keeps a short tail of stdout; round 4's 24.7 KB line fell off it (BENCH_r04.json parsed: null)


def _num(v, digits=6):
    """numbers of the compact line carry 6 significant digits; everything else passes through"""
    if isinstance(v, float) and math.isfinite(v):
        return float(f"{v:.{digits}g}")
    return v


def _short(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[: n - 3] + "..."


def _compact_roofline(r):
    if not isinstance(r, dict):
        return None
    out = {k: _num(r[k]) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "binding_frac", "traffic", "avg_kernel_ms",
                                   "algorithmic_bytes_per_launch") if k in r}
    if "kernel" in out:
        out["kernel"] = _short(str(out["kernel"]).split(" ")[0], 40)
    if r.get("traffic_source"):  # which PMC summary, and the hash of the library it was collected on (== the loaded one, or no traffic)
        out["traffic_source"] = _short(str(r["traffic_source"]), 110)
    return out


def _compact_cpu(c):
    if not isinstance(c, dict):
        return None
    out = {k: _num(c[k]) for k in ("value", "unit", "cores", "kind") if k in c}
    if "sample" in c:
        out["sample"] = _short(c["sample"], 330)
    host = c.get("host")
    if isinstance(host, dict):
        out["host"] = _short(f"{host.get('cpu_model', '?')}, {host.get('nproc', '?')} hw threads, {host.get('threads', '?')} used", 90)
    return out


def _leg_row(leg):
    """[ms_per_step, roofline fraction of the leg's dominant kernel (HBM), fraction of its binding resource]"""
    if not isinstance(leg, dict):
        return None
    if "error" in leg:
        return {"error": _short(leg["error"], 80)}
    r = leg.get("roofline") if isinstance(leg.get("roofline"), dict) else {}
    row = [_num(leg.get("ms_per_step"), 5), _num(r.get("frac"), 4), _num(r.get("binding_frac", r.get("frac")), 4)]
    if "median_ms" in leg:  # (legs timed step by step: the median beside the mean)
        row.append({"median_ms": _num(leg["median_ms"], 5), "max_ms": _num(leg.get("max_ms"), 4)})
    return row


def compact_line(out):
    """The line the driver parses: the contract's headline fields, `roofline`, `cpu_baseline`, and one short row per extra
    leg.  Every leg in full goes out as its own earlier JSON line and into bench_legs.json (emit)."""
    line = {k: _num(out[k], 9) for k in HEADLINE_KEYS if k in out}
    cfg = out.get("config")
    if isinstance(cfg, dict):
        line["config"] = {k: (_short(v, 200) if isinstance(v, str) else v) for k, v in cfg.items()}
    if "roofline" in out:
        line["roofline"] = _compact_roofline(out["roofline"])
    if "cpu_baseline" in out:
        line["cpu_baseline"] = _compact_cpu(out["cpu_baseline"])
    for k in ("device_warmup_steps", "ms_per_step_cold", "ms_per_step_cold_unwarmed", "deadline_exceeded", "error", "ranks_seen", "shared_device", "transport_fallback"):
        if k in out:
            line[k] = _short(out[k], 320) if isinstance(out[k], str) else _num(out[k], 5)
    legs = {}
    for name, leg in out.items():
        if not isinstance(leg, dict) or name in ("config", "roofline", "cpu_baseline", "kernel_ms_avg", "index_parity"):
            continue
        if "ms_per_step" in leg or "error" in leg:
            legs[name] = _leg_row(leg)
        else:  # a group of legs (sharded_world1: {p2p, rccl}; small_n: {rows})
            for sub, v in leg.items():
                if isinstance(v, dict) and ("ms_per_step" in v or "error" in v):
                    legs[f"{name}.{sub}"] = _leg_row(v)
    if legs:
        line["legs"] = legs
        line["legs_columns"] = ["ms_per_step", "hbm_frac", "binding_frac"]
    if isinstance(out.get("sharded"), dict):
        line["sharded"] = {k: out["sharded"].get(k) for k in ("transport", "p2p_timed_out", "ranks_seen") if k in out["sharded"]}
    for k in ("strong_scaling_ceiling", "weak_scaling_ceiling"):
        if k in out:
            line[k] = out[k]
    # resample indices against the LITERAL float walk on identical weights and draws: [differing slots, slots] per scheme (against
    # the integer CDF of the D-spec they are identical at every size: tests/)
    ip = {}
    for src in (out, out.get("mcl_multinomial")):
        d = src.get("index_parity") if isinstance(src, dict) else None
        if isinstance(d, dict) and "differing_slots_vs_literal_float_walk" in d:
            ip[str(d.get("resample", "?"))] = [d["differing_slots_vs_literal_float_walk"], d.get("slots")]
    if ip:
        line["index_parity"] = ip
    if legs:
        line["full"] = "bench_legs.json; every leg also as its own JSON line above this one"
    data = json.dumps(line)
    while len(data) > LINE_LIMIT:  # never again a line the driver cannot read: shed the optional parts, longest first
        for k in ("legs", "cpu_baseline.sample", "config.workload", "roofline"):
            if "." in k:
                a, b = k.split(".")
                if isinstance(line.get(a), dict) and isinstance(line[a].get(b), str) and len(line[a][b]) > 60:
                    line[a][b] = _short(line[a][b], 60)
                    break
            elif k in line and k == "legs":
                line.pop("legs")
                line.pop("legs_columns", None)
                break
        else:
            line = {k: line[k] for k in HEADLINE_KEYS if k in line}
            data = json.dumps(line)
            break
        data = json.dumps(line)
    return data


def emit(out):
    """Last on stdout: ONE compact JSON line (compact_line, < LINE_LIMIT bytes) with the contract's fields.  Before it, every
    extra leg in full as its own JSON line ({"leg": name, ...}), and the whole record in bench_legs.json.  Native libraries
    (RCCL's version banner) write through C stdio, so drain that buffer first."""
    import ctypes

    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    _OUT["emitted"] = True
    chunks = []
    head = {k: v for k, v in out.items() if not (isinstance(v, dict) and k not in ("config", "roofline", "cpu_baseline", "kernel_ms_avg",
                                                                                 "index_parity", "plain_async_step",
                                                                                 "synchronous_try_step"))}
    if len(out) > len(head):
        chunks.append(json.dumps({"leg": "headline", **head}))
        for k, v in out.items():
            if k not in head:
                chunks.append(json.dumps({"leg": k, **v}))
    try:
        with open(os.environ.get("RR_BENCH_LEGS_FILE", os.path.join(ROOT, "bench_legs.json")), "w") as f:
            json.dump(out, f, indent=1)
    except OSError:
        pass
    chunks.append(compact_line(out))
    data = ("\n".join(chunks) + "\n").encode()
    if _OUT["fd"] is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_OUT["fd"], data)
