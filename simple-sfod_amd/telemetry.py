"""What a multi-GPU bench line needs to explain itself (round-5 review item 7): per-GPU clock / power samples taken
beside the timed region (``rocm-smi`` in a host thread of rank 0: no GPU call in this process) and the algorithm /
protocol RCCL chose for the gradient exchange (parsed from the ``NCCL_DEBUG=INFO`` subset RCCL writes to
``NCCL_DEBUG_FILE``).  Standard library only; everything here degrades to ``None`` instead of failing a run.

The reference has no counterpart (d2's ``launch`` + DDP, ``source_free_adaptive_teacher.py:70-73``, log nothing about the
transport); this is measurement plumbing of ``bench.py``.
"""
import json
import os
import re
import subprocess
import threading
import time

ALGO = {0: "TREE", 1: "RING", 2: "COLLNET_DIRECT", 3: "COLLNET_CHAIN", 4: "NVLS", 5: "NVLS_TREE"}
PROTO = {0: "LL", 1: "LL128", 2: "SIMPLE"}


# ---- rocm-smi samples ---------------------------------------------------------------------------------------------------
def parse_smi(text):
    """one ``rocm-smi --showpower --showclocks --json`` document -> {card: (watts or None, sclk MHz or None)}"""
    try:
        doc = json.loads(text)
    except Exception:
        return {}
    out = {}
    for card, fields in doc.items():
        if not isinstance(fields, dict) or not card.startswith("card"):
            continue
        watts = next((float(v) for k, v in fields.items() if "ower" in k and re.match(r"^[0-9.]+$", str(v))), None)
        sclk = next((v for k, v in fields.items() if k.lower().startswith("sclk") and "speed" in k.lower()), None)
        m = re.search(r"([0-9]+)\s*Mhz", str(sclk), re.I) if sclk else None
        out[card] = (watts, int(m.group(1)) if m else None)
    return out


def under_profiler():
    """rocprofv3 preloads its tool library into the process (and, through the environment, into every child): a child that
    hops through ``#!/usr/bin/env`` then counts as a GPU-initialised process replacing itself, which this pool refuses.  No
    sampling beside a profiled run."""
    pre = os.environ.get("LD_PRELOAD", "") + os.environ.get("ROCP_TOOL_LIBRARIES", "")
    return "rocprof" in pre.lower() or "ROCPROFILER_LIBRARY_CTOR" in os.environ


def smi_command(args=("--showpower", "--showclocks", "--json")):
    """``rocm-smi`` is a Python script behind ``#!/usr/bin/env python3``: run it with THIS interpreter directly -- one exec
    in the child, no ``env`` hop in between"""
    import shutil
    import sys
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    real = os.path.realpath(exe)
    try:
        with open(real, "rb") as fh:
            head = fh.read(64)
    except OSError:
        return [exe] + list(args)
    if head.startswith(b"#!") and b"python" in head.split(b"\n")[0]:
        return [sys.executable, real] + list(args)
    return [exe] + list(args)


class SmiSampler:
    """Samples every GPU of the node every ``interval`` seconds from a daemon thread (a child ``rocm-smi`` per sample)."""

    def __init__(self, interval=0.4, cmd=None):
        self.interval, self.cmd = interval, list(cmd) if cmd is not None else smi_command()
        self.samples = []            # (perf_counter time, {card: (W, MHz)})
        self._stop = threading.Event()
        self._thread = None

    def _run(self):
        while not self._stop.is_set():
            t = time.perf_counter()
            try:
                r = subprocess.run(self.cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=5)
                s = parse_smi(r.stdout.decode(errors="replace"))
                if s:
                    self.samples.append((0.5 * (t + time.perf_counter()), s))
            except Exception:
                pass
            self._stop.wait(self.interval)

    def start(self):
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()
        return self

    def stop(self):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=6)

    def summary(self, t0, t1, cards=None):
        """per card over the samples taken in [t0, t1] (perf_counter times): mean / max power, mean / min sclk"""
        rows = [s for t, s in self.samples if t0 <= t <= t1]
        if not rows:
            return None
        out = {}
        for card in sorted({c for s in rows for c in s}, key=lambda c: int(re.sub(r"\D", "", c) or 0)):
            if cards is not None and card not in cards:
                continue
            w = [s[card][0] for s in rows if card in s and s[card][0] is not None]
            f = [s[card][1] for s in rows if card in s and s[card][1] is not None]
            out[card] = {"samples": len(w), "power_W_mean": round(sum(w) / len(w), 1) if w else None,
                         "power_W_max": round(max(w), 1) if w else None,
                         "sclk_MHz_mean": round(sum(f) / len(f)) if f else None, "sclk_MHz_min": min(f) if f else None}
        return out or None


# ---- what RCCL chose ----------------------------------------------------------------------------------------------------
def rccl_debug_setup(rank, directory="/tmp"):
    """Before ``init_process_group``: have RCCL write its INFO lines of the INIT / GRAPH / TUNING subsystems to a per-rank
    file.  -> the file this rank writes.

    What the environment already says is kept where it says MORE: a ``NCCL_DEBUG_FILE`` of the user's is used as it is (with
    their level), ``NCCL_DEBUG=INFO`` / ``TRACE`` keeps its own subsystem list.  A quieter level is raised -- the GPU boxes of
    this pool export ``NCCL_DEBUG=VERSION``, with which RCCL prints a five-line banner on STDOUT, where ``bench.py``'s one JSON
    line goes: the per-rank file takes the banner too."""
    if "NCCL_DEBUG_FILE" in os.environ:
        return os.environ["NCCL_DEBUG_FILE"]
    path = os.path.join(directory, f"sfod_rccl_{os.getpid()}_rank{rank}.log")
    os.environ["NCCL_DEBUG_FILE"] = path
    if os.environ.get("NCCL_DEBUG", "").upper() not in ("INFO", "TRACE"):
        os.environ["NCCL_DEBUG"] = "INFO"
        os.environ["NCCL_DEBUG_SUBSYS"] = "INIT,GRAPH,TUNING"
    return path


# the TUNING line of a collective: RCCL 2.26 (this image) prints NAMES and the channel range --
#   "AllReduce: 466747392 Bytes -> Algo RING proto SIMPLE channel{Lo..Hi}={0..63}"
# -- older builds printed the enum values and a predicted time ("... -> Algo 1 proto 2 time 2345.6"); both are read
_TUNING = re.compile(r"(\w+): (\d+) Bytes -> Algo (\w+) proto (\w+)(?: channel\{Lo\.\.Hi\}=\{(\d+)\.\.(\d+)\})?")


def parse_rccl_log(text):
    """-> {version, channels, transports: {kind: n}, collectives: [{coll, bytes, algo, proto, calls}]} from INFO lines"""
    out = {"version": None, "channels": None, "transports": {}, "collectives": []}
    seen = {}
    for line in text.splitlines():
        m = _TUNING.search(line)
        if m:
            algo = ALGO.get(int(m.group(3)), m.group(3)) if m.group(3).isdigit() else m.group(3).upper()
            proto = PROTO.get(int(m.group(4)), m.group(4)) if m.group(4).isdigit() else m.group(4).upper()
            nch = int(m.group(6)) - int(m.group(5)) + 1 if m.group(5) is not None else None
            key = (m.group(1), int(m.group(2)), algo, proto, nch)
            seen[key] = seen.get(key, 0) + 1
            continue
        m = re.search(r"(RCCL|NCCL) version\s*:?\s*([^\s:]\S*)", line)
        if m and out["version"] is None:
            out["version"] = f"{m.group(1)} {m.group(2)}"
        m = re.search(r"(\d+) coll channels", line)
        if m:
            out["channels"] = int(m.group(1))
        m = re.search(r" via ([A-Za-z0-9_/]+)", line)
        if m and "Channel" in line:
            out["transports"][m.group(1)] = out["transports"].get(m.group(1), 0) + 1
    for (coll, nbytes, algo, proto, nch), n in sorted(seen.items(), key=lambda kv: -kv[0][1]):
        row = {"coll": coll, "bytes": nbytes, "algo": algo, "proto": proto, "calls": n}
        if nch is not None:
            row["channels"] = nch
        out["collectives"].append(row)
    out["collectives"] = out["collectives"][:8]
    return out


def rccl_summary(path):
    if not path:
        return None
    try:
        with open(path, errors="replace") as fh:
            s = parse_rccl_log(fh.read())
    except OSError:
        return None
    if s["version"] is None and not s["collectives"] and not s["transports"]:
        return None
    return s
