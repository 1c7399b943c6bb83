"""bench.py's self-explanation plumbing (simple-sfod_amd/telemetry.py): rocm-smi sample parsing / windowing and the
summary of RCCL's INFO lines -- the parts that cannot be exercised on a one-GPU box."""
import importlib
import time

tel = importlib.import_module("simple-sfod_amd.telemetry")

SMI = ('{"card0": {"sclk clock speed:": "(1928Mhz)", "sclk clock level:": "S", "mclk clock speed:": "(2000Mhz)", '
       '"Current Socket Graphics Package Power (W)": "1288.0"}, '
       '"card1": {"sclk clock speed:": "(2100Mhz)", "Current Socket Graphics Package Power (W)": "900.5"}, "system": {"x": 1}}')

RCCL = """
host:123:456 [0] NCCL INFO RCCL version 2.22.3+hip6.4 HEAD:abcdef
host:123:456 [0] NCCL INFO Channel 00/0 : 0[0] -> 1[1] via P2P/IPC
host:123:456 [0] NCCL INFO Channel 01/0 : 0[0] -> 1[1] via P2P/IPC
host:123:456 [0] NCCL INFO Channel 00/0 : 7[7] -> 0[0] via P2P/direct pointer
host:123:456 [0] NCCL INFO 32 coll channels, 32 collnet channels, 0 nvls channels, 32 p2p channels, 4 p2p channels per peer
host:123:456 [0] NCCL INFO AllReduce: 469762048 Bytes -> Algo 1 proto 2 time 2345.6
host:123:456 [0] NCCL INFO AllReduce: 469762048 Bytes -> Algo 1 proto 2 time 2345.6
host:123:456 [0] NCCL INFO AllReduce: 291504128 Bytes -> Algo 1 proto 2 time 1500.0
host:123:456 [0] NCCL INFO AllReduce: 4096 Bytes -> Algo 0 proto 0 time 12.0
host:123:456 [0] NCCL INFO Broadcast: 36 Bytes -> Algo 1 proto 0 time 9.0
"""


def test_smi_document_and_window():
    s = tel.parse_smi(SMI)
    assert s == {"card0": (1288.0, 1928), "card1": (900.5, 2100)}
    assert tel.parse_smi("not json") == {}
    sm = tel.SmiSampler()
    now = time.perf_counter()
    sm.samples = [(now - 10, {"card0": (100.0, 500)}), (now, s), (now + 1, {"card0": (1300.0, 1900), "card1": (910.5, 2000)})]
    out = sm.summary(now - 1, now + 2)
    assert out["card0"] == {"samples": 2, "power_W_mean": 1294.0, "power_W_max": 1300.0, "sclk_MHz_mean": 1914, "sclk_MHz_min": 1900}
    assert out["card1"]["sclk_MHz_min"] == 2000 and out["card1"]["power_W_mean"] == 905.5
    assert sm.summary(now + 5, now + 6) is None


def test_sampler_survives_a_missing_tool():
    sm = tel.SmiSampler(interval=0.05, cmd=("/nonexistent/rocm-smi",)).start()
    time.sleep(0.2)
    sm.stop()
    assert sm.samples == [] and sm.summary(0, 1e18) is None


def test_rccl_info_lines():
    s = tel.parse_rccl_log(RCCL)
    assert s["version"] == "RCCL 2.22.3+hip6.4" and s["channels"] == 32
    assert s["transports"] == {"P2P/IPC": 2, "P2P/direct": 1}
    big = s["collectives"][0]
    assert big == {"coll": "AllReduce", "bytes": 469762048, "algo": "RING", "proto": "SIMPLE", "calls": 2}
    assert {"coll": "AllReduce", "bytes": 4096, "algo": "TREE", "proto": "LL", "calls": 1} in s["collectives"]
    assert tel.rccl_summary(None) is None and tel.rccl_summary("/nonexistent/file.log") is None


def test_rccl_2_26_formats_of_this_image():
    """What the RCCL inside this image's torch really writes: ``tests/golden/rccl_2_26_one_rank_info.log`` is the INFO log of a
    one-rank communicator on an MI355X box (tools/experiments/probe_rccl_one_rank.py; repeated per-channel lines thinned), the
    multi-rank lines below are its format strings (``strings librccl.so``) filled in: names instead of enum values in the TUNING
    line, a channel range instead of a predicted time, ``version :`` with a colon."""
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    s = tel.rccl_summary(os.path.join(here, "golden", "rccl_2_26_one_rank_info.log"))
    assert s["version"] == "RCCL 2.26.6-HEAD:64f48b6" and s["channels"] == 128 and s["collectives"] == [] and s["transports"] == {}
    multi = """
runc:376:446 [0] NCCL INFO RCCL version : 2.26.6-HEAD:64f48b6
runc:376:446 [0] NCCL INFO Channel 00/0 : 0[75000] -> 1[85000] via P2P/IPC comm 0x55 nRanks 08
runc:376:446 [0] NCCL INFO Channel 01/0 : 0[75000] -> 1[85000] via P2P/IPC comm 0x55 nRanks 08
runc:376:446 [0] NCCL INFO Channel 00/0 : 7[f5000] -> 0[75000] via P2P/direct pointer comm 0x55 nRanks 08
runc:376:446 [0] NCCL INFO 64 coll channels, 64 collnet channels, 0 nvls channels, 64 p2p channels, 8 p2p channels per peer
runc:376:446 [0] NCCL INFO AllReduce: 466747392 Bytes -> Algo RING proto SIMPLE channel{Lo..Hi}={0..63}
runc:376:446 [0] NCCL INFO AllReduce: 466747392 Bytes -> Algo RING proto SIMPLE channel{Lo..Hi}={0..63}
runc:376:446 [0] NCCL INFO AllReduce: 4280320 Bytes -> Algo TREE proto LL128 channel{Lo..Hi}={0..15}
runc:376:446 [0] NCCL INFO Broadcast: 36 Bytes -> Algo RING proto LL channel{Lo..Hi}={0..0}
"""
    s = tel.parse_rccl_log(multi)
    assert s["version"] == "RCCL 2.26.6-HEAD:64f48b6" and s["channels"] == 64 and s["transports"] == {"P2P/IPC": 2, "P2P/direct": 1}
    assert s["collectives"][0] == {"coll": "AllReduce", "bytes": 466747392, "algo": "RING", "proto": "SIMPLE", "calls": 2, "channels": 64}
    assert {"coll": "AllReduce", "bytes": 4280320, "algo": "TREE", "proto": "LL128", "calls": 1, "channels": 16} in s["collectives"]
    assert {"coll": "Broadcast", "bytes": 36, "algo": "RING", "proto": "LL", "calls": 1, "channels": 1} in s["collectives"]


def test_debug_setup_respects_the_users_settings(monkeypatch, tmp_path):
    for k in ("NCCL_DEBUG", "NCCL_DEBUG_FILE", "NCCL_DEBUG_SUBSYS"):
        monkeypatch.delenv(k, raising=False)
    p = tel.rccl_debug_setup(3, directory=str(tmp_path))
    import os
    assert p.endswith("rank3.log") and os.environ["NCCL_DEBUG"] == "INFO" and os.environ["NCCL_DEBUG_FILE"] == p
    monkeypatch.setenv("NCCL_DEBUG", "WARN")
    monkeypatch.setenv("NCCL_DEBUG_FILE", "/elsewhere.log")
    assert tel.rccl_debug_setup(0) == "/elsewhere.log" and os.environ["NCCL_DEBUG"] == "WARN"
    # the GPU boxes of this pool export NCCL_DEBUG=VERSION (RCCL then prints a banner on stdout, beside bench.py's JSON line):
    # a quieter level without a file is raised to INFO and redirected; INFO with the user's own subsystem list is kept
    monkeypatch.delenv("NCCL_DEBUG_FILE")
    monkeypatch.setenv("NCCL_DEBUG", "VERSION")
    p = tel.rccl_debug_setup(1, directory=str(tmp_path))
    assert os.environ["NCCL_DEBUG"] == "INFO" and os.environ["NCCL_DEBUG_FILE"] == p and "TUNING" in os.environ["NCCL_DEBUG_SUBSYS"]
    monkeypatch.delenv("NCCL_DEBUG_FILE")
    monkeypatch.setenv("NCCL_DEBUG", "INFO")
    monkeypatch.setenv("NCCL_DEBUG_SUBSYS", "COLL")
    p = tel.rccl_debug_setup(2, directory=str(tmp_path))
    assert os.environ["NCCL_DEBUG_SUBSYS"] == "COLL" and os.environ["NCCL_DEBUG_FILE"] == p
    for k in ("NCCL_DEBUG", "NCCL_DEBUG_FILE", "NCCL_DEBUG_SUBSYS"):
        monkeypatch.delenv(k, raising=False)
