"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/omx.h declares;
entry points fail loudly (OMX_ERR_NO_DEVICE) instead of falling back to a CPU path; stream sharding +
stats gather over gloo with world_size 2."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "omx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(omx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(omx):
    syms = declared_symbols()
    assert len(syms) > 60
    missing = [s for s in syms if not hasattr(omx.lib, s)]
    assert not missing, f"declared in include/omx.h but not exported: {missing}"


def test_oracle_mirrors_the_single_stream_abi(oracle):
    for s in declared_symbols():
        if "_bank_" in s or "_capture_group_" in s or "_debug_" in s or s in ("omx_last_error", "omx_device_available", "omx_device_count", "omx_set_device"):  # (many-stream forms, device selection)
            continue
        assert hasattr(oracle.lib, "omxo_" + s[4:]), s


def test_no_cpu_fallback_without_a_device(omx):
    import openmeters_amd
    from openmeters_amd import capi
    if openmeters_amd.device_available():
        pytest.skip("a GPU is visible: the no-device contract is checked on CPU-only hosts")
    for family, cfg in (("spectrogram", capi.SpectrogramConfig().to_c()), ("spectrum", capi.SpectrumConfig().to_c()),
                        ("loudness", capi.LoudnessConfig().to_c()), ("stereometer", capi.StereometerConfig().to_c()),
                        ("oscilloscope", capi.OscilloscopeConfig().to_c())):
        h = C.c_void_p()
        rc = omx.fn(f"{family}_create", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)])(C.byref(cfg), C.byref(h))
        assert rc == capi.ERR_NO_DEVICE and not h.value
    with pytest.raises(capi.OmxError) as e:
        capi.SpectrogramProcessor(omx, capi.SpectrogramConfig())
    assert e.value.status == capi.ERR_NO_DEVICE and "no CPU fallback" in str(e.value)
    # the many-stream banks and the capture group (VisualManager fan-out) refuse just as loudly
    from openmeters_amd import banks, pipeline
    for make in (lambda: banks.LoudnessBank(omx, capi.LoudnessConfig(), 4, 2), lambda: banks.StereometerBank(omx, capi.StereometerConfig(), 4),
                 lambda: banks.OscilloscopeBank(omx, capi.OscilloscopeConfig(), 4),
                 lambda: pipeline.CaptureGroup(omx, 4, spectrogram=capi.SpectrogramConfig(), loudness=capi.LoudnessConfig())):
        with pytest.raises(capi.OmxError) as e:
            make()
        assert e.value.status == capi.ERR_NO_DEVICE


def test_device_selection_entry_points(omx):
    """omx_device_count / omx_set_device: on a GPU-less host 0 devices and OMX_ERR_NO_DEVICE; with a device, index 0 is accepted and an
    index past the count is OMX_ERR_INVALID (nothing is selected)."""
    import openmeters_amd
    count = omx.fn("device_count", C.c_int, [])()
    set_device = omx.fn("set_device", C.c_int, [C.c_int])
    if not openmeters_amd.device_available():
        assert count == 0 and set_device(0) == -4   # OMX_ERR_NO_DEVICE
        return
    assert count >= 1 and set_device(0) == 0
    assert set_device(count) == -3 and set_device(-1) == -3   # OMX_ERR_INVALID
    assert openmeters_amd.device_available()


def test_pure_integer_helpers_work_without_a_device(omx, oracle):
    from openmeters_amd import capi
    assert omx.pack_classic_db(-140.0) == oracle.pack_classic_db(-140.0)
    for db in np.linspace(-150.0, 15.0, 331):
        assert omx.pack_classic_db(float(db)) == oracle.pack_classic_db(float(db))
    for kind, pts, req in [(0, 2049, 0), (0, 2049, 8192), (1, 262145, 8192), (1, 513, 3), (0, 8193, 100000)]:
        assert omx.history_columns(kind, pts, req) == oracle.history_columns(kind, pts, req)
    for ch in range(1, 9):
        assert omx.positions_fallback(ch) == oracle.positions_fallback(ch) == capi.positions_fallback(ch)
    partial = [capi.POS_FL, capi.POS_FL, capi.POS_FR] + [capi.POS_UNKNOWN] * 5
    assert omx.positions_normalize(3, partial) == oracle.positions_normalize(3, partial)
    for f in (0.0, 1.0, 31.5, 1000.0, 16000.0):
        assert omx.a_weight(f) == oracle.a_weight(f)
    b1, a1 = omx.k_weighting_coefficients(48000.0)
    b2, a2 = oracle.k_weighting_coefficients(48000.0)
    assert np.array_equal(b1, b2) and np.array_equal(a1, a2)


def test_shard_streams_partition():
    from openmeters_amd.sharding import shard_streams
    for total, world in [(8192, 8), (64, 1), (10, 4), (3, 8)]:
        spans = [shard_streams(total, r, world) for r in range(world)]
        assert sum(c for _, c in spans) == total
        pos = 0
        for first, count in spans:
            assert first == pos or count == 0
            pos += count
    assert shard_streams(8192, 3, 8) == (3072, 1024)


WORKER = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
from openmeters_amd import capi
from openmeters_amd.capi import AudioBlock, SpectrogramConfig, SpectrogramProcessor
from openmeters_amd.sharding import shard_streams, gather_stats
from signals import exp_sweep
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
oracle = capi.Api(os.path.join({root!r}, "oracle", "libomx_oracle.so"), "omxo_")
TOTAL = 5
def stats_for(s):
    cfg = SpectrogramConfig(fft_size=256, hop_size=64, history_length=64)
    left = exp_sweep(2048, phase0=0.3 * s)
    up = SpectrogramProcessor(oracle, cfg).process_block(AudioBlock(np.stack([left, 0.8 * left], 1).reshape(-1), 2, 48000.0))
    counts = [len(c) for c in up.new_columns]
    return [float(len(counts)), float(np.mean(counts)), float(counts[-1])]
first, count = shard_streams(TOTAL, rank, world)
local = torch.tensor([stats_for(s) for s in range(first, first + count)], dtype=torch.float32).reshape(count, 3)
table = gather_stats(local, TOTAL)
full = torch.tensor([stats_for(s) for s in range(TOTAL)], dtype=torch.float32)
assert table.shape == (TOTAL, 3) and torch.equal(table, full), (rank, table, full)
dist.barrier()
dist.destroy_process_group()
os.write(1, ("rank %d ok\n" % rank).encode())  # one write: the two ranks share the pipe
"""


def test_sharded_streams_gather_over_gloo_world_size_2(oracle, tmp_path):
    import socket
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    with socket.socket() as sock:   # a free port: a fixed one may still sit in TIME_WAIT from the previous run
        sock.bind(("127.0.0.1", 0))
        port = str(sock.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", port, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "rank 0 ok" in r.stdout and "rank 1 ok" in r.stdout


def test_generated_rust_ffi_layer_covers_the_header():
    """bindings/rust/omx_sys.rs (tools/gen_rust_ffi.py) is what the reference — a Rust crate — would link against: every
    function of include/omx.h must be there, with the header's parameter count; every struct with the header's field count."""
    import re
    import subprocess
    import sys
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_ffi.py")], check=True, capture_output=True)
    rs = open(os.path.join(ROOT, "bindings", "rust", "omx_sys.rs")).read()
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "omx.h")).read(), flags=re.S)
    header = "\n".join(l for l in header.split("\n") if not l.lstrip().startswith("#"))
    fns = dict(re.findall(r"pub fn (omx_\w+)\((.*?)\)", rs))
    for s in declared_symbols():
        assert s in fns, s
        m = re.search(r"\b" + s + r"\s*\(([^;{}]*?)\)\s*;", header, flags=re.S)
        params = [p for p in m.group(1).split(",") if p.strip() and p.strip() != "void"]
        assert len([p for p in fns[s].split(",") if p.strip()]) == len(params), s
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", header, flags=re.S):
        # field NAMES in header order, multi-declarator fields (`T a, b;`) split: one `pub name: type,` line each
        names = []
        for f in m.group(2).split(";"):
            for d in f.split(","):
                d = " ".join(d.split())
                if d:
                    names.append(re.match(r"^.*?([A-Za-z_]\w*)(?:\[[^\]]+\])*$", d).group(1))
        body = re.search(r"pub struct " + m.group(3) + r" \{\n(.*?)\n\}", rs, flags=re.S).group(1)
        got = re.findall(r"^    pub (?:r#)?(\w+): ([^\n]+),$", body, flags=re.M)
        assert [g[0] for g in got] == names, (m.group(3), [g[0] for g in got], names)
        assert len(got) == len(body.strip().split("\n")), m.group(3)   # nothing but well-formed field lines
        for _, ty in got:   # no C type token, no stray comma survives into a Rust type
            assert not re.search(r"\b(uint\d+_t|int\d+_t|float|double|unsigned|size_t|struct)\b|,", ty.replace("; ", ";")), (m.group(3), ty)
    assert ",," not in rs and "define" not in rs and rs.count("#[repr(C)]") >= 30


@pytest.mark.parametrize("rate,frames", [(48000.0, 256), (44100.0, 512), (96000.0, 1024), (192000.0, 1024)])
def test_k_weighting_block_transition_is_exact_to_double_double(omx, rate, frames):
    """Host logic of the chunk-parallel loudness form, no device: the block transition T = A^frames of the K-weighting TDF-II and its
    powers (loudness.cpp: k_weighting_transitions, double-double recurrence + double-double squaring) against an 80-digit mpmath
    evaluation of the same recurrence on the same f64 coefficients.  The entries reach 3e5 at 192 kHz and their products with the
    state cancel to 1e-6 of their size (an f64 table put 1e-3 dB into a 15 Hz channel there), so every entry must be right to
    ~1e-30 of the largest: hi + lo as two f64."""
    import mpmath as mp
    mp.mp.prec = 300
    out = (C.c_double * 192)()
    f = omx.fn("debug_k_weighting_transition", C.c_int, [C.c_double, C.c_uint64, C.POINTER(C.c_double)])
    assert f(rate, frames, out) == 0
    got = np.array(out[:]).reshape(2, 6, 4, 4)
    # the f64 coefficients, as the library computes them (reference loudness/processor.rs:22-55)
    import math
    f0, g, q = 1681.974450955533, 3.999843853973347, 0.7071752369554196
    k = math.tan(math.pi * f0 / rate)
    vh = 10.0 ** (g / 20.0)
    vb = vh ** 0.4996667741545416
    a0 = 1.0 + k / q + k * k
    pa = [1.0, 2.0 * (k * k - 1.0) / a0, (1.0 - k / q + k * k) / a0]
    f0, q = 38.13547087602444, 0.5003270373238773
    k = math.tan(math.pi * f0 / rate)
    a0 = 1.0 + k / q + k * k
    ra = [1.0, 2.0 * (k * k - 1.0) / a0, (1.0 - k / q + k * k) / a0]
    a = [pa[0] * ra[0], pa[0] * ra[1] + pa[1] * ra[0], pa[0] * ra[2] + pa[1] * ra[1] + pa[2] * ra[0], pa[1] * ra[2] + pa[2] * ra[1], pa[2] * ra[2]]
    am = [mp.mpf(v) for v in a]
    T = mp.matrix(4, 4)
    for m in range(4):
        s = [mp.mpf(0)] * 4
        s[m] = mp.mpf(1)
        for _ in range(frames):   # one zero-input step (:153-162): y = f0; f0' = f1 - a1 y; f1' = f2 - a2 y; f2' = f3 - a3 y; f3' = -a4 y
            y = s[0]
            s = [s[1] - am[1] * y, s[2] - am[2] * y, s[3] - am[3] * y, -am[4] * y]
        for r in range(4):
            T[r, m] = s[r]
    P = T
    scale0 = max(abs(T[i, j]) for i in range(4) for j in range(4))
    worst = mp.mpf(0)
    for p in range(6):
        for i in range(4):
            for j in range(4):
                pair = mp.mpf(float(got[0, p, i, j])) + mp.mpf(float(got[1, p, i, j]))
                # what the scan adds is T^(2^p) x with |x| of the states' size whatever p: the error that matters is absolute, measured
                # against the block transition's own largest entry (the higher powers decay to 1e-12 and are formed by squaring)
                worst = max(worst, abs(pair - P[i, j]) / scale0)
                assert abs(got[1, p, i, j]) <= abs(got[0, p, i, j]) * 2.0 ** -52 + 1e-300   # a normalised pair
        P = P * P
    print('worst error / max |T| = 2^%.1f' % float(mp.log(worst, 2)))
    assert worst <= mp.mpf(2) ** -78   # measured 2^-92 ... 2^-83 (squared pairs instead of per-power recurrences: 2^-54 at 192 kHz); an f64 table is 2^-53
