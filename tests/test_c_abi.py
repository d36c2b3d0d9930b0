"""The boundary is a C ABI: a plain C99 program (tests/c_abi/abi_demo.c) must compile against include/omx.h alone, link the
shared library, and — on the GPU box — produce the same numbers from the HIP product as from the oracle."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi", "abi_demo.c")


def build(tmp_path, oracle_lib):
    out = str(tmp_path / ("demo_oracle" if oracle_lib else "demo_hip"))
    libdir = os.path.join(ROOT, "oracle") if oracle_lib else os.path.join(ROOT, "openmeters_amd", "csrc")
    lib = "omx_oracle" if oracle_lib else "omx_hip"
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-o", out,
           "-L", libdir, "-l" + lib, "-lm", "-Wl,-rpath," + libdir]
    if oracle_lib:
        cmd.insert(1, "-DOMX_PREFIX_ORACLE")
    else:
        cmd += ["-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"]
    subprocess.run(cmd, check=True, capture_output=True)
    return out


def parse(line):
    tok = line.split()
    return {tok[i]: float(tok[i + 1]) for i in range(0, len(tok), 2)}


def test_c99_program_builds_against_the_header_and_runs_on_the_oracle(tmp_path, oracle):
    for lib in (True, False):          # linking the product needs no GPU
        exe = build(tmp_path, lib)
    r = subprocess.run([build(tmp_path, True)], capture_output=True, text=True, check=True)
    v = parse(r.stdout.strip())
    assert v["columns"] == 200 - 31 and abs(v["peak_freq"] - 1000.0) < 0.5 and abs(v["spectrum_peak_hz"] - 996.09) < 12.0
    assert abs(v["true_peak"] - 20 * 0.30103 * -1) < 0.05  # 0.5 amplitude = -6.02 dBTP


@pytest.mark.gpu
def test_c99_program_gets_the_same_numbers_from_the_hip_library(tmp_path, oracle):
    ro = subprocess.run([build(tmp_path, True)], capture_output=True, text=True, check=True)
    rh = subprocess.run([build(tmp_path, False)], capture_output=True, text=True)
    assert rh.returncode == 0, rh.stderr
    a, b = parse(rh.stdout.strip()), parse(ro.stdout.strip())
    assert a["columns"] == b["columns"] and abs(a["points"] - b["points"]) <= 4 * a["columns"]
    assert abs(a["peak_freq"] - b["peak_freq"]) < 1e-3 and abs(a["peak_power"] - b["peak_power"]) <= 1e-5 * b["peak_power"]
    assert a["spectrum_peak_hz"] == b["spectrum_peak_hz"] and abs(a["spectrum_peak_db"] - b["spectrum_peak_db"]) < 0.01
    assert abs(a["momentary"] - b["momentary"]) < 1e-4 and abs(a["true_peak"] - b["true_peak"]) < 1e-4


GROUP_SRC = os.path.join(ROOT, "tests", "c_abi", "group_demo.c")


def build_group(tmp_path):
    out = str(tmp_path / "group_demo")
    libdir = os.path.join(ROOT, "openmeters_amd", "csrc")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), GROUP_SRC, "-o", out,
           "-L", libdir, "-lomx_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib",
           "-Wl,--allow-shlib-undefined"]
    subprocess.run(cmd, check=True, capture_output=True)
    return out


def test_c99_capture_group_host_builds_against_the_header(tmp_path, omx):
    """the VisualManager fan-out is reachable from plain C: tests/c_abi/group_demo.c compiles (-Wall -Wextra -pedantic -Werror) and links"""
    assert os.path.exists(build_group(tmp_path))


@pytest.mark.gpu
def test_c99_capture_group_drives_three_visuals_with_one_ingest_per_block(tmp_path, omx):
    """group_demo.c: a 3-visual group (Spectrogram + Loudness + Stereometer) of 4 captures fed from device memory; its summary rows
    against the single-stream handles of the same blocks (bit-order-exact meters: LUFS within 1e-4 dB, rho within 1e-6)"""
    r = subprocess.run([build_group(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    v = parse(r.stdout.strip())
    assert v["columns"] == (9 * 2048 - 2048) // 256 + 1
    assert v["worst_lufs"] < 1e-4 and v["worst_rho"] < 1e-6 and v["rho_full"] < -0.99


BATCHER_BANK_SRC = os.path.join(ROOT, "tests", "c_abi", "batcher_bank_demo.c")


def build_batcher_bank(tmp_path):
    out = str(tmp_path / "batcher_bank_demo")
    libdir = os.path.join(ROOT, "openmeters_amd", "csrc")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), BATCHER_BANK_SRC, "-o", out,
           "-L", libdir, "-lomx_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib",
           "-Wl,--allow-shlib-undefined"]
    subprocess.run(cmd, check=True, capture_output=True)
    return out


def test_c99_batcher_bank_host_builds_against_the_header(tmp_path, omx):
    """the device-side batchers are reachable from plain C: tests/c_abi/batcher_bank_demo.c compiles (-Wall -Wextra -pedantic -Werror) and links"""
    assert os.path.exists(build_batcher_bank(tmp_path))


@pytest.mark.gpu
def test_c99_batcher_bank_emits_the_host_batchers_chunks(tmp_path, omx):
    """batcher_bank_demo.c: five captures, thirty packet rounds from device memory; every chunk and every remainder against the host
    batcher of the same library (DspBatcher for one capture, KAT-pinned), compared in C with memcmp"""
    r = subprocess.run([build_batcher_bank(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    v = parse(r.stdout.strip())
    assert v["mismatches"] == 0 and v["chunks"] > 100 and v["multi"] > 5


MANAGER_SRC = os.path.join(ROOT, "tests", "c_abi", "group_manager.c")


def build_manager(tmp_path):
    out = str(tmp_path / "group_manager")
    libdir, oradir = os.path.join(ROOT, "openmeters_amd", "csrc"), os.path.join(ROOT, "oracle")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), MANAGER_SRC, "-o", out,
           "-L", libdir, "-lomx_hip", "-L", oradir, "-lomx_oracle", "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + libdir,
           "-Wl,-rpath," + oradir, "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"]
    subprocess.run(cmd, check=True, capture_output=True)
    return out


def test_c99_visual_manager_host_builds_against_the_header(tmp_path, omx, oracle):
    """set_enabled / update_config / note_format / ingest_ragged are reachable from plain C (product and oracle linked side by side)"""
    assert os.path.exists(build_manager(tmp_path))


@pytest.mark.gpu
def test_c99_capture_group_behaves_like_one_visual_manager_per_capture(tmp_path, omx, oracle):
    """group_manager.c: four captures with uneven per-call frame counts, a visual toggled off and on, another created by set_enabled,
    the spectrogram's hop changed mid-stream, one capture reset alone, a format-generation reset — every capture against single-stream
    ORACLE handles fed the same sequence: column counts and `reset` flags exact, LUFS within 1e-4 dB, correlations within 1e-6"""
    r = subprocess.run([build_manager(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    v = parse(r.stdout.strip())
    assert v["columns"] > 100 and v["blocks"] > 40 and v["rho_checks"] > 60   # (blocks = chunks: one per capture and call since round 5)
    assert v["worst_lufs"] < 1e-4 and v["worst_rho"] < 1e-6


RCCL_SRC = os.path.join(ROOT, "tests", "c_abi", "group_rccl.c")


def build_rccl(tmp_path):
    out = str(tmp_path / "group_rccl")
    libdir = os.path.join(ROOT, "openmeters_amd", "csrc")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), RCCL_SRC, "-o", out,
           "-L", libdir, "-lomx_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lrccl", "-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib",
           "-Wl,--allow-shlib-undefined"]
    subprocess.run(cmd, check=True, capture_output=True)
    return out


def parse_rank_line(line):
    tok = line.split()
    return {tok[i]: float(tok[i + 1]) for i in range(0, len(tok) - 1, 2)}


def test_c99_rccl_host_builds_and_restates_the_shard_rule(tmp_path, omx):
    """tests/c_abi/group_rccl.c — the native N-rank host north_star describes (one process per GPU, its own RCCL communicator, ncclAllGather on
    the capture group's summary rows) — compiles as C99 against include/omx.h + librccl, and its shard rule is sharding.shard_streams"""
    from openmeters_amd.sharding import shard_streams
    exe = build_rccl(tmp_path)
    for total, world in ((8192, 8), (16, 1), (10, 4), (3, 8), (1000, 7)):
        r = subprocess.run([exe, "--shards", str(total), str(world)], capture_output=True, text=True, check=True)
        got = [tuple(int(x) for x in ln.split()) for ln in r.stdout.strip().splitlines()]
        assert got == [(k,) + tuple(shard_streams(total, k, world)) for k in range(world)]


@pytest.mark.gpu
def test_c99_rccl_host_gathers_the_summary_rows(tmp_path, omx):
    """world size 1 on every box (the RCCL communicator, the all-gather and the capture group from one plain C process); with more than one
    GPU visible also one process per GPU — every rank must print the SAME gathered table (checksum), and that table must equal the
    one-rank run's: captures are independent, so sharding them changes nothing."""
    import torch
    exe = build_rccl(tmp_path)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    total = 16
    one = subprocess.run([exe, str(total), "9"], capture_output=True, text=True, env=env, timeout=600)
    assert one.returncode == 0, one.stdout + one.stderr
    v = parse_rank_line(one.stdout.strip().splitlines()[-1])
    assert v["world"] == 1 and v["rows"] == total and v["finite"] == 1 and v["shard_count"] == total
    assert -10.0 < v["mean_momentary"] < 0.0 and -1.0 <= v["mean_rho"] < -0.9     # 0.5-amplitude tones, R = -side * L
    n = torch.cuda.device_count()
    if n < 2:
        return
    id_file = str(tmp_path / "nccl_id")
    procs = [subprocess.Popen([exe, str(total), "9"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), OMX_RCCL_ID_FILE=id_file)) for r in range(n)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    lines = [parse_rank_line(o[0].strip().splitlines()[-1]) for o in outs]
    assert all(l["world"] == n and l["rows"] == total and l["finite"] == 1 for l in lines)
    assert len({l["checksum"] for l in lines}) == 1
    assert abs(lines[0]["checksum"] - v["checksum"]) <= 1e-6 * abs(v["checksum"])
