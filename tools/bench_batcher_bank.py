"""omx_batcher_bank_push at the reference's cadence (run on the GPU box): 1024 captures x one packet each, packets resident on the device,
against 1024 host batchers (omx_batcher_push: the reference's structure, one DspBatcher per capture) fed the same packets from host memory.
  python tools/bench_batcher_bank.py [captures] [pushes]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch

import openmeters_amd
from openmeters_amd import capi
from openmeters_amd.pipeline import BatcherBank

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
pushes = int(sys.argv[2]) if len(sys.argv) > 2 else 200
api = openmeters_amd.api()
pos = capi.positions_fallback(2)
rng = np.random.default_rng(1)
# (the first bank of a process times the runtime's own warm-up — 240 ... 310 us per push against 22 for the same pushes later: not printed)
for name, sizes in ((None, [256]), ("256-frame packets", [256]), ("PipeWire quanta of 441 / 480 / 512 / 1024 frames", [441, 480, 512, 1024])):
    bank = BatcherBank(api, S, 1024)
    packets = torch.rand((S, 1024, 2), device="cuda:0")
    lengths = [np.full(S, sizes[k % len(sizes)], np.uint32) for k in range(8)]
    rounds = 0
    for k in range(20):
        rounds += len(bank.push(packets.data_ptr(), 1024, lengths[k % 8], 2, 48000.0, pos, generation=1))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(pushes):
        bank.push(packets.data_ptr(), 1024, lengths[k % 8], 2, 48000.0, pos, generation=1)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / pushes * 1e6
    if name:
        print(f"batcher bank, {S} captures, {name}: {us:.1f} us per push (all captures)")
    bank.close()

# MeterEngine::poll for S captures (meter.rs:100-143): one packet per capture -> the bank -> one omx_capture_group_ingest_ragged per round
from openmeters_amd.pipeline import CaptureGroup
import bench_stream
for name, sizes in (("256-frame packets", [256]), ("PipeWire quanta of 441 / 480 / 512 / 1024 frames", [441, 480, 512, 1024])):
    bank = BatcherBank(api, S, 1024)
    group = CaptureGroup(api, S, stats=True, **bench_stream.default_configs())
    n = torch.arange(1024, device="cuda:0", dtype=torch.float32)
    base = (0.4 * torch.sin(2 * torch.pi * 440.0 * n / 48000.0))[None, :, None] * torch.tensor([1.0, -0.7], device="cuda:0")[None, None, :]
    packets = (base + 0.01 * (torch.rand((S, 1024, 2), device="cuda:0") - 0.5)).contiguous()
    lengths = [np.full(S, sizes[k % len(sizes)], np.uint32) for k in range(8)]

    def poll(k):
        calls = 0
        for ptr, cap, frames in bank.push(packets.data_ptr(), 1024, lengths[k % 8], 2, 48000.0, pos, generation=1):
            group.ingest_ragged(ptr, cap, frames, 2, 48000.0, pos)
            calls += 1
        return calls
    for k in range(40):
        poll(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    calls = frames_in = 0
    for k in range(pushes):
        calls += poll(k)
        frames_in += int(lengths[k % 8][0])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"packets -> batcher bank -> six-visual capture group, {S} captures, {name}: {dt / pushes * 1e6:.0f} us per packet round "
          f"({calls / pushes:.2f} ingest calls per round), {frames_in / 48000.0 / dt:.1f}x real time")
    group.close()
    bank.close()

# the reference's structure: one host batcher per capture, samples through host memory
from test_kat_batcher import Batcher, fmt
hosts = [Batcher(api) for _ in range(min(S, 1024))]
f = fmt(2, 48000.0, 1)
host_packet = np.random.default_rng(2).uniform(-1, 1, (256, 2)).astype(np.float32).reshape(-1)
noop = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_float), C.c_uint64, C.c_void_p)(lambda *a: None)
push = api.fn("batcher_push", C.c_uint64, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p])
t0 = time.perf_counter()
for k in range(20):
    for b in hosts:
        push(b.h, host_packet.ctypes.data, host_packet.size, C.byref(f), noop, None)
us = (time.perf_counter() - t0) / 20 * 1e6
print(f"host batchers (omx_batcher_push per capture, ctypes loop, no-op ingest), {len(hosts)} captures, 256-frame packets: {us:.0f} us per round")
