"""Known-answer tests ported from reference src/meter.rs:194-276 (DspBatcher, ingest_silence): the block
partition the processors' caller imposes.  Host-side integer logic: runs on CPU for both
the oracle and the product library (no GPU needed)."""
import ctypes as C

import numpy as np
import pytest

from openmeters_amd import capi

INGEST = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_float), C.c_uint64, C.c_void_p)
RESET = C.CFUNCTYPE(None, C.c_void_p)


class Fmt(C.Structure):
    _fields_ = [("generation", C.c_uint64), ("sample_rate", C.c_float), ("channels", C.c_uint32), ("positions", C.c_uint8 * 8)]


def fmt(channels, rate, generation):
    return Fmt(generation, rate, channels, (C.c_uint8 * 8)(*capi.positions_fallback(channels)))


class Batcher:
    def __init__(self, api):
        self.api = api
        self.h = C.c_void_p()
        api.fn("batcher_create", C.c_int, [C.POINTER(C.c_void_p)])(C.byref(self.h))
        self.blocks = []
        self.resets = 0
        self._ingest = INGEST(lambda user, p, n, f: self.blocks.append(np.ctypeslib.as_array(p, shape=(n,)).copy()))
        self._reset = RESET(lambda user: setattr(self, "resets", self.resets + 1))

    def push(self, samples, f):
        s = np.ascontiguousarray(samples, np.float32)
        return self.api.fn("batcher_push", C.c_uint64, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, INGEST, C.c_void_p])(
            self.h, s.ctypes.data, s.size, C.byref(f), self._ingest, None)

    def push_silence(self, frames, f):
        return self.api.fn("batcher_push_silence", C.c_uint64, [C.c_void_p, C.c_uint64, C.c_void_p, INGEST, RESET, C.c_void_p])(
            self.h, frames, C.byref(f), self._ingest, self._reset, None)

    def pending(self):
        n = self.api.fn("batcher_pending", C.c_uint64, [C.c_void_p, C.c_void_p, C.c_uint64])(self.h, None, 0)
        buf = np.zeros(max(int(n), 1), np.float32)
        self.api.fn("batcher_pending", C.c_uint64, [C.c_void_p, C.c_void_p, C.c_uint64])(self.h, buf.ctypes.data, n)
        return buf[:n]

    def format(self):
        out = Fmt()
        ok = self.api.fn("batcher_format", C.c_int, [C.c_void_p, C.c_void_p])(self.h, C.byref(out))
        return out if ok else None


@pytest.fixture(params=["oracle", "omx"])
def lib(request):
    return request.getfixturevalue(request.param)


def test_dsp_batches_are_sample_driven(lib):
    # meter.rs:194-227 (the storage-reuse half is an allocator detail with no ABI meaning)
    b = Batcher(lib)
    f = fmt(2, 48000.0, 1)
    block = np.full(64 * 2, 0.25, np.float32)
    for index in range(4):
        assert b.push(block, f) == (1 if index == 3 else 0)
    assert len(b.pending()) == 0 and [len(x) for x in b.blocks] == [256 * 2]
    hi = fmt(2, 96000.0, 1)
    for index in range(8):
        assert b.push(block, hi) == (1 if index == 7 else 0)
    assert [len(x) for x in b.blocks] == [512, 1024]  # 256 frames @48k scale to 512 frames @96k


def test_dsp_batches_coalesce_large_capture_backlogs(lib):
    # meter.rs:229-240
    b = Batcher(lib)
    f = fmt(2, 48000.0, 1)
    assert b.push(np.full((256 * 6 + 17) * 2, 0.25, np.float32), f) == 2
    assert [len(x) // 2 for x in b.blocks] == [1024, 512]
    assert len(b.pending()) == 17 * 2
    assert b.push(np.full(239 * 2, 0.25, np.float32), f) == 1
    assert len(b.pending()) == 0 and len(b.blocks[-1]) == 256 * 2


def test_dsp_batches_never_mix_format_generations(lib):
    # meter.rs:242-255
    b = Batcher(lib)
    old = fmt(2, 48000.0, 1)
    assert b.push(np.full(128 * 2, 0.25, np.float32), old) == 0
    new = fmt(2, 48000.0, 2)
    assert b.push(np.full(2, 0.5, np.float32), new) == 0
    assert b.pending().tolist() == [0.5, 0.5]
    assert b.format().generation == 2


def test_long_silence_resets_without_replaying_samples(lib):
    # meter.rs:257-276
    b = Batcher(lib)
    f = fmt(8, 192000.0, 1)
    assert b.push(np.full(128 * 8, 0.25, np.float32), f) == 0
    assert b.push_silence(2 * 192000 + 1, f) == 0
    assert len(b.pending()) == 0 and b.format() is None and b.resets == 1 and not b.blocks


def test_short_silence_is_replayed_in_the_same_quanta(lib):
    # ingest_silence :145-166 below the 2 s limit: 4096-frame zero chunks go through push()
    b = Batcher(lib)
    f = fmt(2, 48000.0, 1)
    assert b.push(np.full(100 * 2, 0.25, np.float32), f) == 0
    n = b.push_silence(10000, f)
    total = 100 + 10000
    assert sum(len(x) for x in b.blocks) // 2 == total - total % 256 and len(b.pending()) // 2 == total % 256
    assert n == len(b.blocks) and b.resets == 0
    assert all(len(x) // 2 in (256, 512, 768, 1024) for x in b.blocks)
    assert b.blocks[0][:200].tolist() == [0.25] * 200 and not b.blocks[0][200:].any()
