"""debug: one spectrogram random operation sequence (tests/test_gpu_state_machine.py) with every conditioned bar that is exceeded
printed together with the points around the worst pair on both sides: python tools/debug/spec_seq.py <seed>"""
import os, sys
ROOT = os.path.join(os.path.dirname(__file__), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import openmeters_amd
from openmeters_amd.capi import Api
import parity
import test_gpu_state_machine as t

omx = openmeters_amd.api()
oracle = Api(os.path.join(ROOT, "oracle", "libomx_oracle.so"), "omxo_")
seed = int(sys.argv[1])
last = {}
real_metrics = parity.reassigned_column_metrics


def metrics(h, o, rate, hop):
    last["pair"] = (np.array(h), np.array(o), rate, hop)
    if last.get("n", 0) % 2 == 0:      # the test calls this twice per column: (hip, oracle), then (perturbed oracle, oracle)
        last["first"] = last["pair"]
    last["n"] = last.get("n", 0) + 1
    return real_metrics(h, o, rate, hop)


def cbar(name, err, fixed, sens, detail=None, plain=True):
    limit = float(fixed) + parity.CONDITIONING_K * float(sens)
    if float(err) / limit > 1.0:
        h, o, rate, hop = last["first"]
        print("EXCEEDED", name, "err", err, "fixed", fixed, "sens", sens, "ratio", float(err) / limit)
        print("  detail", detail[:2], "scale", detail[-1])
        mp = float(max(o[:, 2].max(), h[:, 2].max()))
        pairs, oa, ob = parity.align_points(h, o, mp)
        pa = np.array([p[0] for p in pairs]); pb = np.array([p[1] for p in pairs])
        r = np.sqrt(o[pb, 2] / mp)
        dt = np.abs(h[pa, 0].astype(np.float64) - o[pb, 0]) * r
        k = int(dt.argmax())
        print("  worst pair index", k, "of", len(pairs), "orphans hip", oa, "oracle", ob, "column max", mp)
        for q in range(max(0, k - 4), min(len(pairs), k + 5)):
            print("   hip", h[pa[q]], " oracle", o[pb[q]], " r", r[q])
        np.save(f"/tmp/spec_h_{seed}.npy", h)
        np.save(f"/tmp/spec_o_{seed}.npy", o)


t.reassigned_column_metrics = metrics
t.conditioned_bar = cbar
real_signal = t.signal


def signal(rng, frames, channels, t0, rate, silent):
    print("block: frames", frames, "channels", channels, "rate", rate, "silent", silent, "t0", t0)
    return real_signal(rng, frames, channels, t0, rate, silent)


t.signal = signal
real_cfg = t.SpectrogramConfig


def cfg(*a, **k):
    c = real_cfg(*a, **k)
    print("config:", {n: getattr(c, n) for n in ("fft_size", "hop_size", "window", "use_reassignment", "zero_padding_factor", "history_length")})
    return c


t.SpectrogramConfig = cfg
t.test_spectrogram_random_operation_sequences(omx, oracle, seed)
