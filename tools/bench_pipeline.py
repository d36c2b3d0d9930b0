"""cfg5 per-GPU shard: 1024 independent 2-ch streams through the full pipeline (reassigned STFT + LUFS + correlation), one
step = 16384 new frames per stream (64 STFT columns, 64 blocks of 256).  Run on the GPU box.
  serial      the three banks one after the other on one stream, rows by torch glue (tests/pipeline_reference.py)
  group       omx_capture_group_ingest: what bench.py --config cfg5 times (fan-out, side streams and rows inside the library)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import openmeters_amd
from openmeters_amd.pipeline import FullPipeline
from pipeline_reference import SeparateBanks


def bench_meters_parity(*rows):
    import bench_meters
    return bench_meters.parity_bar(*rows)


def shard_pipeline(S=1024, frames=16384, out=sys.stdout):
    api = openmeters_amd.api()
    dev = torch.device("cuda", 0)
    t = torch.arange(frames * 6, device=dev, dtype=torch.float32)
    base = 0.4 * torch.sin(t * 0.05 + 1e-7 * t * t)
    pcm = (base[None, :, None] * torch.tensor([1.0, -0.7], device=dev)[None, None, :] + 0.001 * torch.randn((S, frames * 6, 2), device=dev)).contiguous()
    res = {"workload": f"{S} streams x 2 ch: reassigned STFT 4096/256 + LUFS + band correlation, {frames} frames per step"}
    for mode in ("serial", "group"):
        chunks = [pcm[:, k * frames:(k + 1) * frames].contiguous() for k in range(6)]
        if mode == "serial":
            pipe = SeparateBanks(api, S)

            def run(c):
                r = pipe.step(c.data_ptr(), frames, torch.cuda.current_stream().cuda_stream)
                return pipe.stats(torch, dev, *r)
        else:
            pipe = FullPipeline(api, S)
            run = lambda c: pipe.step_with_stats(torch, dev, c.data_ptr(), frames)[1]
        run(chunks[0])
        run(chunks[1])
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        timed = 24
        for k in range(timed):
            table = run(chunks[2 + k % 4])
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / timed
        print(f"{mode}: {ms:.3f} ms per step of {S} streams x {frames} frames -> {S * frames / ms / 1e6:.2f} G stream-frames/s, "
              f"{S * (frames // 256) / ms / 1e3:.2f} M STFT frames/s, {frames / 48000.0 / (ms * 1e-3):.0f}x real time; "
              f"rho mean {float(table[:, 3].mean()):.3f}", file=out)
        res[mode] = {"ms_per_step": ms, "stft_frames_per_s": S * (frames // 256) / (ms * 1e-3), "x_real_time": frames / 48000.0 / (ms * 1e-3),
                     "form": "K2 fused (stft_reassigned_4096_tri_kernel) + chunk-parallel loudness and stereometer (the default of a 64-block call)",
                     "parity_bar": bench_meters_parity("reassigned: |dP| / max P", "loudness (chunk-parallel): |d rms_fast_db|",
                                                       "stereometer (chunk-parallel): |d point| vs oracle")}
        if mode == "group":
            # the step's algorithmic bytes: K2 26 640 B per STFT frame (PCM hop in, 2049 points out) + the loudness form's
            # 8 B per channel-sample + the stereometer's 16 B per stereo frame (the PCM itself is counted once, with K2)
            cols = S * (frames // 256)
            alg = cols * 26640.0 + S * frames * 2 * 8.0 + S * frames * 16.0 - 2 * S * frames * 2 * 4.0
            res[mode]["roofline"] = {"bound": "hbm", "achieved": alg / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                     "frac": alg / (ms * 1e-3) / 1e9 / 8000.0, "algorithmic_bytes_per_step": alg, "ms_per_step": ms,
                                     "traffic": None}
        del pipe
    return res


if __name__ == "__main__":
    shard_pipeline()
