"""Reassigned 4096 / 256 per window kind (run on the GPU box): the three-workgroups-per-CU kernel (form 0, every window on the bins) and — with the
tuning library loaded (OMX_HIP_LIB=.../libomx_hip_tuning.so; the product refuses superseded forms since round 5) — the round-1
five-transform kernel (form 1, windows in the time domain)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import openmeters_amd
from openmeters_amd import banks, capi
api = openmeters_amd.api()
S, W, hop = 64, 4096, 256
cols = 65536 // S
frames = 2 * W + hop * (cols - 1)
n = torch.arange(frames + hop * cols * 4, device="cuda:0", dtype=torch.float64)
tone = (0.5 * torch.sin(2 * torch.pi * (300.0 + 0.002 * n) * n / 48000.0)).to(torch.float32)
pcm = (tone[None, :, None] * torch.tensor([1.0, 0.8], device="cuda:0")[None, None, :] + 0.001 * (torch.rand((S, len(n), 2), device="cuda:0") - 0.5)).contiguous()
for form in ((0, 1) if "tuning" in os.path.basename(openmeters_amd.LIB_PATH) else (0,)):
  for kind, name in enumerate(["rectangular", "hann", "hamming", "blackman", "blackman-harris"]):
    bank = banks.SpectrogramBank(api, capi.SpectrogramConfig(fft_size=W, hop_size=hop, window=kind, use_reassignment=True, history_length=8192), S)
    bank.set_option(capi.OPT_KERNEL_TIMING, 1)
    bank.set_option(capi.OPT_KERNEL_FORM, form)
    pos = capi.positions_fallback(2)
    bank.process_device(pcm[:, :frames].contiguous().data_ptr(), frames, 2, 48000.0, pos)
    bank.kernel_time()
    for it in range(4):
        chunk = pcm[:, frames + it * hop * cols: frames + (it + 1) * hop * cols].contiguous()
        bank.process_device(chunk.data_ptr(), hop * cols, 2, 48000.0, pos)
    torch.cuda.synchronize()
    ms, k = bank.kernel_time()
    print(f"4096/256 form {form} {name}: kernel {ms:.3f} ms per {S * cols} frames -> {S * cols / ms / 1e3:.1f} M frames/s")
