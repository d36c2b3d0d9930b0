"""The spectrum leg of BASELINE configs[1] alone (run on the GPU box): 64 streams, 4096 / hop 256, A-weighted, every hop materialised;
1024 hops per stream and call.  OMX_HIP_LIB selects an A/B build (tools/build_ab.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import openmeters_amd
from openmeters_amd import banks, capi

api = openmeters_amd.api()
S, N, hop, hops, reps = 64, 4096, 256, 1024, 20
frames = N + hop * (hops - 1)
pcm = (torch.rand((S, frames, 2), device="cuda:0") - 0.5).contiguous()
chunks = [(torch.rand((S, hop * hops, 2), device="cuda:0") - 0.5).contiguous() for _ in range(4)]
bank = banks.SpectrumBank(api, capi.SpectrumConfig(fft_size=N, hop_size=hop), S, emit_all_hops=True)
pos = capi.positions_fallback(2)
bank.process_device(pcm.data_ptr(), frames, 2, 48000.0, pos)
for c in chunks:
    bank.process_device(c.data_ptr(), hop * hops, 2, 48000.0, pos)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for it in range(reps):
    bank.process_device(chunks[it % 4].data_ptr(), hop * hops, 2, 48000.0, pos)
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / reps
print(f"{os.path.basename(os.environ.get('OMX_HIP_LIB', 'libomx_hip.so'))}: spectrum 4096/256 {ms:.4f} ms per {S * hops} hops (ingest included) -> {S * hops / ms / 1e3:.2f} M hops/s")
