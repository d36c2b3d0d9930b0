import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import openmeters_amd
from openmeters_amd import banks, capi
api = openmeters_amd.api()
dev = torch.device("cuda", 0)
S, blocks, fs = 256, 64, float(sys.argv[1]) if len(sys.argv) > 1 else 96000.0
block = int(round(256 * fs / 48000.0)); frames = block * blocks
n = torch.arange(frames, device=dev, dtype=torch.float64)
pcm = torch.empty((S, frames, 2), device=dev, dtype=torch.float32)
for s in range(S):
    f = 440.0 * 2.0 ** ((s % 24) / 12.0)
    left = (0.8 * torch.sin(2 * np.pi * f * n / fs)).to(torch.float32)
    pcm[s, :, 0] = left; pcm[s, :, 1] = -0.7 * left
pos = capi.positions_fallback(2)
sc = banks.OscilloscopeBank(api, capi.OscilloscopeConfig(sample_rate=fs, segment_duration=0.02, trigger_mode=capi.TRIGGER_STABLE, num_cycles=2, trigger_source=capi.CH_LEFT, channel_1=capi.CH_LEFT, channel_2=capi.CH_RIGHT), S)
for _ in range(4):
    sc.process_device(pcm.data_ptr(), block, blocks, 2, fs, pos, 0)
torch.cuda.synchronize()
