"""Soak of the ragged bank tests and the random operation sequences with seeds the suite does not use (run on the GPU box): python tools/soak_ragged.py [first] [count]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest
import openmeters_amd
from openmeters_amd.capi import Api

import test_gpu_state_machine as t

omx = openmeters_amd.api()
oracle = Api(conftest._build_oracle(), "omxo_")
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 10
failures = 0
for seed in range(first, first + count):
    cases = [("loudness chunk", lambda: t.test_ragged_loudness_bank_chunk_parallel_form_matches_per_stream_oracles(
                  omx, oracle, seed, (2, 6, 8, 3)[seed % 4], (48000.0, 44100.0, 48000.0, 96000.0)[seed % 4], (256, 128, 64, 512)[seed % 4])),
             ("loudness seq", lambda: t.test_ragged_loudness_bank_random_per_stream_block_counts_match_per_stream_oracles(
                  omx, oracle, seed, (2, 8, 6, 1, 3)[seed % 5], (48000.0, 48000.0, 96000.0, 44100.0, 192000.0)[seed % 5], (256, 256, 100, 37, 64)[seed % 5])),
             ("stereometer chunk", lambda: t.test_ragged_stereometer_bank_random_per_stream_block_counts_match_per_stream_oracles(
                  omx, oracle, seed, 2, seed % 3 != 2, seed % 3 == 0, 2)),
             ("stereometer seq", lambda: t.test_ragged_stereometer_bank_random_per_stream_block_counts_match_per_stream_oracles(
                  omx, oracle, seed, 2 if seed % 2 else 6, True, True, 0)),
             ("waveform", lambda: t.test_ragged_waveform_bank_random_per_stream_frame_counts_match_per_stream_oracles(
                  omx, oracle, seed, 2 if seed % 3 else 6, seed % 2 == 0, (48000.0, 44100.0, 8000.0)[seed % 3])),
             ("oscilloscope", lambda: t.test_ragged_oscilloscope_bank_random_per_stream_block_counts_match_per_stream_processors(
                  omx, oracle, seed, t.capi.TRIGGER_ZERO_CROSSING if seed % 3 == 2 else t.capi.TRIGGER_STABLE)),
             ("oscilloscope ops", lambda: t.test_oscilloscope_random_operation_sequences(omx, oracle, seed)),
             ("spectrogram ops", lambda: t.test_spectrogram_random_operation_sequences(omx, oracle, seed)),
             ("meters", lambda: t.test_meter_processors_random_block_sequences(omx, oracle, seed))]
    for name, fn in cases:
        try:
            fn()
        except Exception as e:  # noqa: BLE001 - report and keep going
            failures += 1
            print(f"FAIL seed {seed} {name}: {type(e).__name__}: {str(e)[:300]}", flush=True)
print(f"RESULT: {count} seeds x 9 cases, {failures} failures")
sys.exit(1 if failures else 0)
